"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the reference's shape-aware F0 DTW, pinned to goldens produced by the
unmodified reference (tests/golden/make_dtw_golden.py -> dtw_ref.npz; tests/test_oracle_golden.py).

reference: modules/voice_conversion/dtw/enhance_sadtw.py:18-113, modules/voice_conversion/dtw/align.py:8-37.
"""
import numpy as np

_WIN = [(-64, -48), (-48, -32), (-32, -16), (-16, 0), (0, 16), (16, 32), (32, 48), (48, 64)]
_WGT = [0.5, 0.75, 0.9, 1.0, 1.0, 0.9, 0.75, 0.5]


def shape_hist(f, scale=1.0):
    """cal_hist_of_f0(f, max_window=64, normalize_hist=True, scale_factor=scale) -> float32 [T, 48]  (:18-83)."""
    f = np.asarray(f, np.float64).reshape(-1)
    T = len(f)
    out = np.zeros((T, 48), np.float64)
    for t in range(T):
        tot = 0
        for w, (l, r) in enumerate(_WIN):
            rl, rr = int(l * scale), int(r * scale)
            if rl == 0:
                rl = 1
            lb, rb = min(max(0, rl + t), T), min(max(0, rr + t), T)
            if rb <= lb:
                continue
            i = np.arange(lb, rb)
            diff = f[i] - f[t]
            a = np.abs(diff / (i - t) * _WGT[w]) if _WGT[w] != 1.0 else np.abs(diff / (i - t))
            up = diff >= 0
            region = np.where(a < 0.57735, np.where(up, 2, 3), np.where(a < 1.73205, np.where(up, 1, 4), np.where(up, 0, 5)))
            np.add.at(out[t], w * 6 + region, 1)
            tot += len(i)
        if tot:
            out[t] /= tot
    return out.astype(np.float32)


def hist_cost(ha, hb):
    """cal_hist_dist(ha, hb).T -> float32 [T_b, T_a]  (:86-100)."""
    ha, hb = np.asarray(ha, np.float32), np.asarray(hb, np.float32)
    mi = hb[:, None, :] - ha[None, :, :]
    pl = hb[:, None, :] + ha[None, :, :]
    return ((np.float32(0.5) * (mi * mi)) / (pl + np.float32(0.00000001))).sum(-1, dtype=np.float32)


def time_warp(cost):
    """align.py:8-17 in the cost's own dtype."""
    cost = np.asarray(cost)
    D = np.zeros_like(cost)
    D[0, 1:] = np.inf
    D[1:, 0] = np.inf
    for i in range(1, cost.shape[0]):
        for j in range(1, cost.shape[1]):
            D[i, j] = cost[i, j] + min(D[i - 1, j], D[i, j - 1], D[i - 1, j - 1])
    return D


def backtrack(D):
    """align.py:19-31."""
    i, j = D.shape[0] - 1, D.shape[1] - 1
    res = [0] * D.shape[0]
    while i > 0 and j > 0:
        res[i] = j
        i, j = min([(i - 1, j), (i, j - 1), (i - 1, j - 1)], key=lambda x: D[x[0], x[1]])
    return np.asarray(res, np.int64)


def ehsadtw(src, tgt):
    hs, ht = shape_hist(src), shape_hist(tgt, len(tgt) / len(src))
    return backtrack(time_warp(hist_cost(hs, ht)))

"""Generate tests/golden/dtw_ref.npz from the UNMODIFIED reference's shape-aware F0 DTW (the offline alignment step of the
binarizer: data_gen/singing/binarize_para.py:168-185 -> modules/voice_conversion/dtw/enhance_sadtw.py:18-113 ->
modules/voice_conversion/dtw/align.py:8-37), run as plain Python (numba's @jit stubbed to the identity).

Build-container only.   python tests/golden/make_dtw_golden.py
Per synthetic amateur / professional F0 pair: the normalised slope histograms of both tracks, the chi-square cost matrix
(as handed to align_from_distances, i.e. transposed: [T_prof, S_amateur]), the accumulated-cost matrix of time_warp, the
back-tracked alignment and EHSADTW's aligned track.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_shims  # noqa: E402


def pairs():
    """(amateur f0, professional f0) in Hz, 0 = unvoiced: vibrato + drift, unvoiced gaps, different lengths."""
    rng = np.random.RandomState(7)
    out = []
    for S, T in ((96, 110), (150, 128), (64, 64), (40, 75)):
        ts, tt = np.arange(S), np.arange(T)
        src = 220 * 2 ** (0.25 * np.sin(ts / 9.0) + 0.02 * rng.randn(S)) + 4 * np.sin(ts * 1.1)
        tgt = 225 * 2 ** (0.25 * np.sin(tt * (S / T) / 9.0 + 0.2) + 0.01 * rng.randn(T)) + 3 * np.sin(tt * 0.9)
        for a, n in ((src, S), (tgt, T)):
            for _ in range(3):
                s = rng.randint(0, n - 6)
                a[s:s + rng.randint(2, 6)] = 0.0
        out.append((src, tgt))
    return out


def main():
    ref_shims.install()
    from modules.voice_conversion.dtw import enhance_sadtw as E
    from modules.voice_conversion.dtw.align import time_warp, align_from_distances
    res = {}
    for p, (src, tgt) in enumerate(pairs()):
        S, T = len(src), len(tgt)
        hs = torch.tensor(E.cal_hist_of_f0(src, normalize_hist=True))
        ht = torch.tensor(E.cal_hist_of_f0(tgt, normalize_hist=True, scale_factor=T / S))
        cost = E.cal_hist_dist(hs, ht, src, tgt).T.cpu().numpy()                  # [T, S], what align_from_distances gets
        dtw = time_warp(cost)
        al = np.asarray(align_from_distances(cost), dtype=np.int64)
        out, al2 = E.EHSADTW(src, tgt, src)
        assert list(al) == list(al2)
        res.update({f"p{p}.src": src, f"p{p}.tgt": tgt, f"p{p}.hist_src": hs.numpy(), f"p{p}.hist_tgt": ht.numpy(),
                    f"p{p}.cost": cost, f"p{p}.dtw": dtw, f"p{p}.align": al, f"p{p}.aligned": np.asarray(out)})
        print(f"pair {p}: S={S} T={T} cost {cost.dtype} dtw[-1,-1]={dtw[-1, -1]:.6f}")
    res["n"] = np.array(len(pairs()))
    sys.path.insert(0, HERE)
    from detnpz import savez_det
    savez_det(os.path.join(HERE, "dtw_ref.npz"), **res)
    print("written", os.path.getsize(os.path.join(HERE, "dtw_ref.npz")), "bytes")


if __name__ == "__main__":
    main()

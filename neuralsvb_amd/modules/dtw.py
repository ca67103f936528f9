"""Shape-aware F0 DTW of the binarizer on the GPU.

Drop-in for the reference's `EHSADTW(src, tgt, input) -> (input[alignment], alignment)`
(modules/voice_conversion/dtw/enhance_sadtw.py:104-113, called by data_gen/singing/binarize_para.py:168-185 to produce
`a2p_f0_alignment`) plus a batched form: the reference aligns one pair per call in interpreted Python / numba; a binarizer
shard hands a few hundred pairs to `ehsadtw_batch` at once (csrc/dtw.hip: slope histograms, chi-square costs, anti-diagonal
DTW sweep, back-tracking -- one workgroup per pair for the sweep).
"""
import numpy as np
import torch

from .. import kernels as K


def ehsadtw_batch(srcs, tgts, device):
    """srcs / tgts: lists of 1-D F0 tracks in Hz (0 = unvoiced), amateur / professional.  -> list of int64 numpy alignments, one
    per pair: for every professional frame the matched amateur frame."""
    assert len(srcs) == len(tgts) and len(srcs) > 0
    P = len(srcs)
    la = np.array([len(s) for s in srcs], np.int32)
    lb = np.array([len(t) for t in tgts], np.int32)
    La, Lb = int(la.max()), int(lb.max())
    fa, fb = np.zeros((P, La), np.float64), np.zeros((P, Lb), np.float64)
    for p, (s, t) in enumerate(zip(srcs, tgts)):
        fa[p, :len(s)] = np.asarray(s, np.float64).reshape(-1)
        fb[p, :len(t)] = np.asarray(t, np.float64).reshape(-1)
    dev = torch.device(device)
    d_la, d_lb = torch.from_numpy(la).to(dev), torch.from_numpy(lb).to(dev)
    ha = K.f0_shape_hist(torch.from_numpy(fa).to(dev), d_la, torch.ones(P, dtype=torch.float64, device=dev))
    hb = K.f0_shape_hist(torch.from_numpy(fb).to(dev), d_lb, torch.from_numpy(lb.astype(np.float64) / la).to(dev))
    align = K.dtw_align(K.hist_cost(ha, d_la, hb, d_lb), d_lb, d_la).cpu().numpy()
    return [align[p, :lb[p]].copy() for p in range(P)]


def EHSADTW(src, tgt, input, device="cuda"):
    """One pair, the reference's signature: src [S] / tgt [T] F0 tracks, input [S, ...] -> (input[alignment], alignment)."""
    al = ehsadtw_batch([np.asarray(src).reshape(-1)], [np.asarray(tgt).reshape(-1)], device)[0]
    return input[al], list(int(a) for a in al)

"""SURVEY 8f4 -- the binarizer's shape-aware F0 DTW (reference modules/voice_conversion/dtw/enhance_sadtw.py:18-113 +
dtw/align.py:8-37) on the GPU, against goldens recorded from the unmodified reference (tests/golden/make_dtw_golden.py).

Slope histograms: exact (integer counts, one fp64 division, one rounding).  Cost matrix: fp32, 48-term sums in a different
order than torch's vectorised reduction -> 2e-6 relative.  Accumulated costs and alignment: BIT-EXACT when fed the reference's
own cost matrix (fp32 additions and minima only); end to end the alignment may differ from the reference only where two
paths tie within the cost tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import dtw_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dtw_ref.npz"))
N = int(G["n"])


def test_oracle_restatement_matches_reference_golden():
    for p in range(N):
        src, tgt = G[f"p{p}.src"], G[f"p{p}.tgt"]
        assert np.array_equal(dtw_ref.shape_hist(src), G[f"p{p}.hist_src"])
        assert np.array_equal(dtw_ref.shape_hist(tgt, len(tgt) / len(src)), G[f"p{p}.hist_tgt"])
        cost = dtw_ref.hist_cost(G[f"p{p}.hist_src"], G[f"p{p}.hist_tgt"])
        assert np.abs(cost - G[f"p{p}.cost"]).max() <= 2e-6 * np.abs(G[f"p{p}.cost"]).max()
        D = dtw_ref.time_warp(G[f"p{p}.cost"])
        assert D.dtype == np.float32 and np.array_equal(D, G[f"p{p}.dtw"])
        assert np.array_equal(dtw_ref.backtrack(D), G[f"p{p}.align"])


def _dev_lens(pairs, dev):
    la = torch.tensor([len(s) for s, _ in pairs], dtype=torch.int32).to(dev)
    lb = torch.tensor([len(t) for _, t in pairs], dtype=torch.int32).to(dev)
    return la, lb


def test_shape_histograms_and_costs(dev):
    from neuralsvb_amd import kernels as K
    pairs = [(G[f"p{p}.src"], G[f"p{p}.tgt"]) for p in range(N)]
    la, lb = _dev_lens(pairs, dev)
    La, Lb = int(la.max()), int(lb.max())
    fa, fb = np.zeros((N, La)), np.zeros((N, Lb))
    for p, (s, t) in enumerate(pairs):
        fa[p, :len(s)], fb[p, :len(t)] = s, t
    ha = K.f0_shape_hist(torch.from_numpy(fa).to(dev), la, torch.ones(N, dtype=torch.float64).to(dev))
    sc = torch.tensor([len(t) / len(s) for s, t in pairs], dtype=torch.float64).to(dev)
    hb = K.f0_shape_hist(torch.from_numpy(fb).to(dev), lb, sc)
    cost = K.hist_cost(ha, la, hb, lb).cpu().numpy()
    for p, (s, t) in enumerate(pairs):
        assert np.array_equal(ha[p, :len(s)].cpu().numpy(), G[f"p{p}.hist_src"]), p
        assert np.array_equal(hb[p, :len(t)].cpu().numpy(), G[f"p{p}.hist_tgt"]), p
        assert float(ha[p, len(s):].abs().sum()) == 0.0
        ref = G[f"p{p}.cost"]
        assert np.abs(cost[p, :len(t), :len(s)] - ref).max() <= 2e-6 * np.abs(ref).max(), p


def test_dtw_sweep_and_backtrack_bit_exact_on_reference_costs(dev):
    from neuralsvb_amd import kernels as K
    pairs = [(G[f"p{p}.src"], G[f"p{p}.tgt"]) for p in range(N)]
    la, lb = _dev_lens(pairs, dev)
    La, Lb = int(la.max()), int(lb.max())
    cost = np.zeros((N, Lb, La), np.float32)
    for p, (s, t) in enumerate(pairs):
        cost[p, :len(t), :len(s)] = G[f"p{p}.cost"]
    align, dtw = K.dtw_align(torch.from_numpy(cost).to(dev), lb, la, want_dtw=True)
    for p, (s, t) in enumerate(pairs):
        assert np.array_equal(dtw[p, :len(t), :len(s)].cpu().numpy(), G[f"p{p}.dtw"]), p
        assert np.array_equal(align[p, :len(t)].cpu().numpy(), G[f"p{p}.align"]), p


def test_ehsadtw_end_to_end(dev):
    """The drop-in call and the batched call: alignments equal the reference's except at cost ties (none in these pairs)."""
    from neuralsvb_amd.modules import dtw
    pairs = [(G[f"p{p}.src"], G[f"p{p}.tgt"]) for p in range(N)]
    als = dtw.ehsadtw_batch([s for s, _ in pairs], [t for _, t in pairs], dev)
    for p, al in enumerate(als):
        ref = G[f"p{p}.align"]
        assert al.shape == ref.shape and (al != ref).mean() <= 0.02, (p, int((al != ref).sum()))
    out, al0 = dtw.EHSADTW(pairs[0][0], pairs[0][1], pairs[0][0], device=dev)
    assert list(al0) == list(als[0]) and np.array_equal(out, pairs[0][0][als[0]])

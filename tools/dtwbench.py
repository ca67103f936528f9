"""Time the batched shape-aware F0 DTW (GPU) at binarizer-like sizes, next to the CPU restatement on one pair.
python tools/dtwbench.py [pairs frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd.modules import dtw  # noqa: E402
from oracle import dtw_ref  # noqa: E402

P, T = [int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (64, 1124))]
rng = np.random.RandomState(0)
srcs, tgts = [], []
for p in range(P):
    S_, T_ = T - rng.randint(0, 40), T - rng.randint(0, 40)
    s = 220 * 2 ** (0.3 * np.sin(np.arange(S_) / 11.0) + 0.02 * rng.randn(S_))
    t = 225 * 2 ** (0.3 * np.sin(np.arange(T_) * S_ / T_ / 11.0 + 0.2) + 0.01 * rng.randn(T_))
    s[rng.rand(S_) < 0.1] = 0
    t[rng.rand(T_) < 0.1] = 0
    srcs.append(s)
    tgts.append(t)
dev = torch.device("cuda:0")
dtw.ehsadtw_batch(srcs[:2], tgts[:2], dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
als = dtw.ehsadtw_batch(srcs, tgts, dev)
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
t0 = time.perf_counter()
ref = dtw_ref.ehsadtw(srcs[0], tgts[0])
t_cpu = time.perf_counter() - t0
print(f"{P} pairs x ~{T} frames: GPU {t_gpu * 1e3:.1f} ms total = {t_gpu / P * 1e3:.2f} ms/pair; CPU restatement (numpy + Python DP) "
      f"{t_cpu:.2f} s for one pair; alignment of pair 0 differs from the CPU result in {(als[0] != ref).mean() * 100:.2f} % of frames")

"""Where do the cycles of one bf16x3 conv workgroup go?  (GPU only; uses the svb_debug_set_timing_buffer hook.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd import _lib as L  # noqa: E402
from neuralsvb_amd import kernels as K  # noqa: E402

B, Cin, Cout, T, k, cfg = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (32, 192, 384, 1124, 5, 2))]
USEQ = len(sys.argv) > 7 and sys.argv[7] == "q"
dev = torch.device("cuda:0")
x = torch.randn(B, Cin, T, device=dev)
w = torch.randn(Cout, Cin, k, device=dev) * 0.05
qa, _ = K.weight_pack_q(w, None, 1)
pad = (k - 1) // 2
xq = K.split_q(x) if USEQ else None
for _ in range(3):
    K.conv1d_forward(x, qa, Cout, k, 1, pad, 1, 1, force_cfg=cfg, x_q=xq)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    K.conv1d_forward(x, qa, Cout, k, 1, pad, 1, 1, force_cfg=cfg, x_q=xq)
e1.record()
torch.cuda.synchronize()
KERNEL_US = e0.elapsed_time(e1) * 1e3 / 20
buf = torch.zeros(64 * 32 * 8, dtype=torch.int64, device=dev)
lib = L.get_lib()
lib.svb_debug_set_timing_buffer(buf.data_ptr())
K.conv1d_forward(x, qa, Cout, k, 1, pad, 1, 1, force_cfg=cfg, x_q=xq)
torch.cuda.synchronize()
lib.svb_debug_set_timing_buffer(None)
t = buf.cpu().numpy().reshape(64, 32, 8).astype(np.int64)
SVBQ_LAST = 31
fast = not t[:, :SVBQ_LAST, 3].any()          # the straight-line loop stamps 0 1 2 (4 5): slot 3 stays empty
if fast:
    names = ["wait weight frags + issue x loads", "compute (MFMA loop)", "store next tile", "barrier"]
else:
    names = ["issue loads", "compute (MFMA loop)", "barrier 1", "stage to LDS", "barrier 2"]
rows = []
for blk in range(64):
    for st in range(SVBQ_LAST):
        s = t[blk, st]
        if fast and s[0] and s[2]:
            rows.append([s[1] - s[0], s[2] - s[1], (s[4] - s[2]) if s[4] else 0, (s[5] - s[4]) if s[5] else 0])
        elif not fast and s[0] and s[5]:
            rows.append([s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4]])
rows = np.array(rows)
print(f"shape B{B} {Cin}->{Cout} k{k} T{T} cfg {cfg} ({'straight-line' if fast else 'generic'} loop): {len(rows)} (block, stage) samples; "
      f"cycles mean / median")
for i, n in enumerate(names):
    print(f"  {n:36s} {rows[:, i].mean():9.0f} {np.median(rows[:, i]):9.0f}")
print(f"  {'stage total':36s} {rows.sum(1).mean():9.0f}")
pro = [(t[b, 0, 7] - t[b, 0, 6]) for b in range(64) if t[b, 0, 7] and t[b, 0, 6]]
if pro:
    print(f"  prologue (first tiles)  {np.median(pro):9.0f}")
epi = [(t[b, 31, 7] - t[b, 31, 6]) for b in range(64) if t[b, 31, 7] and t[b, 31, 6]]
if epi:
    print(f"  epilogue (stores)       {np.median(epi):9.0f}")
tot = [(t[b, 31, 7] - t[b, 0, 6]) for b in range(64) if t[b, 31, 7] and t[b, 0, 6]]
if tot:
    print(f"  workgroup total         {np.median(tot):9.0f}")
    span = max(t[b, 31, 7] for b in range(64) if t[b, 31, 7]) - min(t[b, 0, 6] for b in range(64) if t[b, 0, 6])
    print(f"  first start .. last end of the sampled workgroups {span}; kernel {KERNEL_US:.1f} us "
          f"({2.0 * B * Cout * T * Cin * k / KERNEL_US / 1e6:.1f} TFLOP/s)")

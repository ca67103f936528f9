"""Time the fused rel-pos attention kernel at the PPG encoder's shape (GPU only).   python tools/attnbench.py [B H T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd import kernels as K  # noqa: E402

B, H, T = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 4, 562))]
dk = 64
dev = torch.device("cuda:0")
q, k, v = (torch.randn(B, H * dk, T, device=dev) for _ in range(3))
pu = torch.randn(H, dk, device=dev)
bd = torch.randn(B, H, T, T, device=dev)
keep = torch.ones(B, T, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


t = timeit(lambda: K.relpos_attention(q, k, v, pu, bd, keep, 0.125, H))
fl = 2 * 2.0 * B * H * T * T * dk
print(f"relpos_attention B{B} H{H} T{T}: {t:.1f} us  ({fl / t / 1e6:.1f} TFLOP/s algorithmic, bd read {B * H * T * T * 4 / t / 1e6:.2f} TB/s)")
q4 = q.view(B, H, dk, T)
t2 = timeit(lambda: torch.matmul(q4.transpose(-1, -2), q4))
print(f"rocBLAS fp32 q^T k ([B,H,T,T] out): {t2:.1f} us")
pv = torch.randn(H, dk, device=dev)
p = torch.randn(1, H * dk, T, device=dev)
pt_hi, pt_lo = K.relpos_pos_table(p, H)
t3 = timeit(lambda: K.relpos_attention_pos(q, k, v, pu, pv, pt_hi, pt_lo, keep, 0.125, H))
print(f"relpos_attention_pos (position scores in the kernel): {t3:.1f} us  ({3 * 2.0 * B * H * T * T * dk / t3 / 1e6:.1f} TFLOP/s algorithmic incl. the "
      f"position product)   vs  {t:.1f} + {t2:.1f} us for the bd form + the GEMM that writes bd")

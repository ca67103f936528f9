"""Weight-gradient kernels of the step's shapes, one launch sequence each (split-K kernel + reduce), timed on the GPU.
`SVB_WG_NO_XCD=1` with the instrumentation library (SVB_LIB=instr) switches the XCD-aware work ids off for an A/B.

  python tools/wgbench.py [--iters N]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralsvb_amd import _lib  # noqa: E402

if os.environ.get("SVB_LIB") == "instr":
    _lib.LIB_PATH = os.path.join(ROOT, "neuralsvb_amd", "libsvb_hip_instr.so")
from neuralsvb_amd import kernels as K  # noqa: E402

# (B, Cout = rows of dy, Cin = rows of x, T, k)
SHAPES = [(32, 256, 256, 1124, 5), (32, 384, 192, 1124, 5), (32, 192, 384, 1124, 5), (32, 384, 192, 281, 5),
          (32, 192, 384, 281, 5), (32, 1536, 256, 1124, 1), (32, 256, 1536, 1124, 1), (32, 3072, 256, 281, 1),
          (32, 256, 768, 1124, 1), (32, 256, 256, 1124, 1), (32, 384, 192, 1124, 1), (32, 384, 192, 281, 1),
          (32, 256, 256, 562, 5), (64, 256, 256, 512, 7), (16, 256, 256, 1124, 5)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    print(f"{'shape':34s} {'us':>8s} {'TF':>7s}   (kernel + reduce; checksum)")
    for B, ca, cb, T, k in SHAPES:
        dy = torch.randn(B, ca, T, generator=g).to(dev)
        x = torch.randn(B, cb, T, generator=g).to(dev)

        def run():
            return K.conv1d_wgrad(dy, x, k, 1, k // 2, 1, 1, bf16x3=True)
        out = run()
        dw = out[0] if isinstance(out, (tuple, list)) else out
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        tf = 2.0 * B * T * ca * cb * k / us / 1e6
        print(f"wgrad B{B} {ca}x{cb} k{k} T{T:<6d}      {us:8.1f} {tf:7.1f}   {float(dw.double().abs().sum()):.6e}")


if __name__ == "__main__":
    main()

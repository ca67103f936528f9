"""Pointwise (1-tap) conv shapes of the train step / inference pipeline: every tile configuration of the bf16x3 family back to back
(interleaved repetitions, HIP events per launch, median), the committed table's choice marked, and a bit-identity check of the
pointwise GEMM kernel (csrc/conv1d_pw.hip, configurations 18..23) against a tap-table tile on the same inputs (GPU only)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd import kernels as K  # noqa: E402

SHAPES = [  # (op, B, Cin, Cout, T)
    ("fwd", 32, 256, 256, 562), ("fwd", 32, 192, 384, 281), ("fwd", 32, 1024, 256, 562), ("fwd", 32, 256, 1024, 562),
    ("fwd", 32, 256, 768, 562), ("fwd", 32, 256, 512, 562), ("fwd", 32, 192, 384, 1124), ("fwd", 32, 256, 1536, 1124),
    ("fwd", 32, 256, 3072, 281), ("fwd", 32, 768, 256, 1124), ("fwd", 32, 256, 256, 1124), ("fwd", 32, 192, 192, 1124),
    ("convT", 32, 384, 192, 281), ("convT", 32, 384, 192, 1124), ("convT", 32, 1536, 256, 1124), ("convT", 32, 3072, 256, 281),
    ("convT", 32, 256, 768, 1124), ("convT", 32, 256, 256, 1124),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--cfgs", default="2,3,11,12,13,18,19,20,21,22,23")
    ap.add_argument("--epilogue", action="store_true", help="bias + residual + mask (the conformer / res-skip form)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfgs = [int(c) for c in a.cfgs.split(",")]
    print(f"{'shape':38s} table " + " ".join(f"{c:>7d}" for c in cfgs) + "   best   TF   (us per launch)")
    for op, B, cin, cout, T in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout)
        x = torch.randn(B, cin, T, generator=g).to(dev)
        if op == "fwd":
            w = (torch.randn(cout, cin, 1, generator=g) * 0.05).to(dev)
            pk = K.weight_pack_q(w, None, 1)[0]
            sig = ("qf", B, cin, cout, 1, T, 1, 1, 0, 1, False)
        else:
            w = (torch.randn(cin, cout, 1, generator=g) * 0.05).to(dev)
            pk = K.weight_pack_q(w, None, 1)[1]
            sig = ("qt", B, cin, cout, 1, T, T, 1, 1, 0, 1, False)
        epi = {}
        if a.epilogue:
            epi = dict(bias=torch.randn(cout, device=dev), residual=torch.randn(B, cout, T, device=dev),
                       mask=(torch.rand(B, T, device=dev) > 0.1).float())
        y = torch.empty(B, cout, T, device=dev)

        def run(cfg):
            if op == "fwd":
                return K.conv1d_forward(x, pk, cout, 1, 1, 0, 1, 1, out=y, force_cfg=cfg, **epi)
            return K.conv1d_transposed(x, pk, cout, T, 1, 1, 0, 1, 1, out=y, force_cfg=cfg, **epi)
        ref = run(2).clone()
        bad = []
        for c in cfgs:
            for _ in range(3):
                if not torch.equal(run(c), ref):
                    bad.append(c)
                    break
        ev = {c: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)] for c in cfgs}
        for c in cfgs:
            run(c)
        for r in range(a.reps):
            for c in cfgs:
                e0, e1 = ev[c][r]
                e0.record()
                run(c)
                e1.record()
        torch.cuda.synchronize()
        med = {}
        for c in cfgs:
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev[c])
            med[c] = ts[len(ts) // 2] * 1e3
        best = min(cfgs, key=lambda c: med[c])
        fl = 2.0 * B * cout * T * cin
        tab = K._TUNED.get(sig, 0)
        print(f"{str((op, B, cin, cout, T)):38s} {tab:5d} " + " ".join(f"{med[c]:7.1f}" for c in cfgs) +
              f"   {best:4d} {fl / med[best] / 1e6:6.1f}" + (f"   NOT BIT-IDENTICAL: {bad}" if bad else ""))


if __name__ == "__main__":
    main()

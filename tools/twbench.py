"""Back-to-back timing of every bf16x3 conv tile configuration (12 tiles of conv1d_bf16.hip + the 3 tile-walking variants of
conv1d_tw.hip) on the train step's dominant stride-1 shapes (GPU only), plus a repeatability / parity check of the
tile-walking variants against a conv1d_bf16.hip tile (races in the LDS-DMA pipeline would show here, not on the emulator)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd import kernels as K  # noqa: E402

SHAPES = [
    # op, B, Cin, Cout, T, k
    ("fwd", 32, 192, 384, 1124, 5), ("fwd", 32, 192, 384, 281, 5), ("convT", 32, 384, 192, 1124, 5), ("convT", 32, 384, 192, 281, 5),
    ("fwd", 32, 256, 256, 1124, 5), ("convT", 32, 256, 256, 1124, 5), ("fwd", 32, 256, 256, 562, 1), ("fwd", 32, 1024, 256, 562, 1),
    ("fwd", 32, 256, 1024, 562, 1), ("fwd", 32, 192, 384, 1124, 1), ("fwd", 32, 192, 384, 281, 1), ("convT", 32, 384, 192, 281, 1),
    ("convT", 32, 384, 192, 1124, 1), ("fwd", 32, 256, 1536, 1124, 1), ("fwd", 32, 768, 256, 1124, 1), ("fwd", 32, 256, 3072, 281, 1),
    ("convT", 32, 1536, 256, 1124, 1), ("fwd", 32, 256, 256, 562, 5), ("fwd", 16, 192, 384, 1124, 5), ("fwd", 64, 256, 256, 512, 7),
]


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cfgs", default="1-15")
    a = ap.parse_args()
    lo, hi = [int(v) for v in a.cfgs.split("-")]
    cfgs = list(range(lo, hi + 1))
    dev = torch.device("cuda:0")
    print("shape".ljust(36) + " ".join(f"{c:>6d}" for c in cfgs) + "   best_old best_tw  (us; TF of the best)")
    for op, B, ca, cb, T, k in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn(B, ca, T, generator=g).to(dev)
        pad = (k - 1) // 2
        if op == "fwd":
            w = (torch.randn(cb, ca, k, generator=g) * 0.05).to(dev)
            pk = K.weight_pack_q(w, None, 1)[0]
            fn = lambda cfg: K.conv1d_forward(x, pk, cb, k, 1, pad, 1, 1, force_cfg=cfg)
        else:
            w = (torch.randn(ca, cb, k, generator=g) * 0.05).to(dev)
            pk = K.weight_pack_q(w, None, 1)[1]
            fn = lambda cfg: K.conv1d_transposed(x, pk, cb, T, k, 1, pad, 1, 1, force_cfg=cfg)
        flops = 2.0 * B * ca * cb * T * k
        ref = fn(2)
        times = {}
        bad = []
        for c in cfgs:
            try:
                y = fn(c)
            except Exception as e:  # noqa: BLE001
                times[c] = float("nan")
                continue
            if c >= 13:
                err = ((y - ref).abs().max() / ref.abs().max()).item()
                y2 = fn(c)
                same = bool(torch.equal(y, y2))
                if err > 2e-5 or not same:
                    bad.append((c, err, same))
            times[c] = timeit(lambda: fn(c), a.iters)
        old = min((times[c], c) for c in cfgs if c <= 12 and times[c] == times[c])
        tw = min([(times[c], c) for c in cfgs if c >= 13 and times[c] == times[c]] or [(float("nan"), 0)])
        name = f"{op} B{B} {ca}->{cb} k{k} T{T}"
        print(name.ljust(36) + " ".join(f"{times[c] * 1e6:6.1f}" for c in cfgs) +
              f"   {old[0] * 1e6:6.1f}@{old[1]:<2d} {tw[0] * 1e6:6.1f}@{tw[1]:<2d}  {flops / old[0] / 1e12:5.0f} {flops / tw[0] / 1e12:5.0f} TF"
              + (f"  MISMATCH {bad}" if bad else ""), flush=True)


if __name__ == "__main__":
    main()

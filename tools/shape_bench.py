"""Per-shape cost of the conv family inside the train step (GPU only): collect every (op, shape) the step launches,
then time each distinct one back-to-back (GPU-bound, unlike events around single launches issued from Python)."""
import argparse
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuralsvb_amd import kernels as K  # noqa: E402


def timeit(fn, iters=10):
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
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    args = argparse.Namespace(gpus=1, steps=1, warmup=1, batch=16, seconds=6.0, sample_rate=24000, bf16=False,
                              precision=a.precision, graph=False)
    dev = torch.device("cuda:0")
    with tempfile.TemporaryDirectory() as tmp:
        task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp)
        bench.run_steps(trainer, task, batch, 2, 1)
        K.PROFILE = []
        bench.run_steps(trainer, task, batch, 1, 3)
        torch.cuda.synchronize()
        rec, K.PROFILE = K.PROFILE, None
    q = a.precision == "bf16x3"
    counts = {}
    for name, flops, e0, e1, tag in rec:
        counts[tag] = counts.get(tag, 0) + 1
    rows = []
    for tag, cnt in counts.items():
        op, B, ca, cb, G, T, k, s, dil = tag
        pad = dil * (k - 1) // 2
        if op == "fwd":
            x = torch.randn(B, ca, T, device=dev)
            w = torch.randn(cb, ca // G, k, device=dev) * 0.05
            pk = (K.weight_pack_q(w, None, G) if q else K.weight_pack(w))[0]
            fn = lambda: K.conv1d_forward(x, pk, cb, k, s, pad, dil, G)
            tout = K.conv_out_len(T, k, s, pad, dil)
            fl = 2.0 * B * cb * tout * (ca // G) * k
        elif op == "convT":
            x = torch.randn(B, ca, T, device=dev)
            w = torch.randn(ca, cb // G, k, device=dev) * 0.05       # ConvTranspose1d weight [Cin, Cout/G, k]
            pk = (K.weight_pack_q(w, None, G) if q else K.weight_pack(w))[1]
            tout = (T - 1) * s - 2 * pad + dil * (k - 1) + 1
            fn = lambda: K.conv1d_transposed(x, pk, cb, tout, k, s, pad, dil, G)
            fl = 2.0 * B * ca * T * (cb // G) * k
        else:
            A = torch.randn(B, ca, T, device=dev)
            tb = (T - 1) * s + dil * (k - 1) + 1 - 2 * pad
            Bt = torch.randn(B, cb, max(tb, 1), device=dev)
            fn = lambda: K.conv1d_wgrad(A, Bt, k, s, pad, dil, G, bf16x3=q)
            fl = 2.0 * B * ca * T * (cb // G) * k
        try:
            t = timeit(fn)
        except Exception as e:  # noqa: BLE001
            print("skip", tag, e)
            continue
        rows.append((cnt * t, cnt, t, fl / t / 1e12, tag))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"total conv-family time {tot * 1e3:.2f} ms/step over {sum(r[1] for r in rows)} launches")
    for ct, cnt, t, tf, tag in rows[:a.top]:
        print(f"  {ct * 1e3:7.3f} ms  {cnt:4d} x {t * 1e6:7.1f} us  {tf:6.1f} TF  {tag}")


if __name__ == "__main__":
    main()

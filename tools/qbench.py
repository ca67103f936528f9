"""bf16x3 conv with fp32 input (split while staging) vs pre-split Q input, per tile configuration (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd import kernels as K  # noqa: E402

SHAPES = [  # name, B, Cin, Cout, T, k
    ("dec_in 192->384 k5 T1124", 32, 192, 384, 1124, 5),
    ("enc_in 192->384 k5 T281", 32, 192, 384, 281, 5),
    ("dec_rs 192->384 k1 T1124", 32, 192, 384, 1124, 1),
    ("enc_rs 192->384 k1 T281", 32, 192, 384, 281, 1),
    ("pitch 256->256 k5 T1124", 32, 256, 256, 1124, 5),
]
NAMES = ["64x128", "128x96", "128x128", "64x64", "32x128", "64x192", "64x256"]


def timeit(fn, iters=20):
    for _ in range(3):
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
    dev = torch.device("cuda:0")
    K.AUTOTUNE = False
    for name, B, cin, cout, T, k in SHAPES:
        x = torch.randn(B, cin, T, device=dev)
        w = torch.randn(cout, cin, k, device=dev) * 0.05
        qa, qb = K.weight_pack_q(w, None, 1)
        xq = K.split_q(x)
        y = torch.empty(B, cout, T, device=dev)
        fl = 2.0 * B * cout * T * cin * k
        t_split = timeit(lambda: K.split_q(x))
        print(f"{name}: split_q {t_split * 1e6:.1f} us ({x.numel() * 8 / t_split / 1e12:.2f} TB/s)")
        for cfg in range(1, 8):
            try:
                t0 = timeit(lambda: K.conv1d_forward(x, qa, cout, k, 1, k // 2, 1, 1, out=y, force_cfg=cfg))
                t1 = timeit(lambda: K.conv1d_forward(x, qa, cout, k, 1, k // 2, 1, 1, out=y, force_cfg=cfg, x_q=xq))
            except Exception as e:  # noqa: BLE001
                print(f"   {NAMES[cfg - 1]:8s} skipped: {e}")
                continue
            print(f"   {NAMES[cfg - 1]:8s} fp32-in {t0 * 1e6:7.1f} us {fl / t0 / 1e12:6.1f} TF   Q-in {t1 * 1e6:7.1f} us {fl / t1 / 1e12:6.1f} TF   x{t0 / t1:.2f}")


if __name__ == "__main__":
    main()

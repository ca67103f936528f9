"""Standalone bandwidth of the step's streaming kernels at the step's shapes (GPU only): us per call and fp32 bytes moved / time.
Durations in a rocprofv3 table of the benchmarked configuration are inflated by the kernels of the other streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd import _lib  # noqa: E402

if os.environ.get("SVB_LIB_FILE"):           # (experiments: a variant build of the library, e.g. another SSIM tile height)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["SVB_LIB_FILE"])
from neuralsvb_amd import kernels as K  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=30):
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


def show(name, t, nbytes):
    print(f"{name:44s} {t * 1e6:8.1f} us  {nbytes / t / 1e12:6.2f} TB/s")


for T in (1124, 281):
    B, C = 32, 192
    xin = torch.randn(B, 2 * C, T, device=dev)
    G = torch.randn(B, 2 * C * 4, T, device=dev)
    dacts = torch.randn(B, C, T, device=dev)
    x = torch.randn(B, C, T, device=dev)
    rs = torch.randn(B, 2 * C, T, device=dev)
    out = torch.randn(B, C, T, device=dev)
    mask = torch.ones(B, T, device=dev)
    e = C * B * T * 4
    show(f"T{T} gate_fwd", timeit(lambda: K.wn_gate_fwd(xin, G, 0)), 5 * e)
    show(f"T{T} gate_bwd", timeit(lambda: K.wn_gate_bwd(xin, G, dacts, 0)), 7 * e)
    show(f"T{T} res_skip", timeit(lambda: K.wn_res_skip(x, rs, mask, out, False)), 6 * e)
    show(f"T{T} res_skip_bwd", timeit(lambda: K.wn_res_skip_bwd(x, out, mask)), 5 * e)
for (B, C, T, Gn) in ((32, 256, 1124, 8), (32, 256, 1124, 16), (32, 256, 562, 8)):
    h = torch.randn(B, C, T, device=dev)
    res = torch.randn(B, C, T, device=dev)
    gm, bt = torch.randn(C, device=dev), torch.randn(C, device=dev)
    e = B * C * T * 4
    y, stats = K.gn_relu_fwd(h, res, gm, bt, Gn, 1e-5)
    show(f"gn_relu_fwd C{C} T{T} G{Gn} (+res)", timeit(lambda: K.gn_relu_fwd(h, res, gm, bt, Gn, 1e-5)), 3 * e)
    show(f"gn_relu_bwd C{C} T{T} G{Gn}", timeit(lambda: K.gn_relu_bwd(y, h, gm, bt, stats, Gn)), 3 * e)
    show(f"layernorm_nct_fwd C{C} T{T}", timeit(lambda: K.layernorm_nct_fwd(h, gm, bt)), 2 * e)
    a = torch.empty_like(h)
    show(f"torch copy C{C} T{T}", timeit(lambda: a.copy_(h)), 2 * e)
    show(f"torch add C{C} T{T}", timeit(lambda: torch.add(h, res, out=a)), 3 * e)
# fused mel loss (masked L1 + SSIM) at the step's shape: pred read through the decoder's [B,80,T] strides
B, T, Fb = 32, 1124, 80
pred = torch.randn(B, Fb, T, device=dev).transpose(1, 2)
tgt = torch.randn(B, T, Fb, device=dev)
e = B * T * Fb * 4
show("mel_loss_fwd [32,1124,80]", timeit(lambda: K.mel_loss_fwd(pred, tgt, 6.0, 3)), 2 * e)
gout = torch.ones(3, device=dev)
sums_t = K.mel_loss_fwd(pred, tgt, 6.0, 3)
show("mel_loss_bwd [32,1124,80]", timeit(lambda: K.mel_loss_bwd(pred, tgt, gout, sums_t, 6.0, 3)), 9 * e)
for terms in (1, 2):
    show(f"mel_loss_fwd terms={terms}", timeit(lambda: K.mel_loss_fwd(pred, tgt, 6.0, terms)), 2 * e)
    show(f"mel_loss_bwd terms={terms}", timeit(lambda: K.mel_loss_bwd(pred, tgt, gout, sums_t, 6.0, terms)), 9 * e)
predc = pred.contiguous()
show("mel_loss_fwd contiguous pred", timeit(lambda: K.mel_loss_fwd(predc, tgt, 6.0, 3)), 2 * e)
show("ssim_fwd", timeit(lambda: K.ssim_fwd(pred, tgt, 6.0)), 3 * e)

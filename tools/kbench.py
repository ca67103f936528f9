"""Kernel micro-benchmark at hot-path shapes (GPU only).  Prints achieved TFLOP/s per conv kernel against the
fp32-MFMA peak (157.3 TF) and, for perspective, the same op through torch/MIOpen on the same GPU."""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd import kernels as K  # noqa: E402

PEAK_F32 = 157.3e12

SHAPES = [
    # name, B, Cin, Cout, G, T, k, s, pad, dil
    ("vae_dec_wn_in  192->384 k5 T1124", 16, 192, 384, 1, 1124, 5, 1, 2, 1),
    ("vae_enc_wn_in  192->384 k5 T281", 16, 192, 384, 1, 281, 5, 1, 2, 1),
    ("vae_dec_res    192->384 k1 T1124", 16, 192, 384, 1, 1124, 1, 1, 0, 1),
    ("vae_dec_cond   256->1536 k1 T1124", 16, 256, 1536, 1, 1124, 1, 1, 0, 1),
    ("pitch_enc      256->256 k5 T1124", 16, 256, 256, 1, 1124, 5, 1, 2, 1),
    ("enc_prenet     80->192 k8 s4 T1124", 16, 80, 192, 1, 1124, 8, 4, 2, 1),
    ("cond_proj      768->256 k1 T1124", 16, 768, 256, 1, 1124, 1, 1, 0, 1),
    ("hifi_rb        256->256 k11 d5 L512", 64, 256, 256, 1, 512, 11, 1, 25, 5),
    ("hifi_rb        32->32 k3 L8192", 64, 32, 32, 1, 8192, 3, 1, 1, 1),
    ("msd_g16        256->512 k41 s4 g16 L2048", 64, 256, 512, 16, 2048, 41, 4, 20, 1),
    ("mpd            512->1024 k5 s3 L152 (B*p)", 128, 512, 1024, 1, 152, 5, 3, 2, 1),
]


def timeit(fn, iters):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-torch", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = SHAPES[:5] if args.quick else SHAPES
    rows = []
    for name, B, Cin, Cout, G, T, k, s, pad, dil in shapes:
        x = torch.randn(B, Cin, T, device=dev)
        w = torch.randn(Cout, Cin // G, k, device=dev) * 0.05
        bias = torch.randn(Cout, device=dev)
        Tout = K.conv_out_len(T, k, s, pad, dil)
        dy = torch.randn(B, Cout, Tout, device=dev)
        flops = 2.0 * B * Cout * Tout * (Cin // G) * k
        pa, pb = K.weight_pack(w)
        y = K.conv1d_forward(x, pa, Cout, k, s, pad, dil, G, bias=bias)
        ref = F.conv1d(x, w, bias, s, pad, dil, G)
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        t_f = timeit(lambda: K.conv1d_forward(x, pa, Cout, k, s, pad, dil, G, bias=bias), args.iters)
        t_d = timeit(lambda: K.conv1d_transposed(dy, pb, Cin, T, k, s, pad, dil, G), args.iters)
        t_w = timeit(lambda: K.conv1d_wgrad(dy, x, k, s, pad, dil, G), args.iters)
        t_p = timeit(lambda: K.weight_pack(w), args.iters)
        qa, qb = K.weight_pack_q(w, None, G)
        yq = K.conv1d_forward(x, qa, Cout, k, s, pad, dil, G, bias=bias)
        errq = ((yq - ref).abs().max() / ref.abs().max()).item()
        t_qf = timeit(lambda: K.conv1d_forward(x, qa, Cout, k, s, pad, dil, G, bias=bias), args.iters)
        t_qd = timeit(lambda: K.conv1d_transposed(dy, qb, Cin, T, k, s, pad, dil, G), args.iters)
        t_qw = timeit(lambda: K.conv1d_wgrad(dy, x, k, s, pad, dil, G, bf16x3=True), args.iters)
        row = dict(name=name, gflop=flops / 1e9, err=err, fwd_ms=t_f * 1e3, fwd_tf=flops / t_f / 1e12,
                   dgrad_ms=t_d * 1e3, dgrad_tf=flops / t_d / 1e12, wgrad_ms=t_w * 1e3, wgrad_tf=flops / t_w / 1e12,
                   pack_ms=t_p * 1e3, fwd_frac=flops / t_f / PEAK_F32, q_err=errq, q_fwd_ms=t_qf * 1e3,
                   q_fwd_tf=flops / t_qf / 1e12, q_dgrad_ms=t_qd * 1e3, q_dgrad_tf=flops / t_qd / 1e12,
                   q_wgrad_ms=t_qw * 1e3, q_wgrad_tf=flops / t_qw / 1e12)
        if not args.no_torch:
            xr = x.clone().requires_grad_(True)
            wr = w.clone().requires_grad_(True)
            t_tf = timeit(lambda: F.conv1d(x, w, bias, s, pad, dil, G), args.iters)

            def tb():
                yy = F.conv1d(xr, wr, None, s, pad, dil, G)
                torch.autograd.grad(yy, (xr, wr), dy)
            t_tb = timeit(tb, args.iters) - t_tf
            row.update(torch_fwd_ms=t_tf * 1e3, torch_bwd_ms=t_tb * 1e3)
        rows.append(row)
        print(json.dumps({k_: (float(f"{v:.4g}") if isinstance(v, float) else v) for k_, v in row.items()}), flush=True)
    # streaming kernels: GB/s against 8 TB/s
    B, C, T = 16, 192, 1124
    xin = torch.randn(B, 2 * C, T, device=dev)
    g = torch.randn(B, 2 * C * 4, T, device=dev)
    t = timeit(lambda: K.wn_gate_fwd(xin, g, 2 * C), args.iters)
    byt = (2 * C + 2 * C + C) * B * T * 4
    print(json.dumps(dict(name="wn_gate_fwd", ms=t * 1e3, gbps=byt / t / 1e9, frac_8tbs=byt / t / 8e12)))
    wav = torch.rand(16, 24000 * 6, device=dev) * 2 - 1
    from oracle import frontend as ofe
    basis = torch.from_numpy(ofe.librosa_mel_filterbank(24000, 512, 80, 50, 12000)).to(dev)
    win = torch.hann_window(512, periodic=True, device=dev)
    t = timeit(lambda: K.stft_mel(wav, win, basis, 512, 128, 0, 1e-10), args.iters)
    byt = wav.numel() * 4 + 16 * (1 + 144000 // 128) * 80 * 4
    print(json.dumps(dict(name="stft_mel 16x6s", ms=t * 1e3, gbps=byt / t / 1e9, frac_8tbs=byt / t / 8e12,
                          audio_s_per_s=16 * 6 / t)))
    f0 = torch.rand(64, 64, device=dev) * 300 + 100
    ri = torch.rand(64, 9, device=dev)
    nz = torch.randn(64, 8192, 9, device=dev)
    lw, lb = torch.randn(9, device=dev), torch.randn(1, device=dev)
    t = timeit(lambda: K.nsf_source(f0, ri, nz, lw, lb, 128, 24000.0), args.iters)
    byt = nz.numel() * 4 + 64 * 8192 * 4
    print(json.dumps(dict(name="nsf_source 64x8192", ms=t * 1e3, gbps=byt / t / 1e9, frac_8tbs=byt / t / 8e12)))


if __name__ == "__main__":
    main()

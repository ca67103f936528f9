"""Kernel-level parity: every C-ABI entry point vs the op-level CPU oracle (oracle/ops.py, oracle/frontend.py).

Each test runs twice: `emu` (CPU lane emulator of the same kernel sources -- runs in the GPU-less build
container) and `gpu` (the product: libsvb_hip.so on an MI355X, marked `gpu`).
Tolerances: fp32 conv/GEMM results differ from the oracle only by summation order -> rtol 2e-5 of the
output scale (stated per test); integer outputs are compared exactly.
"""

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from neuralsvb_amd import kernels as K
from oracle import frontend as ofe
from oracle import ops as oops


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


CONV_CASES = [
    # B, Cin, Cout, G, T, k, s, pad, dil
    (2, 8, 16, 1, 50, 5, 1, 2, 1),
    (1, 20, 70, 1, 140, 5, 1, 2, 1),      # ragged channel / time tails
    (2, 6, 10, 2, 37, 3, 2, 1, 1),        # grouped + strided
    (1, 16, 24, 1, 64, 8, 4, 2, 1),       # pre-net shape family: k8 s4 p2 (fs2_vae.py:109-114)
    (1, 12, 12, 1, 90, 7, 1, 9, 3),       # dilated resblock conv (hifigan.py:33-41)
    (1, 32, 32, 4, 70, 41, 4, 20, 1),     # MSD grouped k41 s4 (hifigan.py:264-266)
    (1, 1, 16, 1, 300, 15, 1, 7, 1),      # single input channel (hifigan.py:262)
    (1, 24, 1, 1, 120, 7, 1, 3, 1),       # single output channel (conv_post, hifigan.py:140)
    (2, 40, 40, 1, 33, 1, 1, 0, 1),       # 1x1 conv / Linear
    (3, 5, 3, 1, 29, 3, 3, 1, 1),         # MPD-style stride 3
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d_forward_dgrad_wgrad(dev, case):
    B, Cin, Cout, G, T, k, s, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin // G, k, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = oops.conv1d(xr, wr, br, s, pad, dil, G)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)

    xd, wd, bd, dyd = x.to(dev), w.to(dev), bias.to(dev), dy.to(dev)
    pa, pb = K.weight_pack(wd)
    y = K.conv1d_forward(xd, pa, Cout, k, s, pad, dil, G, bias=bd)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-5
    dx = K.conv1d_transposed(dyd, pb, Cin, T, k, s, pad, dil, G)
    assert rel_err(dx, xr.grad) < 2e-5
    dw = K.conv1d_wgrad(dyd, xd, k, s, pad, dil, G)
    assert dw.shape == w.shape
    assert rel_err(dw, wr.grad) < 2e-5
    db = K.bias_grad(dyd)
    assert rel_err(db, br.grad) < 2e-5


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_conv1d_all_tile_configs(dev, cfg):
    g = torch.Generator().manual_seed(cfg)
    B, Cin, Cout, T, k = 2, 18, 150, 200, 5
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    ref = oops.conv1d(x, w, None, 1, 2, 1, 1)
    pa, pb = K.weight_pack(w.to(dev))
    y = K.conv1d_forward(x.to(dev), pa, Cout, k, 1, 2, 1, 1, force_cfg=cfg)
    assert rel_err(y, ref) < 2e-5
    dy = torch.randn(ref.shape, generator=g)
    dref = torch.autograd.grad(oops.conv1d(x.requires_grad_(True), w, None, 1, 2, 1, 1), x, dy)[0]
    dx = K.conv1d_transposed(dy.to(dev), pb, Cin, T, k, 1, 2, 1, 1, force_cfg=cfg)
    assert rel_err(dx, dref) < 2e-5


def test_conv1d_fused_epilogue(dev):
    """y = mask * (residual + out_gate' * act(conv(lrelu(x)) + bias))  -- the fusions used by the HifiGAN
    resblocks (hifigan.py:54-61) and the masked pre-nets (fs2_vae.py:121-123)."""
    g = torch.Generator().manual_seed(7)
    B, Cin, Cout, T, k, d = 2, 24, 40, 77, 3, 3
    pad = (k * d - d) // 2
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    og = torch.randn(B, Cout, T, generator=g)
    mask = (torch.rand(B, T, generator=g) > 0.3).float()
    for act, fn in ((K.ACT_NONE, lambda v: v), (K.ACT_RELU, torch.relu), (K.ACT_LRELU, lambda v: F.leaky_relu(v, 0.1)),
                    (K.ACT_TANH, torch.tanh)):
        ref = fn(oops.conv1d(F.leaky_relu(x, 0.1), w, bias, 1, pad, d)) * oops.lrelu_gate(og, 0.2)
        ref = (ref + res) * mask[:, None, :]
        pa, _ = K.weight_pack(w.to(dev))
        y = K.conv1d_forward(x.to(dev), pa, Cout, k, 1, pad, d, 1, bias=bias.to(dev), in_gate=x.to(dev), in_slope=0.1,
                             out_act=act, out_slope=0.1, out_gate=og.to(dev), out_gate_slope=0.2,
                             residual=res.to(dev), mask=mask.to(dev))
        assert rel_err(y, ref) < 2e-5, act


CONVT_CASES = [
    # B, Cin, Cout, T, k, s, pad   (decoder pre_net k4 s4: vae_models.py:115-120; HifiGAN ups: hifigan.py:122-125)
    (2, 16, 24, 21, 4, 4, 0),
    (1, 32, 16, 17, 16, 8, 4),
    (1, 16, 8, 30, 8, 4, 2),
    (2, 8, 4, 25, 4, 2, 1),
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose1d(dev, case):
    B, Cin, Cout, T, k, s, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, k, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = oops.conv_transpose1d(xr, wr, bias, s, pad)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    Tout = ref.shape[-1]
    pa, pb = K.weight_pack(w.to(dev))
    y = K.conv1d_transposed(x.to(dev), pb, Cout, Tout, k, s, pad, 1, 1, bias=bias.to(dev))
    assert rel_err(y, ref) < 2e-5
    # data gradient of a transposed conv is an ordinary strided conv with the `pa` pack
    dx = K.conv1d_forward(dy.to(dev), pa, Cin, k, s, pad, 1, 1)
    assert dx.shape == x.shape and rel_err(dx, xr.grad) < 2e-5
    # weight gradient: A = x (q-indexed), B = dy (tap-strided)
    dw = K.conv1d_wgrad(x.to(dev), dy.to(dev), k, s, pad, 1, 1)
    assert rel_err(dw, wr.grad) < 2e-5


def test_weight_norm_pack_and_backward(dev):
    """w = g v/||v|| (fs2_vae.py:48) forward through the pack kernel, (dv, dg) through the reduce kernel."""
    g_ = torch.Generator().manual_seed(11)
    B, Cin, Cout, T, k = 2, 12, 20, 60, 5
    x = torch.randn(B, Cin, T, generator=g_)
    v = (torch.randn(Cout, Cin, k, generator=g_) * 0.3).requires_grad_(True)
    gn = (torch.rand(Cout, 1, 1, generator=g_) + 0.5).requires_grad_(True)
    ref = oops.conv1d(x, oops.weight_norm(v, gn), None, 1, 2)
    dy = torch.randn(ref.shape, generator=g_)
    ref.backward(dy)
    vd, gd = v.detach().to(dev), gn.detach().to(dev)
    pa, _ = K.weight_pack(vd, gd)
    y = K.conv1d_forward(x.to(dev), pa, Cout, k, 1, 2)
    assert rel_err(y, ref) < 2e-5
    dv, dg = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, 2, 1, 1, v=vd, g=gd)
    assert rel_err(dv, v.grad) < 5e-5
    assert rel_err(dg, gn.grad) < 5e-5


def test_wgrad_with_activation_gates(dev):
    """dW of  y = relu(conv(lrelu(x)))  given dy: gates on both operands (a_gate = y, b_gate = x)."""
    g_ = torch.Generator().manual_seed(5)
    B, Cin, Cout, T, k = 2, 10, 14, 45, 3
    x = torch.randn(B, Cin, T, generator=g_)
    w = (torch.randn(Cout, Cin, k, generator=g_) * 0.3).requires_grad_(True)
    b = torch.randn(Cout, generator=g_).requires_grad_(True)
    y = torch.relu(oops.conv1d(F.leaky_relu(x, 0.1), w, b, 1, 1))
    dy = torch.randn(y.shape, generator=g_)
    y.backward(dy)
    dw = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, 1, 1, 1, a_gate=y.detach().to(dev), a_slope=0.0,
                        b_gate=x.to(dev), b_slope=0.1)
    assert rel_err(dw, w.grad) < 2e-5
    db = K.bias_grad(dy.to(dev), y.detach().to(dev), 0.0)
    assert rel_err(db, b.grad) < 2e-5


@pytest.mark.parametrize("T", [36, 37])
def test_wn_gate_and_res_skip(dev, T):
    g_ = torch.Generator().manual_seed(T)
    B, C, L = 2, 6, 3
    xin = torch.randn(B, 2 * C, T, generator=g_).requires_grad_(True)
    gc = torch.randn(B, 2 * C * L, T, generator=g_).requires_grad_(True)
    off = 2 * C
    acts = oops.wn_gate(xin, gc[:, off:off + 2 * C])
    da = torch.randn(acts.shape, generator=g_)
    acts.backward(da)
    a = K.wn_gate_fwd(xin.detach().to(dev), gc.detach().to(dev), off)
    assert rel_err(a, acts) < 1e-5
    dgc = torch.zeros_like(gc).to(dev)
    dxin = K.wn_gate_bwd(xin.detach().to(dev), gc.detach().to(dev), da.to(dev), off, dg=dgc)
    assert rel_err(dxin, xin.grad) < 1e-5
    assert rel_err(dgc, gc.grad) < 1e-5
    # res/skip (fs2_vae.py:83-89) and its backward
    x = torch.randn(B, C, T, generator=g_)
    rs = torch.randn(B, 2 * C, T, generator=g_)
    out = torch.randn(B, C, T, generator=g_)
    mask = (torch.rand(B, T, generator=g_) > 0.3).float()
    xn, on = K.wn_res_skip(x.to(dev), rs.to(dev), mask.to(dev), out.to(dev), last=False)
    assert torch.equal(xn.cpu(), (x + rs[:, :C]) * mask[:, None])
    assert torch.equal(on.cpu(), out + rs[:, C:])
    _, ol = K.wn_res_skip(None, rs[:, :C].contiguous().to(dev), None, None, last=True)
    assert torch.equal(ol.cpu(), rs[:, :C])
    drs, dxm = K.wn_res_skip_bwd(x.to(dev), out.to(dev), mask.to(dev))
    assert torch.equal(drs.cpu(), torch.cat([x * mask[:, None], out], 1))
    assert torch.equal(dxm.cpu(), x * mask[:, None])


@pytest.mark.parametrize("C", [256, 80])
def test_layernorm(dev, C):
    g_ = torch.Generator().manual_seed(C)
    x = (torch.randn(3, 17, C, generator=g_) * 2 + 0.5).requires_grad_(True)
    gamma = torch.randn(C, generator=g_).requires_grad_(True)
    beta = torch.randn(C, generator=g_).requires_grad_(True)
    ref = oops.layernorm(x, gamma, beta)
    dy = torch.randn(ref.shape, generator=g_)
    ref.backward(dy)
    y, mean, rstd = K.layernorm_fwd(x.detach().to(dev), gamma.detach().to(dev), beta.detach().to(dev), 1e-5, True)
    assert (y.cpu() - ref.detach()).abs().max() < 2e-5
    dx, dgm, dbt = K.layernorm_bwd(x.detach().to(dev), gamma.detach().to(dev), dy.to(dev), mean, rstd, n_part=5)
    assert rel_err(dx, x.grad) < 2e-5
    assert rel_err(dgm, gamma.grad) < 2e-5
    assert rel_err(dbt, beta.grad) < 2e-5


def _synth_wav(n, sr, seed):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / sr
    f0 = 220 * 2 ** ((3 * np.sin(2 * np.pi * 0.25 * t) + 0.3 * np.sin(2 * np.pi * 5.5 * t)) / 12)
    ph = 2 * np.pi * np.cumsum(f0) / sr
    w = sum(np.sin(h * ph) / h for h in range(1, 8))
    w = 0.5 * w / np.abs(w).max() + 0.03 * rng.randn(n)
    return w.astype(np.float32)


@pytest.mark.parametrize("sr,fmax", [(24000, 12000), (22050, 11025)])
def test_stft_mel_offline(dev, sr, fmax):
    """D2: data_gen_utils.py:123-134.  Bit-exact frame count (1 + N//hop); log10-mel within 1e-4 abs
    (north-star: mel-L1 <= 1e-4)."""
    n = 128 * 37 + 51
    wav = np.stack([_synth_wav(n, sr, 0), _synth_wav(n, sr, 1)])
    basis = ofe.librosa_mel_filterbank(sr, 512, 80, 50, fmax)
    win = ofe.hann_periodic(512).astype(np.float32)
    ref = np.stack([ofe.wav2mel_offline(w, 512, 128, 512, 80, 50, fmax, sr)[1] for w in wav])
    out = K.stft_mel(torch.from_numpy(wav).to(dev), torch.from_numpy(win).to(dev), torch.from_numpy(basis).to(dev),
                     512, 128, 0, 1e-10)
    assert out.shape == (2, 1 + n // 128, 80) == ref.shape
    assert np.abs(out.cpu().numpy() - ref).mean() < 1e-5
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-3


def test_stft_mel_ingraph(dev):
    """D4: mel_utils.py:59-76 (reflect pad, center=False, ln, clamp 1e-5) -> [B, 80, N/hop]."""
    sr, n = 24000, 8192
    wav = np.stack([_synth_wav(n, sr, 2) * 1.9, _synth_wav(n, sr, 3)])  # first clip exercises the [-1,1] clamp
    ref = ofe.mel_spectrogram_ingraph(torch.from_numpy(wav), 512, 128, 512, 80, 50, 12000, sr)
    basis = ofe.librosa_mel_filterbank(sr, 512, 80, 50, 12000)
    win = torch.hann_window(512, periodic=True)
    out = K.stft_mel(torch.from_numpy(wav).to(dev), win.to(dev), torch.from_numpy(basis).to(dev), 512, 128, 1, 1e-5)
    assert out.shape == (2, 80, 64) == tuple(ref.shape)
    assert (out.cpu() - ref).abs().mean() < 1e-5
    assert (out.cpu() - ref).abs().max() < 1e-3


def test_nsf_source(dev):
    """V2: source.py:44-137,385-398.  waveform within 2e-4 of the reference's fp32 arithmetic; the fp64-phase
    restatement bounds how much of that is the reference's own cumsum rounding."""
    g_ = torch.Generator().manual_seed(3)
    B, frames, upp, H, sr = 3, 40, 128, 9, 24000.0
    f0 = 100 + 300 * torch.rand(B, frames, generator=g_)
    f0[:, 5:9] = 0.0
    f0[1, 20:] = 0.0
    rand_ini = torch.rand(B, H, generator=g_)
    rand_ini[:, 0] = 0
    noise = torch.randn(B, frames * upp, H, generator=g_)
    lw = torch.randn(H, generator=g_) * 0.5
    lb = torch.randn(1, generator=g_) * 0.1
    m_ref, sw_ref, uv_ref = oops.sine_source(f0, rand_ini, noise, lw, lb, upp, sr)
    m64, sw64, _ = oops.sine_source_f64(f0, rand_ini, noise, lw, lb, upp, sr)
    m, sw, uv = K.nsf_source(f0.to(dev), rand_ini.to(dev), noise.to(dev), lw.to(dev), lb.to(dev), upp, sr,
                             want_sine_waves=True, want_uv=True)
    assert torch.equal(uv.cpu(), uv_ref)
    assert (sw.cpu() - sw64).abs().max() < 2e-5          # vs exact-phase restatement
    assert (sw.cpu() - sw_ref).abs().max() < 2e-4          # vs the reference's fp32 cumsum
    assert (m.cpu() - m_ref).abs().max() < 2e-4


def test_f0_to_coarse_bit_exact(dev):
    """D3: pitch_utils.py:130-146; numpy (rint) and torch ((x+0.5).long()) branches, exact integers."""
    rng = np.random.RandomState(0)
    f0 = np.concatenate([np.zeros(50), rng.uniform(40, 1300, 5000), [50.0, 1100.0, 1e-3, 49.9]])
    ref = ofe.f0_to_coarse(f0)
    out = K.f0_to_coarse(torch.from_numpy(f0).to(dev))
    assert out.dtype == torch.int64 and np.array_equal(out.cpu().numpy(), ref)
    f32 = torch.from_numpy(f0.astype(np.float32))
    ref32 = ofe.f0_to_coarse(f32)
    out32 = K.f0_to_coarse(f32.to(dev))
    assert torch.equal(out32.cpu(), ref32)          # torch branch: correctly rounded fp32 log -> exact integers too
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "f0_to_coarse.npz"))      # the reference's own outputs
    g32 = K.f0_to_coarse(torch.from_numpy(d["f0"].astype(np.float32)).to(dev))
    assert np.array_equal(g32.cpu().numpy(), d["coarse_torch"])
    g64 = K.f0_to_coarse(torch.from_numpy(d["f0"]).to(dev))
    assert np.array_equal(g64.cpu().numpy(), d["coarse_np"])


def test_ssim_map_forward_backward(dev):
    """L1: modules/commons/ssim.py:331-351 on [B,1,T,80]+6; pred is read through its [B,80,T] (transposed) strides."""
    from oracle import modules_ref as R
    g_ = torch.Generator().manual_seed(4)
    B, T, Fb = 2, 37, 80
    pred_nct = (torch.randn(B, Fb, T, generator=g_) * 0.8 - 3).requires_grad_(True)
    tgt = torch.randn(B, T, Fb, generator=g_) * 0.8 - 3
    tgt[1, 30:] = 0.0
    ref = R.ssim_map(pred_nct.transpose(1, 2)[:, None] + 6.0, tgt[:, None] + 6.0)
    dm = torch.randn(ref.shape, generator=g_)
    ref.backward(dm)
    p = pred_nct.detach().to(dev).transpose(1, 2)            # non-contiguous view, as in the model
    out = K.ssim_fwd(p, tgt.to(dev), 6.0)
    # the fp32 oracle is itself 3.0e-5 away from the same map in fp64 on this input (E[xy] - mu_x mu_y cancels to ~1e-6 of its
    # operands where the target is silent); the separable kernel sits at 1.3e-5 from fp64 and 2.9e-5 from the fp32 oracle
    assert (out.cpu() - ref.detach()).abs().max() < 5e-5
    dp = K.ssim_bwd(p, tgt.to(dev), dm.to(dev), 6.0)
    assert rel_err(dp, pred_nct.grad.transpose(1, 2)) < 1e-4


@pytest.mark.parametrize("shape", [(1, 5, 128), (2, 16, 8), (1, 33, 1)])
def test_ssim_map_shape_corners(dev, shape):
    """The widest image the stencil kernels take (128 bins: 95 KB of dynamic LDS), clips shorter than one 16-frame tile, one bin."""
    from oracle import modules_ref as R
    B, T, Fb = shape
    g_ = torch.Generator().manual_seed(B * 1000 + T + Fb)
    pred = (torch.randn(B, T, Fb, generator=g_) * 0.8 - 3).requires_grad_(True)
    tgt = torch.randn(B, T, Fb, generator=g_) * 0.8 - 3
    ref = R.ssim_map(pred[:, None] + 6.0, tgt[:, None] + 6.0)
    dm = torch.randn(ref.shape, generator=g_)
    ref.backward(dm)
    out = K.ssim_fwd(pred.detach().to(dev), tgt.to(dev), 6.0)
    assert (out.cpu() - ref.detach()).abs().max() < 5e-5
    dp = K.ssim_bwd(pred.detach().to(dev), tgt.to(dev), dm.to(dev), 6.0)
    assert rel_err(dp, pred.grad) < 1e-4


@pytest.mark.parametrize("terms", [(True, True), (True, False), (False, True)])
def test_mel_loss_fused_forward_backward(dev, terms):
    """l1_loss + ssim_loss with weights_nonzero_speech (tasks/tts/fs2.py:143-175) as one pass: values and d/d pred vs the
    oracle's restatement; pred read through transposed strides, silent (all-zero) target frames carry no weight."""
    from neuralsvb_amd import functional as SF
    from oracle import modules_ref as R
    g_ = torch.Generator().manual_seed(14)
    B, T, Fb = 3, 41, 80
    pred_nct = (torch.randn(B, Fb, T, generator=g_) * 0.8 - 3).requires_grad_(True)
    tgt = torch.randn(B, T, Fb, generator=g_) * 0.8 - 3
    tgt[1, 30:] = 0.0
    tgt[2, 17:] = 0.0
    tgt[0, 5, 3] = 0.0                                       # a single zero bin does not silence its frame
    pr = pred_nct.transpose(1, 2)
    l1, ss = R.l1_loss(pr, tgt), R.ssim_loss(pr, tgt)
    cl1, css = 0.7, -1.3
    ((cl1 * l1 if terms[0] else 0) + (css * ss if terms[1] else 0)).backward()
    p = pred_nct.detach().to(dev).transpose(1, 2).requires_grad_(True)
    out = SF.mel_loss(p, tgt.to(dev), 6.0, l1=terms[0], ssim=terms[1])
    if terms[0]:
        assert abs(out[0].item() - l1.item()) < 2e-6 * abs(l1.item()) + 1e-7
    if terms[1]:
        assert abs(out[1].item() - ss.item()) < 2e-5
    assert out[2].item() == R.weights_nonzero_speech(tgt).sum().item()
    (out * torch.tensor([cl1, css, 0.0], device=dev)).sum().backward()
    assert rel_err(p.grad, pred_nct.grad.transpose(1, 2)) < 1e-4


def test_layernorm_nct(dev):
    g_ = torch.Generator().manual_seed(8)
    x = torch.randn(3, 96, 70, generator=g_) * 2 + 0.3
    gm, bt = torch.randn(96, generator=g_), torch.randn(96, generator=g_)
    ref = oops.layernorm(x.transpose(1, 2), gm, bt).transpose(1, 2)
    y = K.layernorm_nct_fwd(x.to(dev), gm.to(dev), bt.to(dev))
    assert (y.cpu() - ref).abs().max() < 3e-5


@pytest.mark.parametrize("case", CONV_CASES + [(2, 200, 70, 1, 150, 5, 1, 2, 1), (1, 48, 96, 1, 300, 1, 1, 0, 1)])
def test_conv1d_bf16x3_forward_dgrad(dev, case):
    """bf16x3 variant (hi*hi + hi*lo + lo*hi on the bf16 matrix cores): relative error bound 6e-5 per conv."""
    B, Cin, Cout, G, T, k, s, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin // G, k, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = oops.conv1d(xr, w, bias, s, pad, dil, G)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    qa, qb = K.weight_pack_q(w.to(dev), None, G)
    y = K.conv1d_forward(x.to(dev), qa, Cout, k, s, pad, dil, G, bias=bias.to(dev))
    assert y.shape == ref.shape and rel_err(y, ref) < 6e-5
    dx = K.conv1d_transposed(dy.to(dev), qb, Cin, T, k, s, pad, dil, G)
    assert rel_err(dx, xr.grad) < 6e-5


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_conv1d_bf16x3_tile_configs_and_convt(dev, cfg):
    g = torch.Generator().manual_seed(cfg)
    B, Cin, Cout, T, k, s, pad = 2, 40, 72, 37, 8, 4, 2
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, k, generator=g) * 0.2
    gn = torch.rand(Cin, 1, 1, generator=g) + 0.5
    ref = oops.conv_transpose1d(x, oops.weight_norm(w, gn), None, s, pad)
    qa, qb = K.weight_pack_q(w.to(dev), gn.to(dev), 1)
    y = K.conv1d_transposed(x.to(dev), qb, Cout, ref.shape[-1], k, s, pad, 1, 1, force_cfg=cfg)
    assert rel_err(y, ref) < 6e-5
    dy = torch.randn(ref.shape, generator=g)
    dref = torch.autograd.grad(oops.conv_transpose1d(x.requires_grad_(True), oops.weight_norm(w, gn), None, s, pad), x, dy)[0]
    dx = K.conv1d_forward(dy.to(dev), qa, Cin, k, s, pad, 1, 1, force_cfg=cfg)
    assert rel_err(dx, dref) < 6e-5


WGQ_STRIDED = [
    # B, Cin, Cout, G, T, k, s, pad
    (2, 6, 10, 2, 137, 3, 2, 1),         # grouped + strided
    (1, 16, 24, 1, 264, 8, 4, 2),        # pre-net family k8 s4 p2 (fs2_vae.py:109-114)
    (1, 32, 32, 4, 400, 41, 4, 20),      # MSD grouped k41 s4 (hifigan.py:264-266): 4 phases x 3 tap groups
    (3, 5, 3, 1, 229, 5, 3, 2),          # MPD stride 3 (hifigan.py:202-215)
    (2, 8, 8, 1, 150, 4, 4, 0),          # k = s (ConvTranspose-style upsampling weight gradient)
]


WGQ_GROUPED16 = [
    # B, Cin, Cout, G, T, k, s, pad     (per group: Cout/G output x Cin/G input channels)
    (2, 32, 64, 4, 300, 41, 2, 20),      # MSD family (hifigan.py:259-268): 16 x 8 channels per group, 3 tap tiles, stride 2
    (1, 64, 128, 4, 260, 41, 4, 20),     # 32 x 16 per group (two 16-row blocks per group), stride 4
    (2, 64, 64, 2, 150, 41, 1, 20),      # 32 x 32 per group, stride 1
    (1, 128, 256, 4, 90, 41, 4, 20),     # 64 x 32 per group
    (2, 16, 64, 4, 200, 9, 1, 4),        # 16 x 4 per group, one tap tile
    (1, 32, 32, 2, 130, 20, 2, 3),       # two tap tiles, odd padding / stride phase
    (1, 8, 32, 2, 5, 3, 1, 1),           # a clip shorter than one 64-position chunk, 4 input channels per group
    (1, 64, 32, 2, 70, 48, 4, 24),       # the envelope's corners: k = 48, stride 4, 32 input channels per group (two workgroups)
    (3, 16, 16, 1, 40, 5, 1, 2),         # groups == 1: NOT this kernel (dispatch check: same result through the 32x32 path)
]


@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("case", WGQ_GROUPED16)
def test_conv1d_wgrad_bf16x3_grouped_16row_kernel(dev, case, gated):
    """Grouped weight gradients with 16 | (output channels per group) and 4..32 input channels per group run on the turned
    GEMM (16 output channels x 16 taps per MFMA tile, csrc/conv1d_bf16.hip svb_conv1d_wgrad_g16_kernel): weight and bias
    gradients against torch autograd, plain and with both activation-derivative gates."""
    B, Cin, Cout, G, T, k, s, pad = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, T, generator=g)
    w = (torch.randn(Cout, Cin // G, k, generator=g) * 0.2).requires_grad_(True)
    bias = torch.randn(Cout, generator=g).requires_grad_(True)
    xin = F.leaky_relu(x, 0.1) if gated else x
    pre = oops.conv1d(xin, w, bias, s, pad, 1, G)
    y = F.leaky_relu(pre, 0.2) if gated else pre
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    kw = dict(a_gate=y.detach().to(dev), a_slope=0.2, b_gate=x.to(dev), b_slope=0.1) if gated else {}
    dw, db = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, s, pad, 1, G, bf16x3=True, want_bias=True, **kw)
    assert dw.shape == w.shape
    assert rel_err(dw, w.grad) < 6e-5
    assert rel_err(db, bias.grad) < 1e-5


@pytest.mark.parametrize("case", WGQ_STRIDED)
def test_conv1d_wgrad_bf16x3_strided(dev, case):
    """Strided weight gradient = `stride` stride-1 problems over the phase subsequences of the input."""
    B, Cin, Cout, G, T, k, s, pad = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, T, generator=g)
    w = (torch.randn(Cout, Cin // G, k, generator=g) * 0.2).requires_grad_(True)
    y = oops.conv1d(x, w, None, s, pad, 1, G)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw, db = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, s, pad, 1, G, bf16x3=True, want_bias=True)
    assert rel_err(dw, w.grad) < 6e-5
    assert rel_err(db, dy.sum((0, 2))) < 1e-5


WGQ_CASES = [
    # B, Cin, Cout, G, T, k, pad, dil
    (2, 8, 16, 1, 50, 5, 2, 1),
    (1, 70, 100, 1, 150, 5, 2, 1),       # ragged channel tiles, 3 chunks with a ragged tail
    (2, 40, 40, 1, 33, 1, 0, 1),         # 1x1
    (2, 12, 20, 1, 90, 3, 1, 1),
    (1, 12, 12, 1, 200, 7, 9, 3),        # dilated, two tap groups (general shifted-operand path)
    (1, 10, 6, 1, 130, 11, 25, 5),       # k11 d5 (hifigan.py:33-41)
    (2, 8, 12, 2, 77, 4, 2, 1),          # grouped, even taps
    (1, 6, 6, 1, 64, 2, 0, 1),
    (2, 40, 200, 1, 70, 1, 0, 1),        # 1x1, Cout (A rows) >= 128: 128x64 workgroup tile, ragged second tile
    (2, 130, 70, 1, 90, 1, 0, 1),        # 1x1, Cin (B rows) >= 128 only: 64x128 workgroup tile
    (1, 256, 384, 1, 130, 1, 0, 1),      # both sides wide: the A side is doubled
    (2, 24, 40, 1, 110, 5, 22, 11),      # MPD (5,1) stride-1 conv as a dilation-p conv over [H, p] planes: tap groups narrowed to 3
    (2, 20, 12, 1, 150, 2, 7, 7),        # ... and its space-to-depth form of the strided layers: 2 taps, dilation p
    (2, 32, 64, 4, 140, 9, 4, 1),        # grouped, 16 x 8 channels per group: all 4 groups packed into one tile
    (1, 48, 96, 6, 100, 3, 1, 1),        # 6 groups of 16 x 8: packed in pairs (largest power of two dividing 6)
    (2, 128, 64, 2, 90, 3, 1, 1),        # 32 x 64 per group: nothing to pack
    (2, 40, 200, 1, 150, 5, 2, 1),       # 5 taps, 200 output channels: the 128 x 64 tile at one workgroup per CU (round 4), ragged rows
]


@pytest.mark.parametrize("case", WGQ_CASES)
def test_conv1d_wgrad_bf16x3(dev, case):
    """Weight gradient on the bf16x3 kernel (shifted operand assembled in registers) against torch autograd."""
    B, Cin, Cout, G, T, k, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, T, generator=g)
    w = (torch.randn(Cout, Cin // G, k, generator=g) * 0.2).requires_grad_(True)
    y = oops.conv1d(x, w, None, 1, pad, dil, G)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw, db = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, pad, dil, G, bf16x3=True, want_bias=True)
    assert dw.shape == w.shape
    assert rel_err(dw, w.grad) < 6e-5
    assert rel_err(db, dy.sum((0, 2))) < 1e-5          # bias-gradient partials fused into the same two launches


@pytest.mark.parametrize("gates", ["none", "a", "self_b", "both", "b"])
@pytest.mark.parametrize("case", [(3, 64, 64, 53), (2, 130, 70, 281), (1, 256, 384, 130), (5, 96, 200, 16), (4, 64, 128, 1),
                                  (2, 192, 64, 97)])
def test_conv1d_wgrad_pointwise_direct_operands(dev, case, gates):
    """Weight gradient of the 1-tap convs on the direct-operand kernel (csrc/conv1d_wgrad_pw.hip: both operands global -> VGPR -> split
    -> MFMA, no LDS): clips whose length is not a multiple of the 16-position step (ragged last step, chunks with fewer than four
    steps, T = 1), ragged row tiles on both sides, 64 x 64 / 64 x 32 / 32 x 64 wave tiles, every gate form (dy gated by the saved
    output, x gated by itself, by a second tensor, both), bias partials -- against torch autograd."""
    B, Cin, Cout, T = case
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + T + len(gates))
    x = torch.randn(B, Cin, T, generator=g)
    w = (torch.randn(Cout, Cin, 1, generator=g) * 0.2).requires_grad_(True)
    gx = torch.randn(B, Cin, T, generator=g)
    xin = x
    kw = {}
    if gates in ("self_b", "both"):
        xin = F.leaky_relu(x, 0.1) if gates == "self_b" else x * torch.where(gx > 0, 1.0, 0.1)
        kw.update(b_gate=(x if gates == "self_b" else gx).to(dev), b_slope=0.1)
    if gates == "b":
        xin = x * torch.where(gx > 0, 1.0, 0.3)
        kw.update(b_gate=gx.to(dev), b_slope=0.3)
    y = oops.conv1d(xin, w, None, 1, 0)
    if gates in ("a", "both"):
        y = F.leaky_relu(y, 0.2)
        kw.update(a_gate=y.detach().to(dev), a_slope=0.2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw, db = K.conv1d_wgrad(dy.to(dev), x.to(dev), 1, 1, 0, 1, 1, bf16x3=True, want_bias=True, **kw)
    assert rel_err(dw, w.grad) < 6e-5
    dyg = dy * torch.where(y > 0, 1.0, 0.2) if gates in ("a", "both") else dy
    assert rel_err(db, dyg.sum((0, 2))) < 1e-5


@pytest.mark.parametrize("case", [(1, 200, 72, 700, 12), (1, 512, 128, 333, 1), (2, 130, 40, 150, 3)])
def test_conv1d_wgrad_two_taps_a_gated(dev, case):
    """Two-tap weight gradients whose A operand alone is gated (the mel critic's towers: dy gated by the saved conv output, the
    space-to-depth input as it is; one-sided tap pairs, pad = P + 1 or 1), wide B sides with ragged blocks and ragged A rows, with the
    bias partials -- against fp64."""
    B, CB, CA, T, pad = case
    g = torch.Generator().manual_seed(CB + CA + T)
    x = torch.randn(B, CB, T, generator=g)
    dy = torch.randn(B, CA, T, generator=g)
    yact = torch.randn(B, CA, T, generator=g)
    gdy = (dy * torch.where(yact > 0, 1.0, 0.2)).double()
    ref = torch.zeros(CA, CB, 2, dtype=torch.float64)
    for j in range(2):
        off = j - pad
        xs = torch.zeros(B, CB, T, dtype=torch.float64)
        lo, hi = max(0, -off), min(T, T - off)
        if hi > lo:
            xs[:, :, lo:hi] = x[:, :, lo + off:hi + off].double()
        ref[:, :, j] = torch.einsum("bat,bct->ac", gdy, xs)
    dw, db = K.conv1d_wgrad(dy.to(dev), x.to(dev), 2, 1, pad, 1, 1, a_gate=yact.to(dev), a_slope=0.2, bf16x3=True, want_bias=True)
    assert rel_err(dw, ref.float()) < 6e-5
    assert rel_err(db, gdy.sum((0, 2)).float()) < 1e-5


@pytest.mark.parametrize("case", [(24, 10, 3, 1), (36, 24, 5, 3), (16, 40, 1, 1), (48, 12, 2, 4)])
def test_weight_pack_bf16x3_writes_its_padding(dev, case):
    """The bf16x3 weight pack fills the padding entries of both layouts itself (channel counts that are not multiples of 16, grouped
    convs): packed into buffers pre-filled with a NaN bit pattern, forward and data gradient must come out as from zero-filled
    buffers (a stale padding entry would put NaN into every output)."""
    d0, d1g, k, G = case
    g = torch.Generator().manual_seed(d0 + d1g + k + G)
    w = (torch.randn(d0, d1g, k, generator=g) * 0.3).to(dev)
    x = torch.randn(2, d1g * G, 50, generator=g).to(dev)
    dy = torch.randn(2, d0, 50, generator=g).to(dev)
    qa0, qb0 = K.weight_pack_q(w, None, G)
    qa, qb = K.weight_pack_q_alloc(w, G, True, True)
    for t in (qa.hi, qa.lo, qb.hi, qb.lo):
        t.fill_(0x7FC1)                                   # bf16 NaN
    K.weight_pack_q_into(w, None, G, qa, qb)
    pad = (k - 1) // 2
    y0 = K.conv1d_forward(x, qa0, d0, k, 1, pad, 1, G)
    y1 = K.conv1d_forward(x, qa, d0, k, 1, pad, 1, G)
    assert torch.equal(y0, y1) and torch.isfinite(y1).all()
    tout = x.shape[-1]
    dx0 = K.conv1d_transposed(dy[:, :, :y0.shape[-1]].contiguous(), qb0, d1g * G, tout, k, 1, pad, 1, G)
    dx1 = K.conv1d_transposed(dy[:, :, :y0.shape[-1]].contiguous(), qb, d1g * G, tout, k, 1, pad, 1, G)
    assert torch.equal(dx0, dx1) and torch.isfinite(dx1).all()
    ref = oops.conv1d(x.cpu(), w.cpu(), None, 1, pad, 1, G)
    assert rel_err(y1, ref) < 6e-5


@pytest.mark.parametrize("bf16x3", [True, False])
def test_conv1d_wgrad_bias_sink_only(dev, bf16x3):
    """bias_sink: the bias gradient alone is accumulated into an existing buffer (svb_wgrad_reduce accumulate = 2) while the
    weight gradient is returned -- the critic's convs, whose weight gradient still passes through a re-layout."""
    g_ = torch.Generator().manual_seed(12)
    B, Cin, Cout, T, k = 2, 12, 20, 97, 2
    x, dy = torch.randn(B, Cin, T, generator=g_), torch.randn(B, Cout, T, generator=g_)
    dw0, db0 = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, 0, 1, 1, bf16x3=bf16x3, want_bias=True)
    sink = torch.full((Cout,), 0.5, device=dev)
    dw1, db1 = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, 0, 1, 1, bf16x3=bf16x3, want_bias=True, bias_sink=sink)
    assert db1 is None and torch.equal(dw1, dw0)
    assert (sink - 0.5 - db0).abs().max() < 1e-5 * db0.abs().max()
    assert rel_err(db0, dy.sum((0, 2))) < 1e-5


@pytest.mark.parametrize("cin,k", [(1040, 5), (2064, 4)])
def test_conv1d_wgrad_weight_norm_sinks_long_rows(dev, cin, k):
    """WeightNorm backward accumulated straight into `.grad` sinks for rows longer than the old 4096-element register window
    (the period discriminators' 1024 -> 1024 (5,1) convs: 5120 elements per row -- round 6 widened the window to 8192 so that these
    gradients qualify for the side stream) and, beyond 8192, the refusal that sends the caller down the returned-tensor path."""
    g_ = torch.Generator().manual_seed(cin)
    B, Cout, T = 1, 4, 24
    x = torch.randn(B, cin, T, generator=g_)
    v = (torch.randn(Cout, cin, k, generator=g_) * 0.1).requires_grad_(True)
    gn = (torch.rand(Cout, 1, 1, generator=g_) + 0.5).requires_grad_(True)
    w = gn * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
    pad = (k - 1) // 2
    y = oops.conv1d(x, w, None, 1, pad)
    dy = torch.randn(y.shape, generator=g_)
    y.backward(dy)
    ta = y.shape[-1]
    sk = (torch.full(v.shape, 0.25, device=dev), torch.full(gn.shape, -0.5, device=dev), torch.full((Cout,), 1.5, device=dev))
    if cin * k <= 8192:
        r = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, pad, 1, 1, v=v.detach().to(dev), g=gn.detach().to(dev), bf16x3=True,
                           want_bias=True, sinks=sk)
        assert all(t is None for t in r)
        assert rel_err(sk[0] - 0.25, v.grad) < 1e-4 and rel_err(sk[1] + 0.5, gn.grad) < 1e-4
        assert rel_err(sk[2] - 1.5, dy.sum((0, 2))) < 1e-5
    else:
        dv, dg, db = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, pad, 1, 1, v=v.detach().to(dev), g=gn.detach().to(dev), bf16x3=True,
                                    want_bias=True, sinks=sk)
        got_v = dv if dv is not None else sk[0] - 0.25
        got_g = dg if dg is not None else sk[1] + 0.5
        assert rel_err(got_v, v.grad) < 1e-4 and rel_err(got_g, gn.grad) < 1e-4
    del ta


def test_conv1d_wgrad_bf16x3_gates_and_weight_norm(dev):
    g_ = torch.Generator().manual_seed(11)
    B, Cin, Cout, T, k = 2, 10, 14, 145, 3
    x = torch.randn(B, Cin, T, generator=g_)
    v = (torch.randn(Cout, Cin, k, generator=g_) * 0.3).requires_grad_(True)
    gn = (torch.rand(Cout, 1, 1, generator=g_) + 0.5).requires_grad_(True)
    w = gn * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
    y = torch.relu(oops.conv1d(F.leaky_relu(x, 0.1), w, None, 1, 1))
    dy = torch.randn(y.shape, generator=g_)
    y.backward(dy)
    dv, dg, db = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, 1, 1, 1, a_gate=y.detach().to(dev), a_slope=0.0,
                                b_gate=x.to(dev), b_slope=0.1, v=v.detach().to(dev), g=gn.detach().to(dev), bf16x3=True,
                                want_bias=True)
    assert rel_err(dv, v.grad) < 1e-4
    assert rel_err(dg, gn.grad) < 1e-4
    assert rel_err(db, (dy * (y > 0)).sum((0, 2))) < 1e-5


@pytest.mark.parametrize("cfg", [2, 3, 8, 9, 10, 11, 12, 16, 17])
@pytest.mark.parametrize("shape", [(64, 1), (128, 1), (80, 5), (160, 5), (48, 3), (64, 3)])
def test_conv1d_bf16x3_direct_tiles_whole_phase_loops(dev, cfg, shape):
    """Direct-A tiles (weight fragments straight from global memory) on K extents that divide into whole 5-slab phases
    (k5: 5 taps x 1 chunk; 80 / 160 channels x 1 tap), whole 4-slab phases (1x1 and k3 convs over 4 / 8 / 4 chunks) or
    neither (48 channels x 3 taps: the generic loop) -- with bias + LeakyReLU in the hoisted epilogue, and the transposed
    form with residual + mask through the general epilogue."""
    Cin, k = shape
    g = torch.Generator().manual_seed(cfg * 100 + Cin + k)
    B, Cout, T = 2, 72, 150
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    ref = F.leaky_relu(oops.conv1d(x, w, bias, 1, k // 2), 0.2)
    qa, qb = K.weight_pack_q(w.to(dev), None, 1)
    y = K.conv1d_forward(x.to(dev), qa, Cout, k, 1, k // 2, 1, 1, bias=bias.to(dev), out_act=K.ACT_LRELU, out_slope=0.2,
                         force_cfg=cfg)
    assert rel_err(y, ref) < 6e-5
    dy = torch.randn(ref.shape, generator=g)
    res = torch.randn(B, Cin, T, generator=g)
    mask = (torch.rand(B, T, generator=g) > 0.2).float()
    dref = (torch.autograd.grad(oops.conv1d(x.requires_grad_(True), w, None, 1, k // 2), x, dy)[0] + res) * mask[:, None]
    dx = K.conv1d_transposed(dy.to(dev), qb, Cin, T, k, 1, k // 2, 1, 1, residual=res.to(dev), mask=mask.to(dev), force_cfg=cfg)
    assert rel_err(dx, dref) < 6e-5


@pytest.mark.parametrize("cfg", [13, 14, 15])
@pytest.mark.parametrize("shape", [(64, 1, 1), (32, 3, 1), (48, 5, 1), (32, 3, 3), (96, 1, 1), (192, 5, 1)])
def test_conv1d_bf16x3_tile_walking_kernel(dev, cfg, shape):
    """The producer / consumer tile-walking kernel (csrc/conv1d_tw.hip, configurations 13..15): weights through LDS by LDS-DMA,
    columns flattened over the batch (tiles straddle clip boundaries: 3 clips x 150 positions in 64- / 128- / 256-column tiles),
    several tiles per workgroup (the emulator's device has 4 CUs: deferred stores drained behind the next tile, or flushed when
    the K extent is short), 1 / 2 / 4 chunks per K phase, a ragged last row tile (72 outputs); forward with bias + LeakyReLU
    (the deferred epilogue), the transposed form with residual + mask (the immediate one), and the input gate, against the
    oracle."""
    Cin, k, dil = shape
    g = torch.Generator().manual_seed(cfg * 100 + Cin + k)
    B, Cout, T = 3, 72, 150
    pad = dil * (k - 1) // 2
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    ref = F.leaky_relu(oops.conv1d(x, w, bias, 1, pad, dil), 0.2)
    qa, qb = K.weight_pack_q(w.to(dev), None, 1)
    y = K.conv1d_forward(x.to(dev), qa, Cout, k, 1, pad, dil, 1, bias=bias.to(dev), out_act=K.ACT_LRELU, out_slope=0.2,
                         force_cfg=cfg)
    assert rel_err(y, ref) < 6e-5
    dy = torch.randn(ref.shape, generator=g)
    res = torch.randn(B, Cin, T, generator=g)
    mask = (torch.rand(B, T, generator=g) > 0.2).float()
    dref = (torch.autograd.grad(oops.conv1d(x.requires_grad_(True), w, None, 1, pad, dil), x, dy)[0] + res) * mask[:, None]
    dx = K.conv1d_transposed(dy.to(dev), qb, Cin, T, k, 1, pad, dil, 1, residual=res.to(dev), mask=mask.to(dev), force_cfg=cfg)
    assert rel_err(dx, dref) < 6e-5
    gate = torch.randn(B, Cin, T, generator=g)
    refg = oops.conv1d(x.detach() * torch.where(gate > 0, 1.0, 0.2), w, None, 1, pad, dil)
    yg = K.conv1d_forward(x.detach().to(dev), qa, Cout, k, 1, pad, dil, 1, in_gate=gate.to(dev), in_slope=0.2, force_cfg=cfg)
    assert rel_err(yg, refg) < 6e-5


@pytest.mark.parametrize("cfg", [18, 19, 20, 21, 22, 23])
@pytest.mark.parametrize("shape", [(64, 72, 150), (128, 96, 150), (192, 264, 37), (64, 40, 281)])
def test_conv1d_bf16x3_pointwise_gemm_kernel(dev, cfg, shape):
    """The pointwise GEMM form (csrc/conv1d_pw.hip, configurations 18..23): 1-tap convs with the batch's positions flattened into
    one column space (3 clips: tiles straddle clip boundaries, the last tile is ragged), the x operand global -> VGPR -> split, the
    weights by LDS-DMA in 2- / 4-chunk phases, ragged last row tiles (72, 264, 40 outputs on 64- / 96- / 128-row
    tiles).  Forward with bias + LeakyReLU (the plain epilogue), with output gate + residual + mask and with tanh (the general
    one), the transposed form (data gradient) with residual + mask -- each against the oracle AND bit for bit against a tap-table
    tile of the family (same split, same accumulation order per output)."""
    Cin, Cout, T = shape
    g = torch.Generator().manual_seed(cfg * 100 + Cin + Cout)
    B = 3
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, 1, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    qa, qb = K.weight_pack_q(w.to(dev), None, 1)
    xd = x.to(dev)
    ref = F.leaky_relu(oops.conv1d(x, w, bias, 1, 0), 0.2)
    kw = dict(bias=bias.to(dev), out_act=K.ACT_LRELU, out_slope=0.2)
    y = K.conv1d_forward(xd, qa, Cout, 1, 1, 0, 1, 1, force_cfg=cfg, **kw)
    assert rel_err(y, ref) < 6e-5
    assert torch.equal(y, K.conv1d_forward(xd, qa, Cout, 1, 1, 0, 1, 1, force_cfg=4, **kw))
    gate = torch.randn(B, Cout, T, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    mask = (torch.rand(B, T, generator=g) > 0.2).float()
    ref2 = ((oops.conv1d(x, w, bias, 1, 0)) * torch.where(gate > 0, 1.0, 0.3) + res) * mask[:, None]
    kw = dict(bias=bias.to(dev), out_gate=gate.to(dev), out_gate_slope=0.3, residual=res.to(dev), mask=mask.to(dev))
    y2 = K.conv1d_forward(xd, qa, Cout, 1, 1, 0, 1, 1, force_cfg=cfg, **kw)
    assert rel_err(y2, ref2) < 6e-5
    assert torch.equal(y2, K.conv1d_forward(xd, qa, Cout, 1, 1, 0, 1, 1, force_cfg=4, **kw))
    y3 = K.conv1d_forward(xd, qa, Cout, 1, 1, 0, 1, 1, bias=bias.to(dev), out_act=K.ACT_TANH, force_cfg=cfg)
    assert rel_err(y3, torch.tanh(oops.conv1d(x, w, bias, 1, 0))) < 6e-5
    if Cout % 64 == 0 or cfg in (18, 21):        # the data gradient contracts over Cout: inside the kernel's domain when Cout % 64 == 0
        dy = torch.randn(B, Cout, T, generator=g)
        resx = torch.randn(B, Cin, T, generator=g)
        dref = (torch.autograd.grad(oops.conv1d(x.requires_grad_(True), w, None, 1, 0), x, dy)[0] + resx) * mask[:, None]
        kw = dict(residual=resx.to(dev), mask=mask.to(dev))
        dx = K.conv1d_transposed(dy.to(dev), qb, Cin, T, 1, 1, 0, 1, 1, force_cfg=cfg, **kw)
        assert rel_err(dx, dref) < 6e-5
        assert torch.equal(dx, K.conv1d_transposed(dy.to(dev), qb, Cin, T, 1, 1, 0, 1, 1, force_cfg=4, **kw))


def _taps_ref(x, w, offsets):
    B, Cin, T = x.shape
    y = torch.zeros(B, w.shape[0], T, dtype=torch.float64)
    for t, off in enumerate(offsets):
        xs = torch.zeros(B, Cin, T, dtype=torch.float64)
        lo, hi = max(0, -off), min(T, T - off)
        if hi > lo:
            xs[:, :, lo:hi] = x[:, :, lo + off:hi + off].double()
        y += torch.einsum("oc,bct->bot", w[:, :, t].double(), xs)
    return y.float()


@pytest.mark.parametrize("cfg", [18, 19, 20, 21, 22, 23])
@pytest.mark.parametrize("shape", [(64, 72, (-1, 0, 1), 150), (32, 40, (-8, -4, 0, 4, 8), 97), (48, 64, (-43, -42, -1, 0), 61),
                                   (64, 264, (0, 3), 40), (16, 32, (-70, -3, -2, -1, 0, 1, 2, 70), 45),
                                   (32, 64, tuple(range(-15, 16, 3)), 130), (16, 40, tuple(range(-8, 8)), 33)])
def test_conv1d_bf16x3_gemm_kernel_with_taps_and_input_gates(dev, cfg, shape):
    """The same kernel over (chunk, tap) slabs: stride-1 equal-length convs of up to 16 taps with an arbitrary offset table
    (dilated, one-sided as the period critics' slot convs are, offsets past the clip's length), the tap's shift applied to the
    column's position with the zero padding answered by the buffer bounds check; the input gated by its own sign
    (`conv(leaky_relu(x))`) and by a second tensor (the critic towers' data gradients); forward and transposed forms against the
    oracle and a tap-table tile."""
    Cin, Cout, offsets, T = shape
    k = len(offsets)
    g = torch.Generator().manual_seed(cfg * 131 + Cin + k)
    B = 3
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    qa, qb = K.weight_pack_q(w.to(dev), None, 1)
    xd, bd = x.to(dev), bias.to(dev)
    ref = _taps_ref(x, w, offsets) + bias[None, :, None]
    y = K.conv1d_taps(xd, qa, Cout, offsets, bias=bd, force_cfg=cfg)
    assert rel_err(y, ref) < 6e-5
    assert rel_err(y, K.conv1d_taps(xd, qa, Cout, offsets, bias=bd, force_cfg=4).cpu()) < 2e-6
    yg = K.conv1d_taps(xd, qa, Cout, offsets, bias=bd, in_gate=xd, in_slope=0.1, force_cfg=cfg)
    assert rel_err(yg, _taps_ref(F.leaky_relu(x, 0.1), w, offsets) + bias[None, :, None]) < 6e-5
    gate = torch.randn(B, Cin, T, generator=g)
    yh = K.conv1d_taps(xd, qa, Cout, offsets, bias=bd, in_gate=gate.to(dev), in_slope=0.2, force_cfg=cfg)
    assert rel_err(yh, _taps_ref(x * torch.where(gate > 0, 1.0, 0.2), w, offsets) + bias[None, :, None]) < 6e-5
    if offsets == tuple(range(offsets[0], offsets[0] + (k - 1) * (offsets[1] - offsets[0]) + 1, offsets[1] - offsets[0])) \
            and offsets[0] == -offsets[-1]:
        dil, pad = offsets[1] - offsets[0], -offsets[0]
        dy = torch.randn(B, Cout, T, generator=g)
        gy = torch.randn(B, Cout, T, generator=g)
        xr = x.clone().requires_grad_(True)
        dref = torch.autograd.grad(oops.conv1d(xr, w, None, 1, pad, dil), xr, dy * torch.where(gy > 0, 1.0, 0.2))[0]
        dx = K.conv1d_transposed(dy.to(dev), qb, Cin, T, k, 1, pad, dil, 1, in_gate=gy.to(dev), in_slope=0.2, force_cfg=cfg)
        assert rel_err(dx, dref) < 6e-5


@pytest.mark.parametrize("cfg", [18, 21, 23])
def test_conv1d_bf16x3_gemm_kernel_edge_shapes(dev, cfg):
    """Edges of the GEMM-form kernel's domain: a single position per clip (T = 1), clips shorter than one 32-column block with many
    clips per tile, the smallest channel counts it takes (Cin = 16, Cout = 32), a single clip, tap tables whose every offset points
    outside the clip (the output is the bias) and offsets at the 16-bit limit's order of magnitude -- against the oracle."""
    g = torch.Generator().manual_seed(cfg)
    for B, Cin, Cout, T, offs in [(5, 16, 32, 1, (0,)), (40, 32, 40, 3, (-1, 0, 1)), (1, 64, 32, 700, (-2, 0, 2, 5)),
                                  (3, 16, 72, 9, (-20, 30)), (2, 48, 64, 50, (-3000, 0, 3000, 1)), (7, 32, 32, 33, (0, 1))]:
        x = torch.randn(B, Cin, T, generator=g)
        w = torch.randn(Cout, Cin, len(offs), generator=g) * 0.2
        bias = torch.randn(Cout, generator=g)
        qa, _ = K.weight_pack_q(w.to(dev), None, 1)
        ref = _taps_ref(x, w, offs) + bias[None, :, None]
        y = K.conv1d_taps(x.to(dev), qa, Cout, offs, bias=bias.to(dev), force_cfg=cfg)
        assert y.shape == ref.shape
        if all(abs(o) >= T for o in offs):
            assert torch.allclose(y.cpu(), bias[None, :, None].expand_as(ref), rtol=0, atol=0)
        assert rel_err(y, ref) < 6e-5, (B, Cin, Cout, T, offs)


@pytest.mark.parametrize("cfg", [18, 20])
def test_conv1d_bf16x3_pointwise_gemm_falls_back_outside_its_domain(dev, cfg):
    """Convs outside the kernel's domain (more than 16 taps, Cin % 16 != 0, strides, outputs longer than the input) run a tap-table
    tile of the family when its configuration is forced -- same results as that tile."""
    g = torch.Generator().manual_seed(cfg)
    B, T = 2, 90
    for Cin, Cout, k, s, pad in [(64, 64, 17, 1, 8), (40, 64, 1, 1, 0), (64, 64, 3, 1, 2), (64, 64, 1, 2, 0)]:
        x = torch.randn(B, Cin, T, generator=g).to(dev)
        w = torch.randn(Cout, Cin, k, generator=g) * 0.2
        qa, _ = K.weight_pack_q(w.to(dev), None, 1)
        y = K.conv1d_forward(x, qa, Cout, k, s, pad, 1, 1, force_cfg=cfg)
        assert torch.equal(y, K.conv1d_forward(x, qa, Cout, k, s, pad, 1, 1, force_cfg=4))


@pytest.mark.parametrize("cfg", [13, 14])
@pytest.mark.parametrize("T,k,pad,dil", [(4, 3, 1, 1), (7, 3, 9, 9), (131, 5, 2, 1), (260, 3, 27, 27), (66, 3, 5, 1)])
def test_conv1d_bf16x3_tile_walking_clip_edges(dev, cfg, T, k, pad, dil):
    """Clips shorter than the tap span, padding wider than the clip, over-wide padding that lengthens the output, dilation 27:
    the flattened column space (pitch = Tout + tap span) against the oracle; plus a strided conv, which is outside the
    kernel's domain and must come out of the heuristic tile unchanged."""
    g = torch.Generator().manual_seed(1000 * cfg + T + k)
    B, Cin, Cout = 2, 32, 40
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    qa, _ = K.weight_pack_q(w.to(dev), None, 1)
    ref = oops.conv1d(x, w, None, 1, pad, dil)
    y = K.conv1d_forward(x.to(dev), qa, Cout, k, 1, pad, dil, 1, force_cfg=cfg)
    assert y.shape == ref.shape and rel_err(y, ref) < 6e-5
    if T > 60:
        refs = oops.conv1d(x, w, None, 2, pad, dil)
        ys = K.conv1d_forward(x.to(dev), qa, Cout, k, 2, pad, dil, 1, force_cfg=cfg)
        assert ys.shape == refs.shape and rel_err(ys, refs) < 6e-5


@pytest.mark.parametrize("cfg", [6, 7, 9, 10])
def test_conv1d_bf16x3_wide_tiles_three_position_groups(dev, cfg):
    """64x192 / 64x256 tiles on a long sequence: the register-staged x path with up to 3 groups of 128 positions."""
    g = torch.Generator().manual_seed(cfg)
    B, Cin, Cout, T, k = 1, 24, 40, 300, 5
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    ref = oops.conv1d(x, w, None, 1, 2)
    qa, qb = K.weight_pack_q(w.to(dev), None, 1)
    y = K.conv1d_forward(x.to(dev), qa, Cout, k, 1, 2, 1, 1, force_cfg=cfg)
    assert rel_err(y, ref) < 6e-5
    dy = torch.randn(ref.shape, generator=g)
    dref = torch.autograd.grad(oops.conv1d(x.requires_grad_(True), w, None, 1, 2), x, dy)[0]
    dx = K.conv1d_transposed(dy.to(dev), qb, Cin, T, k, 1, 2, 1, 1, force_cfg=cfg)
    assert rel_err(dx, dref) < 6e-5


@pytest.mark.parametrize("cfg", [1, 2, 4, 11, 16, 17])
@pytest.mark.parametrize("T,k,pad,dil", [(4, 3, 1, 1), (5, 5, 2, 1), (7, 3, 9, 9), (129, 3, 3, 3), (131, 5, 2, 1), (260, 3, 27, 27),
                                         (66, 3, 5, 1)])
def test_conv1d_bf16x3_clip_edges(dev, cfg, T, k, pad, dil):
    """Tiles that straddle the clip's first or last position: padding wider than the clip, clips of 4..7 positions, T not a
    multiple of 4, over-wide padding that lengthens the output -- plain and with the LeakyReLU-derivative gate on the input,
    against the oracle."""
    g = torch.Generator().manual_seed(1000 * cfg + T + k)
    B, Cin, Cout = 2, 32, 40
    x = torch.randn(B, Cin, T, generator=g)
    gate = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    qa, _ = K.weight_pack_q(w.to(dev), None, 1)
    ref = oops.conv1d(x, w, None, 1, pad, dil)
    y = K.conv1d_forward(x.to(dev), qa, Cout, k, 1, pad, dil, 1, force_cfg=cfg)
    assert y.shape == ref.shape and rel_err(y, ref) < 6e-5
    refg = oops.conv1d(x * torch.where(gate > 0, 1.0, 0.2), w, None, 1, pad, dil)
    yg = K.conv1d_forward(x.to(dev), qa, Cout, k, 1, pad, dil, 1, in_gate=gate.to(dev), in_slope=0.2, force_cfg=cfg)
    assert rel_err(yg, refg) < 6e-5


@pytest.mark.parametrize("cfg", [5, 16, 17])
@pytest.mark.parametrize("k,dil", [(3, 1), (7, 3), (11, 5), (11, 1)])
def test_conv1d_bf16x3_narrow_channel_wide_time_tiles(dev, cfg, k, dil):
    """The vocoder's last generator stages (reference modules/hifigan/hifigan.py:30-67: ResBlock1 convs on 32 channels, kernel sizes
    3 / 7 / 11 with dilations up to 5) on the 32-row tiles: the 32 x 128 one and the round-5 32 x 256 forms (configurations 16 /
    17: LDS-staged and direct weight fragments) -- LeakyReLU on the operand load, bias, residual epilogue, a sequence that is not a
    multiple of the tile, forward and transposed, against the oracle."""
    g = torch.Generator().manual_seed(cfg * 31 + k + dil)
    B, C_, T = 2, 32, 700
    pad = dil * (k - 1) // 2
    x = torch.randn(B, C_, T, generator=g)
    w = torch.randn(C_, C_, k, generator=g) * 0.2
    bias = torch.randn(C_, generator=g)
    res = torch.randn(B, C_, T, generator=g)
    ref = oops.conv1d(F.leaky_relu(x, 0.1), w, bias, 1, pad, dil) + res
    qa, qb = K.weight_pack_q(w.to(dev), None, 1)
    y = K.conv1d_forward(x.to(dev), qa, C_, k, 1, pad, dil, 1, bias=bias.to(dev), in_gate=x.to(dev), in_slope=0.1,
                         residual=res.to(dev), force_cfg=cfg)
    assert rel_err(y, ref) < 6e-5
    dy = torch.randn(ref.shape, generator=g)
    dref = torch.autograd.grad(oops.conv1d(x.requires_grad_(True), w, None, 1, pad, dil), x, dy)[0]
    dx = K.conv1d_transposed(dy.to(dev), qb, C_, T, k, 1, pad, dil, 1, force_cfg=cfg)
    assert rel_err(dx, dref) < 6e-5


@pytest.mark.parametrize("T", [37, 64, 97, 130])
def test_relpos_attention_fused_matches_espnet(dev, T):
    """svb_relpos_attn_fwd (content scores + rel-shifted position scores + scale + key mask + softmax + value product in one
    kernel) against the reference's op sequence (espnet_transformer_attn.py:125-186) in fp32: legacy rel_shift via
    pad/view/slice, masked_fill(min), softmax, masked_fill(0), matmul with v.  One clip has padded keys, one is fully padded."""
    g = torch.Generator().manual_seed(100 + T)
    B, H, dk = 3, 2, 64
    q = torch.randn(B, H, dk, T, generator=g)
    k = torch.randn(B, H, dk, T, generator=g)
    v = torch.randn(B, H, dk, T, generator=g)
    pu, pv = torch.randn(H, dk, generator=g) * 0.5, torch.randn(H, dk, generator=g) * 0.5
    pe = torch.randn(1, H, dk, T, generator=g)
    keep = torch.ones(B, T)
    keep[1, T - 7:] = 0
    keep[2, :] = 0
    ac = torch.matmul((q + pu[None, :, :, None]).transpose(-1, -2), k)
    bd = torch.matmul((q + pv[None, :, :, None]).transpose(-1, -2), pe)             # [B,H,T,T], unshifted
    x = F.pad(bd, (1, 0)).view(B, H, T + 1, T)[:, :, 1:].reshape(B, H, T, T)
    scores = (ac + x) / (dk ** 0.5)
    drop = ~(keep.bool())[:, None, None, :]
    attn = torch.softmax(scores.masked_fill(drop, torch.finfo(torch.float32).min), -1).masked_fill(drop, 0.0)
    ref = torch.matmul(v, attn.transpose(-1, -2)).reshape(B, H * dk, T)
    out = K.relpos_attention(q.reshape(B, H * dk, T).to(dev), k.reshape(B, H * dk, T).to(dev), v.reshape(B, H * dk, T).to(dev),
                             pu.to(dev), bd.to(dev), keep.to(dev), 1.0 / dk ** 0.5, H)
    assert out.shape == ref.shape
    assert float(out[2].abs().max()) == 0.0                      # fully padded clip: zeros, as the reference
    assert rel_err(out, ref) < 5e-5
    # the same with the position scores computed in the kernel (svb_relpos_attn_pos_fwd: band product + skew through LDS; no bd
    # tensor), q / k / v handed over as equal-pitch slices of one [B, 3D, T] tensor (a fused projection's output)
    qkv = torch.cat([q.reshape(B, H * dk, T), k.reshape(B, H * dk, T), v.reshape(B, H * dk, T)], 1).to(dev)
    D = H * dk
    pt_hi, pt_lo = K.relpos_pos_table(pe.reshape(1, D, T).to(dev), H)
    out2 = K.relpos_attention_pos(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], pu.to(dev), pv.to(dev), pt_hi, pt_lo, keep.to(dev),
                                  1.0 / dk ** 0.5, H)
    assert float(out2[2].abs().max()) == 0.0
    assert rel_err(out2, ref) < 5e-5


def test_relpos_attention_three_way_split_in_the_fp32_mode(dev):
    """`conv_precision: fp32` evaluates the PPG encoder's attention with a THREE-way bf16 operand split (24 mantissa bits, six
    products per pair; svb_attn_set_split3, set by SF.set_precision; the position table carries its third part): against an fp64
    evaluation of the reference's op sequence it must be fp32-class -- within 4x the error of the reference's own fp32 op sequence (stock torch) and at
    least 8x closer than the hi + lo / three-product form of bf16x3 -- for both entry points (position scores handed over / computed in the kernel)."""
    from neuralsvb_amd import functional as SF
    g = torch.Generator().manual_seed(31)
    B, H, dk, T = 2, 2, 64, 130
    q, k, v = (torch.randn(B, H, dk, T, generator=g) for _ in range(3))
    pu, pv = torch.randn(H, dk, generator=g) * 0.5, torch.randn(H, dk, generator=g) * 0.5
    pe = torch.randn(1, H, dk, T, generator=g)
    keep = torch.ones(B, T)
    keep[1, T - 9:] = 0
    qd, kd, vd, ped = q.double(), k.double(), v.double(), pe.double()
    ac = torch.matmul((qd + pu.double()[None, :, :, None]).transpose(-1, -2), kd)
    bd = torch.matmul((qd + pv.double()[None, :, :, None]).transpose(-1, -2), ped)
    x = F.pad(bd, (1, 0)).view(B, H, T + 1, T)[:, :, 1:].reshape(B, H, T, T)
    drop = ~(keep.bool())[:, None, None, :]
    attn = torch.softmax(((ac + x) / dk ** 0.5).masked_fill(drop, -1e300), -1).masked_fill(drop, 0.0)
    ref = torch.matmul(vd, attn.transpose(-1, -2)).reshape(B, H * dk, T)
    D = H * dk
    qkv = torch.cat([q.reshape(B, D, T), k.reshape(B, D, T), v.reshape(B, D, T)], 1).to(dev)
    pt_hi, pt_lo = K.relpos_pos_table(pe.reshape(1, D, T).to(dev), H)
    err = {}
    try:
        for mode in ("bf16x3", "fp32"):
            SF.set_precision(mode)
            out = K.relpos_attention_pos(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], pu.to(dev), pv.to(dev), pt_hi, pt_lo,
                                         keep.to(dev), 1.0 / dk ** 0.5, H)
            err[mode] = ((out.cpu().double() - ref).abs().mean() / ref.abs().mean()).item()
            out = K.relpos_attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], pu.to(dev), bd.float().to(dev), keep.to(dev),
                                     1.0 / dk ** 0.5, H)
            err[mode + "/bd"] = ((out.cpu().double() - ref).abs().mean() / ref.abs().mean()).item()
    finally:
        SF.set_precision("fp32")
    # the yardstick: the reference's own op sequence in fp32 (stock torch) against the same fp64 evaluation
    ac32 = torch.matmul((q + pu[None, :, :, None]).transpose(-1, -2), k)
    bd32 = torch.matmul((q + pv[None, :, :, None]).transpose(-1, -2), pe)
    x32 = F.pad(bd32, (1, 0)).view(B, H, T + 1, T)[:, :, 1:].reshape(B, H, T, T)
    a32 = torch.softmax(((ac32 + x32) / dk ** 0.5).masked_fill(drop, torch.finfo(torch.float32).min), -1).masked_fill(drop, 0.0)
    e32 = ((torch.matmul(v, a32.transpose(-1, -2)).reshape(B, D, T).double() - ref).abs().mean() / ref.abs().mean()).item()
    err["torch fp32"] = e32
    assert err["fp32"] < 4 * e32 and err["fp32"] * 8 < err["bf16x3"], err
    assert err["fp32/bd"] < 4 * e32 and err["fp32/bd"] * 8 < err["bf16x3/bd"], err
    print(err)


@pytest.mark.parametrize("T", [37, 130])
def test_relpos_softmax_matches_espnet_rel_shift(dev, T):
    """svb_relpos_softmax against the reference's op sequence (espnet_transformer_attn.py:125-186): legacy rel_shift via
    pad/view/slice, add, 1/sqrt(dk), masked_fill(min), softmax, masked_fill(0)."""
    g = torch.Generator().manual_seed(T)
    B, H, dk = 2, 3, 16
    ac = torch.randn(B, H, T, T, generator=g) * 3
    bd = torch.randn(B, H, T, T, generator=g) * 3
    keep = torch.ones(B, T)
    keep[1, T - 5:] = 0
    x = F.pad(bd, (1, 0)).view(B, H, T + 1, T)[:, :, 1:].reshape(B, H, T, T)
    scores = (ac + x) / (dk ** 0.5)
    drop = ~(keep.bool())[:, None, None, :]
    ref = torch.softmax(scores.masked_fill(drop, torch.finfo(torch.float32).min), -1).masked_fill(drop, 0.0)
    out = K.relpos_softmax(ac.to(dev), bd.to(dev), keep.to(dev), 1.0 / dk ** 0.5)
    assert (out.cpu() - ref).abs().max().item() < 2e-6


def test_glu_dwconv_bn_swish(dev):
    """Conformer conv module core (conformer/layers.py:47-63, eval): Swish(BN_eval(depthwise_conv1d_k31(GLU(y))))."""
    g_ = torch.Generator().manual_seed(12)
    for B, C, T, k in ((2, 24, 70, 31), (1, 8, 600, 31), (2, 5, 33, 7)):
        y = torch.randn(B, 2 * C, T, generator=g_)
        w = torch.randn(C, 1, k, generator=g_) * 0.2
        b = torch.randn(C, generator=g_) * 0.1
        bw, bb = 1 + 0.1 * torch.randn(C, generator=g_), 0.1 * torch.randn(C, generator=g_)
        mean, var = 0.1 * torch.randn(C, generator=g_), torch.rand(C, generator=g_) + 0.5
        u = F.glu(y, dim=1)
        z = F.batch_norm(F.conv1d(u, w, b, padding=(k - 1) // 2, groups=C), mean, var, bw, bb, False, 0.1, 1e-5)
        ref = z * torch.sigmoid(z)
        out = K.glu_dwconv_bn_swish(y.to(dev), w.to(dev), b.to(dev), bw.to(dev), bb.to(dev), mean.to(dev), var.to(dev), 1e-5)
        assert (out.cpu() - ref).abs().max() < 2e-5, (B, C, T, k)


def test_deferred_wgrad_reduces_equal_immediate(dev):
    """Weight gradients recorded in deferred mode (partials left in the arena, one svb_wgrad_reduce_multi call at the end)
    against the immediate two-launch form: plain, weight-normalised and biased layers, more than one 24-descriptor batch,
    and an arena that has to grow (flush in the middle)."""
    g_ = torch.Generator().manual_seed(31)
    layers = []
    for i in range(27):
        cin, cout, k = [(8, 12, 3), (16, 8, 1), (12, 20, 5)][i % 3]
        wn = i % 2 == 0
        layers.append((torch.randn(2, cin, 60 + i, generator=g_), torch.randn(2, cout, 60 + i, generator=g_),
                       torch.randn(cout, cin, k, generator=g_) * 0.3, (torch.rand(cout, 1, 1, generator=g_) + 0.5) if wn else None, k))

    def run(deferred):
        sinks_all = []
        old_min = K.ARENA_MIN_FLOATS
        K._ARENA.clear()
        K.ARENA_MIN_FLOATS = 1 << 12          # small: the arena must grow while descriptors are pending
        if deferred:
            K.begin_deferred_reduces()
        try:
            for x, dy, v, gn, k in layers:
                sk = (torch.full(v.shape, 0.25, device=dev), torch.full(gn.shape, -0.5, device=dev) if gn is not None else None,
                      torch.full((v.shape[0],), 1.5, device=dev))
                r = K.conv1d_wgrad(dy.to(dev), x.to(dev), k, 1, (k - 1) // 2, 1, 1, v=v.to(dev) if gn is not None else None,
                                   g=gn.to(dev) if gn is not None else None, bf16x3=True, want_bias=True, sinks=sk)
                assert all(t is None for t in r)
                sinks_all.append(sk)
            if deferred:
                assert K._DEFERRED["descs"]            # still pending
        finally:
            if deferred:
                K.flush_deferred_reduces()
            K.ARENA_MIN_FLOATS = old_min
        assert K._DEFERRED is None
        return sinks_all
    now, later = run(False), run(True)
    _join(dev)
    for a_, b_ in zip(now, later):
        for t0, t1 in zip(a_, b_):
            if t0 is not None:
                assert torch.equal(t0, t1)

    # the same sinks used twice in one pass (shared weights): the second record must not race the first
    x, dy, v, gn, k = layers[0]

    def twice(deferred):
        sk = (torch.zeros(v.shape, device=dev), torch.zeros(gn.shape, device=dev), torch.zeros((v.shape[0],), device=dev))
        if deferred:
            K.begin_deferred_reduces()
        try:
            for sgn in (1.0, -0.5):
                K.conv1d_wgrad((dy * sgn).to(dev), x.to(dev), k, 1, (k - 1) // 2, 1, 1, v=v.to(dev), g=gn.to(dev), bf16x3=True,
                               want_bias=True, sinks=sk)
        finally:
            if deferred:
                K.flush_deferred_reduces()
        return sk
    r0, r1 = twice(False), twice(True)
    _join(dev)
    for t0, t1 in zip(r0, r1):
        assert torch.equal(t0, t1)


def _join(dev):
    """Results that went into sinks may have been produced on kernels.WGRAD_STREAM: join before reading them."""
    if dev.type == "cuda":
        if K.WGRAD_STREAM is not None:
            torch.cuda.current_stream().wait_stream(K.WGRAD_STREAM)
        torch.cuda.synchronize()


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("affine", [True, False])
def test_batchnorm_nct_train_groups(dev, groups, affine):
    """csrc/batchnorm.hip, train mode: each batch group normalised with its own statistics, running statistics and the batch
    counter updated group after group -- against nn.BatchNorm1d called once per group (what the reference does: the a2a / p2p
    ways are separate forward calls); forward, running buffers, dx, dgamma, dbeta.  fp32 kernel vs torch fp32: 2e-5."""
    from neuralsvb_amd import functional as SF
    g_ = torch.Generator().manual_seed(31 + groups)
    B, C, T = 6, 40, 37
    x = torch.randn(B, C, T, generator=g_) * 1.7 + 0.4
    dy = torch.randn(B, C, T, generator=g_)
    ref_bn = torch.nn.BatchNorm1d(C, affine=affine, momentum=0.1)
    if affine:
        with torch.no_grad():
            ref_bn.weight.copy_(torch.randn(C, generator=g_))
            ref_bn.bias.copy_(torch.randn(C, generator=g_))
    ref_bn.running_mean.copy_(torch.randn(C, generator=g_))
    ref_bn.running_var.copy_(torch.rand(C, generator=g_) + 0.5)
    import copy
    bn = copy.deepcopy(ref_bn).to(dev)
    xr = x.clone().requires_grad_(True)
    ref = torch.cat([ref_bn(c) for c in xr.chunk(groups, 0)], 0)
    ref.backward(dy)
    xd = x.to(dev).requires_grad_(True)
    y = SF.batch_norm_nct(bn, xd, groups)
    y.backward(dy.to(dev))
    assert (y.detach().cpu() - ref.detach()).abs().max() < 2e-5
    assert (xd.grad.cpu() - xr.grad).abs().max() < 2e-5
    assert (bn.running_mean.cpu() - ref_bn.running_mean).abs().max() < 1e-6
    assert (bn.running_var.cpu() - ref_bn.running_var).abs().max() < 2e-6
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == groups
    if affine:
        assert rel_err(bn.weight.grad, ref_bn.weight.grad) < 2e-5
        assert rel_err(bn.bias.grad, ref_bn.bias.grad) < 2e-5


def test_batchnorm_nct_eval_mask(dev):
    """Eval mode: running statistics, output times the [B,T] mask (the PPG pre-net's `bn(x) * nonpadding`, pe.py:36-40)."""
    from neuralsvb_amd import functional as SF
    g_ = torch.Generator().manual_seed(5)
    B, C, T = 3, 24, 301
    x = torch.randn(B, C, T, generator=g_)
    mask = (torch.rand(B, T, generator=g_) > 0.3).float()
    bn = torch.nn.BatchNorm1d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g_))
        bn.bias.copy_(torch.randn(C, generator=g_))
        bn.running_mean.copy_(torch.randn(C, generator=g_))
        bn.running_var.copy_(torch.rand(C, generator=g_) + 0.5)
        ref = bn(x) * mask[:, None, :]
        ref0 = bn(x)
        import copy
        bd = copy.deepcopy(bn).to(dev)
        assert (SF.batch_norm_nct(bd, x.to(dev), mask=mask.to(dev)).cpu() - ref).abs().max() < 2e-6
        assert (SF.batch_norm_nct(bd, x.to(dev)).cpu() - ref0).abs().max() < 2e-6
    assert int(bd.num_batches_tracked) == 0


def test_gather_segments(dev):
    """Adopted gradients -> their slices of the flat buffer, one launch for all (csrc/optim.hip); more segments than one
    launch carries, an empty one, unaligned lengths; untouched gaps stay as they were."""
    g_ = torch.Generator().manual_seed(2)
    lens = [1, 7, 1000, 0, 33, 4099] + [5] * 60
    srcs = [torch.randn(n, generator=g_).to(dev) for n in lens]
    offs, off = [], 3
    for n in lens:
        offs.append(off)
        off += n + 2
    dst = torch.full((off + 5,), -7.0, device=dev)
    K.gather_segments(srcs, offs, dst)
    ref = torch.full((off + 5,), -7.0)
    for s_, o in zip(srcs, offs):
        ref[o:o + s_.numel()] = s_.cpu()
    assert torch.equal(dst.cpu(), ref)


def test_batchnorm_nct_eval_under_autograd(dev):
    """A BatchNorm1d frozen with .eval() whose input (and affine) still take gradients: forward on the running statistics,
    dx / dgamma / dbeta against torch, with and without the output mask."""
    from neuralsvb_amd import functional as SF
    import copy
    g_ = torch.Generator().manual_seed(9)
    B, C, T = 4, 20, 53
    x = torch.randn(B, C, T, generator=g_)
    dy = torch.randn(B, C, T, generator=g_)
    mask = (torch.rand(B, T, generator=g_) > 0.3).float()
    bn = torch.nn.BatchNorm1d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g_))
        bn.bias.copy_(torch.randn(C, generator=g_))
        bn.running_mean.copy_(torch.randn(C, generator=g_))
        bn.running_var.copy_(torch.rand(C, generator=g_) + 0.5)
    for m in (None, mask):
        rb = copy.deepcopy(bn)
        xr = x.detach().clone().requires_grad_(True)
        ref = rb(xr) if m is None else rb(xr) * m[:, None, :]
        ref.backward(dy)
        bd = copy.deepcopy(bn).to(dev)
        xd = x.detach().clone().to(dev).requires_grad_(True)
        y = SF.batch_norm_nct(bd, xd, mask=None if m is None else m.to(dev))
        y.backward(dy.to(dev))
        assert (y.detach().cpu() - ref.detach()).abs().max() < 2e-6
        assert (xd.grad.cpu() - xr.grad).abs().max() < 2e-6
        assert rel_err(bd.weight.grad, rb.weight.grad) < 1e-5 and rel_err(bd.bias.grad, rb.bias.grad) < 1e-5
        assert torch.equal(bd.running_mean.cpu(), bn.running_mean) and int(bd.num_batches_tracked) == 0


def test_tile_table_is_committed_and_well_formed():
    """The tile configuration of a conv signature comes from the committed table (tools/tune_tiles.py on the MI355X), not from a
    first-sight measurement: the table loads at import, every choice names an existing configuration of its kernel family, the
    three bench workloads' dominant signatures are in it, and a missing table degrades to the on-line tuner, not to an error."""
    import json
    from neuralsvb_amd import kernels as K
    info = K.load_tile_table()
    try:
        assert info["entries"] >= 300 and info["sha256_16"] and info["path"] == "tile_table.json"
        doc = json.load(open(K.TILE_TABLE_PATH))
        assert doc["reps"] >= 20 and doc["arch"] == "gfx950" and doc["configurations"] == K._CFG_NAMES[:len(doc["configurations"])]
        for sig, cfg in K._TUNED.items():
            ncfg = 5 if sig[0] in ("f", "t") or (sig[0] == "taps" and not sig[1]) else len(doc["configurations"])
            assert 1 <= cfg <= ncfg, (sig, cfg)
            med = doc["medians_us"][K._sig_key(sig)]
            assert len(med) == ncfg and med[cfg - 1] <= min(med) + 0.011      # the choice is the measured best (file rounds to 0.01 us)
        # configs[1]: the decoder stack's k=5 conv and the 1x1 res/skip conv at B=32 (two ways stacked) x T=1124
        assert ("qf", 32, 192, 384, 1, 1124, 5, 1, 2, 1, False) in K._TUNED
        assert ("qf", 32, 192, 384, 1, 1124, 1, 1, 0, 1, False) in K._TUNED
        assert K.tuned_choice(("qf", 32, 192, 384, 1, 1124, 5, 1, 2, 1, False), True) == K._TUNED[("qf", 32, 192, 384, 1, 1124, 5, 1, 2, 1, False)]
        assert K.load_tile_table(None)["entries"] == 0 and not K._TUNED
        assert K.tuned_choice(("qf", 1, 2, 3), True) == 0             # no table: the library's heuristic tile, nothing is measured
    finally:
        K.load_tile_table()


def test_tile_choice_for_unseen_batch_shapes_is_the_nearest_table_entry_of_the_family():
    """The reference batches length-sorted clips by a token budget (utils/__init__.py:163-217, tasks/tts/tts.py:57-101): B and T
    change from batch to batch, so an exact-signature table misses on every batch of a real run.  A signature the table does not
    hold takes the choice of the nearest entry (in log B T, log T) of its launch family -- the signature without its batch / length
    fields -- and a family the table has never seen takes the heuristic tile; nothing is measured (no launch, no synchronise),
    and `tile_table_info()` counts what was resolved this way."""
    from neuralsvb_amd import kernels as K
    K.load_tile_table()
    try:
        exact = ("qf", 32, 192, 384, 1, 1124, 5, 1, 2, 1, False)
        short = ("qf", 32, 192, 384, 1, 281, 5, 1, 2, 1, False)
        assert exact in K._TUNED and short in K._TUNED
        launched = []
        near_long = ("qf", 29, 192, 384, 1, 1180, 5, 1, 2, 1, False)           # a token-budget batch near the T = 1124 entry
        assert K._tuned_cfg(near_long, launched.append, K._NCFG_Q) == K._TUNED[exact]
        near_short = ("qf", 34, 192, 384, 1, 300, 5, 1, 2, 1, False)
        assert K._tuned_cfg(near_short, launched.append, K._NCFG_Q) == K._TUNED[short]
        assert K.tuned_choice(near_long, True) == K._TUNED[exact] and not launched
        assert K._TUNED_NEAREST[near_long][1] == exact and K._TUNED_NEAREST[near_short][1] == short
        tq = ("qt", 30, 384, 192, 1, 1200, 1200, 5, 1, 2, 1, False)              # transposed form: B, Tin, Tout are the size fields
        assert K._tuned_cfg(tq, launched.append, K._NCFG_Q) == K._TUNED[("qt", 32, 384, 192, 1, 1124, 1124, 5, 1, 2, 1, False)]
        unknown = ("qf", 32, 200, 392, 1, 1124, 5, 1, 2, 1, False)               # a family the table has never seen
        assert K._tuned_cfg(unknown, launched.append, K._NCFG_Q) == 0 and K.tuned_choice(unknown, True) == 0
        info = K.tile_table_info()
        assert info["online_tuned_signatures"] == 0 and info["nearest_bucket_signatures"] == 3 and not launched
        assert K._family(("qf", 2, 3, 4, 1, 50, 5, 1, 2, 1, False)) == (("qf", 3, 4, 1, 5, 1, 2, 1, False), 2, 50)
    finally:
        K.load_tile_table()


@pytest.mark.parametrize("cfg", [1, 2, 5, 11, 16])
def test_conv1d_single_product_bf16_mode(dev, cfg):
    """`conv_precision: bf16` (svb_conv_set_single_product): the bf16x3 conv entry points with ONE bf16 product per operand pair.
    The result must equal a conv of the bf16-ROUNDED operands accumulated in fp32 (to summation order), i.e. it is plain bf16
    arithmetic -- about 1e-3 off the fp32 conv, where the three-product split is 1e-5 off -- forward (LeakyReLU on the operand load,
    bias, residual) and transposed; and switching the mode off restores the split."""
    from neuralsvb_amd import functional as SF
    g = torch.Generator().manual_seed(cfg)
    B, Cin, Cout, T, k = 2, 48, 72, 300, 5
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) * 0.2
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, T, generator=g)
    bf = lambda t_: t_.to(torch.bfloat16).to(torch.float32)
    ref32 = oops.conv1d(F.leaky_relu(x, 0.1), w, bias, 1, 2) + res
    refbf = oops.conv1d(bf(F.leaky_relu(x, 0.1)), bf(w), bias, 1, 2) + res
    qa, qb = K.weight_pack_q(w.to(dev), None, 1)
    kw = dict(bias=bias.to(dev), in_gate=x.to(dev), in_slope=0.1, residual=res.to(dev), force_cfg=cfg)
    SF.set_precision("bf16")
    try:
        y1 = K.conv1d_forward(x.to(dev), qa, Cout, k, 1, 2, 1, 1, **kw)
        dy = torch.randn(ref32.shape, generator=g)
        d1 = K.conv1d_transposed(dy.to(dev), qb, Cin, T, k, 1, 2, 1, 1, force_cfg=cfg)
    finally:
        SF.set_precision("bf16x3")
    y3 = K.conv1d_forward(x.to(dev), qa, Cout, k, 1, 2, 1, 1, **kw)
    assert rel_err(y1, refbf) < 2e-6                     # = the conv of the rounded operands
    assert 2e-4 < rel_err(y1, ref32) < 1e-2               # plain bf16 accuracy
    assert rel_err(y3, ref32) < 6e-5                      # the split is back
    dref = torch.autograd.grad(oops.conv1d(x.requires_grad_(True), w, None, 1, 2), x, dy)[0]
    drefbf = torch.autograd.grad(oops.conv1d(x, bf(w), None, 1, 2), x, bf(dy))[0]
    assert rel_err(d1, drefbf) < 2e-6 and 2e-4 < rel_err(d1, dref) < 1e-2


@pytest.mark.parametrize("cfg", [0, 2, 13, 15, 16])
@pytest.mark.parametrize("k,dil", [(1, 1), (3, 1), (3, 5), (7, 3), (11, 1)])
def test_self_gated_conv_and_wgrad_variants_equal_the_general_gate(dev, k, dil, cfg):
    """`conv(leaky_relu(x))` hands the kernels the input itself as the gate tensor (in_gate == x; b_gate == b in its weight
    gradient).  Those launches take the self-gated instantiations (forward MODE 5, weight gradient GATED 2: the activation
    derivative from the value just loaded, no second load, no staging registers for it -- reference modules/hifigan/hifigan.py:
    54-61); a gate in a tensor of its own takes the general ones.  Same arithmetic, so both must agree bit for bit -- forward with
    bias + residual on the heuristic tile, a direct-A tile, two tile-walking variants and the 32 x 256 tile, and the weight / bias
    gradient, over 1 / 3 / 7 / 11 taps and dilations."""
    g = torch.Generator().manual_seed(7 * k + dil)
    B, Cin, Cout, T = 2, 48, 40, 333
    pad = dil * (k - 1) // 2
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, generator=g) * 0.2).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    res = torch.randn(B, Cout, T, generator=g).to(dev)
    dy = torch.randn(B, Cout, T, generator=g).to(dev)
    qa, _ = K.weight_pack_q(w, None, 1)
    y_self = K.conv1d_forward(x, qa, Cout, k, 1, pad, dil, 1, bias=bias, in_gate=x, in_slope=0.1, residual=res, force_cfg=cfg)
    y_gen = K.conv1d_forward(x, qa, Cout, k, 1, pad, dil, 1, bias=bias, in_gate=x.clone(), in_slope=0.1, residual=res, force_cfg=cfg)
    assert torch.equal(y_self, y_gen)
    ref = oops.conv1d(F.leaky_relu(x.cpu(), 0.1), w.cpu(), bias.cpu(), 1, pad, dil) + res.cpu()
    assert rel_err(y_self, ref) < 6e-5
    d_self = K.conv1d_wgrad(dy, x, k, 1, pad, dil, 1, b_gate=x, b_slope=0.1, bf16x3=True, want_bias=True)
    d_gen = K.conv1d_wgrad(dy, x, k, 1, pad, dil, 1, b_gate=x.clone(), b_slope=0.1, bf16x3=True, want_bias=True)
    assert torch.equal(d_self[0], d_gen[0]) and torch.equal(d_self[1], d_gen[1])
    xr = x.cpu().clone().requires_grad_(False)
    wr = w.cpu().clone().requires_grad_(True)
    oops.conv1d(F.leaky_relu(xr, 0.1), wr, None, 1, pad, dil).backward(dy.cpu())
    assert rel_err(d_self[0], wr.grad) < 1e-4


# ---- the round-6 kernels at the bench step's own sizes (BASELINE configs[1]: 16 clips x 2 ways = 32 rows of T = 1124 / 562 / 281) -----
BENCH_PW_SHAPES = [(32, 256, 256, 562, 1, 1), (32, 192, 384, 281, 1, 1), (32, 1024, 256, 562, 1, 1), (32, 256, 1536, 1124, 1, 1),
                   (32, 192, 384, 1124, 5, 1), (32, 256, 256, 1124, 3, 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", BENCH_PW_SHAPES)
def test_gemm_conv_at_bench_sizes_properties(gpu_only, shape):
    """The GEMM-form conv kernel (configurations 18..23) at the train step's own shapes, through properties that do not need an
    oracle run of that size: every tile configuration agrees with the tap-table tile 4 (bit for bit for one tap, 2e-6 otherwise);
    linearity in x to rounding; a time shift of the input inside the clip shifts the output; and 64 sampled output elements against
    an fp64 dot product of the same inputs."""
    dev = gpu_only
    B, Cin, Cout, T, k, dil = shape
    g = torch.Generator().manual_seed(Cin + Cout + T + k)
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    x2 = torch.randn(B, Cin, T, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, generator=g) * 0.05).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    qa, _ = K.weight_pack_q(w, None, 1)
    pad = dil * (k - 1) // 2
    run = lambda inp, cfg, **kw: K.conv1d_forward(inp, qa, Cout, k, 1, pad, dil, 1, force_cfg=cfg, **kw)
    ref = run(x, 4, bias=bias)
    for cfg in range(18, 24):
        y = run(x, cfg, bias=bias)
        if k == 1:
            assert torch.equal(y, ref), cfg
        else:
            assert rel_err(y.cpu(), ref.cpu()) < 2e-6, cfg
    y1, y2, y12 = run(x, 21), run(x2, 21), run(x + x2, 21)
    assert rel_err((y1 + y2).cpu(), y12.cpu()) < 2e-5
    # shift by 3 positions: away from the clip's ends the outputs move with the input
    xs = torch.zeros_like(x)
    xs[:, :, 3:] = x[:, :, :-3]
    ys = run(xs, 21)
    m = pad + 3
    assert rel_err(ys[:, :, m + 3:T - m].cpu(), y1[:, :, m:T - m - 3].cpu()) < 1e-6
    # sampled elements against fp64
    idx = torch.randint(0, B * Cout * (T - 2 * pad), (64,), generator=g)
    xc, wc = x.cpu().double(), w.cpu().double()
    for i in idx.tolist():
        b, r = divmod(i, Cout * (T - 2 * pad))
        co, t = divmod(r, T - 2 * pad)
        t += pad
        want = sum((wc[co, :, j] * xc[b, :, t + (j * dil - pad)]).sum() for j in range(k)).item()
        got = y1[b, co, t].item()
        assert abs(got - want) <= 3e-5 * max(1.0, abs(want)), (b, co, t, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(32, 1536, 256, 1124), (32, 3072, 256, 281), (32, 192, 384, 281), (32, 256, 256, 1124)])
def test_pointwise_weight_gradient_at_bench_sizes(gpu_only, shape):
    """The direct-operand 1-tap weight gradient at the step's own shapes: against an fp64 GEMM of the same operands (whole matrix),
    its bias partials against the row sums, and additivity over a split of the batch (two halves sum to the whole, to rounding)."""
    dev = gpu_only
    B, Cin, Cout, T = shape
    g = torch.Generator().manual_seed(Cin + Cout + T)
    x = torch.randn(B, Cin, T, generator=g)
    dy = torch.randn(B, Cout, T, generator=g)
    ref = torch.einsum("bot,bit->oi", dy.double(), x.double())
    xd, dyd = x.to(dev), dy.to(dev)
    dw, db = K.conv1d_wgrad(dyd, xd, 1, 1, 0, 1, 1, bf16x3=True, want_bias=True)
    assert rel_err(dw[:, :, 0].cpu(), ref.float()) < 2e-5
    assert rel_err(db.cpu(), dy.sum((0, 2))) < 1e-5
    h = B // 2
    da = K.conv1d_wgrad(dyd[:h].contiguous(), xd[:h].contiguous(), 1, 1, 0, 1, 1, bf16x3=True)
    dbb = K.conv1d_wgrad(dyd[h:].contiguous(), xd[h:].contiguous(), 1, 1, 0, 1, 1, bf16x3=True)
    assert rel_err((da + dbb).cpu(), dw.cpu()) < 2e-5


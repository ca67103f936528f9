"""NSF-HifiGAN generator / MPD / MSD on the HIP kernels vs golden vectors produced by the unmodified reference
(tests/golden/make_golden.py), with the reference's state_dict layout loaded strictly."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import procedural
from tests.test_oracle_golden import HIFIGAN_CFG

G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = json.load(open(os.path.join(G, "ref_state_keys.json")))


def t(a):
    return torch.from_numpy(np.asarray(a))


def _load(module, name, prefix):
    ref = sorted((k, tuple(s)) for k, s, _ in KEYS[name])
    mine = sorted((k, tuple(v.shape)) for k, v in module.state_dict().items())
    assert mine == ref, set(mine) ^ set(ref)
    module.load_state_dict(procedural.state_dict_for(KEYS[name], prefix=prefix), strict=True)
    return module


def test_state_dict_layouts_match_reference():
    from neuralsvb_amd.modules.hifigan import HifiGanGenerator, MultiPeriodDiscriminator, MultiScaleDiscriminator
    _load(HifiGanGenerator(HIFIGAN_CFG), "HifiGanGenerator", "model_gen.")
    _load(MultiPeriodDiscriminator(), "MultiPeriodDiscriminator", "model_disc.mpd.")
    _load(MultiScaleDiscriminator(), "MultiScaleDiscriminator", "model_disc.msd.")


# Stated tolerances per conv arithmetic (`conv_precision`): fp32 = fp32 MFMA, bf16x3 = bf16 matrix cores with the fp32-class operand
# split.  |wav| <= 1.  Measured on the MI355X in round 5 (profiles/r05_vocoder_parity_bf16x3_first.log): generator waveform
# 5.7e-7 / 4.2e-6 (fp32 / bf16x3), eval-mode logits 3.9e-6 / 1.5e-5, train-mode gradient norms 7.4e-6 / 2.7e-4,
# d(loss)/d(y_hat) samples 5.0e-4 / 5.6e-3 (the feature-matching L1 and LeakyReLU kinks flip single samples on rounding noise).
PREC = ["fp32", "bf16x3"]
WAV_TOL = {"fp32": 2e-5, "bf16x3": 5e-5}
SCORE_TOL = {"fp32": 5e-5, "bf16x3": 1e-4}
GRAD_TOL = {"fp32": 3e-3, "bf16x3": 3e-3}
GY_TOL = {"fp32": 3e-3, "bf16x3": 3e-2}


def _set_precision(precision):
    from neuralsvb_amd import functional as SF
    SF.set_precision(precision)          # (tests/conftest.py's autouse fixture restores fp32 afterwards)


def _slow(dev):
    if dev.type == "cpu" and not os.environ.get("SVB_SLOW_EMU"):
        pytest.skip("full-size vocoder through the lane emulator takes minutes; set SVB_SLOW_EMU=1 (runs on the GPU by default)")


@pytest.mark.parametrize("precision", PREC)
def test_generator_matches_reference_golden(dev, precision):
    _slow(dev)
    _set_precision(precision)
    from neuralsvb_amd.modules.hifigan import HifiGanGenerator
    d = np.load(os.path.join(G, "hifigan_gen.npz"))
    gen = _load(HifiGanGenerator(HIFIGAN_CFG), "HifiGanGenerator", "model_gen.").to(dev).eval()
    with torch.no_grad():
        wav = gen(t(d["mel"]).to(dev), t(d["f0"]).to(dev), rand_ini=t(d["rand_ini"]).to(dev), noise=t(d["noise"]).to(dev))
    assert wav.shape == d["wav"].shape
    # waveform tolerance: 2e-4 abs in fp32, 5e-4 in bf16x3 (|wav| <= 1); the NSF phase differs from the reference's fp32 cumsum
    # by its own rounding
    err = np.abs(wav.cpu().numpy() - d["wav"]).max()
    print(f"[{precision}] generator golden: waveform max abs error {err:.3e}")
    assert err < WAV_TOL[precision]
    # weight-norm folding (vocoders/hifigan.py:29) must not change the output
    gen.remove_weight_norm()
    with torch.no_grad():
        wav2 = gen(t(d["mel"]).to(dev), t(d["f0"]).to(dev), rand_ini=t(d["rand_ini"]).to(dev), noise=t(d["noise"]).to(dev))
    assert (wav2 - wav).abs().max().item() < 2e-5


@pytest.mark.parametrize("precision", PREC)
@pytest.mark.parametrize("name", ["mpd", "msd"])
def test_discriminators_match_reference_golden(dev, name, precision):
    _slow(dev)
    _set_precision(precision)
    from neuralsvb_amd.modules.hifigan import MultiPeriodDiscriminator, MultiScaleDiscriminator
    from tests.golden.make_golden import fmap_stats
    d = np.load(os.path.join(G, "hifigan_disc.npz"))
    if name == "mpd":
        m = _load(MultiPeriodDiscriminator(), "MultiPeriodDiscriminator", "model_disc.mpd.")
    else:
        m = _load(MultiScaleDiscriminator(), "MultiScaleDiscriminator", "model_disc.msd.")
    m = m.to(dev).eval()
    with torch.no_grad():
        rs, gs, fr, fg = m(t(d["y"]).to(dev), t(d["y_hat"]).to(dev))
    worst = 0.0
    for i, (a, b) in enumerate(zip(rs, gs)):
        assert a.shape == d[f"{name}.y_d_r.{i}"].shape
        for got, ref in ((a, d[f"{name}.y_d_r.{i}"]), (b, d[f"{name}.y_d_g.{i}"])):
            # relative bound: the procedural spectral-norm buffers give the MSD scale-0 logits a ~1e13 magnitude
            e = np.abs(got.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, e)
            assert e < SCORE_TOL[precision], (name, i, e)
    shapes = [list(x.shape) + [0] * (4 - x.dim()) for fm in fr for x in fm]
    assert np.array_equal(np.array(shapes), d[f"{name}.fmap_shapes"])
    st = np.stack([fmap_stats(x.cpu()) for fm in fr for x in fm])
    ref_st = d[f"{name}.fmap_r_stats"]
    k_ = SCORE_TOL[precision] / 5e-5
    assert (np.abs(st - ref_st) <= k_ * (5e-5 + 3e-4 * np.abs(ref_st).max(axis=1, keepdims=True))).all()
    print(f"[{precision}] {name} eval-mode logits: worst relative error {worst:.3e}")


def test_small_generator_and_discriminators_autograd(dev):
    """Small-width G / MPD / MSD step (G loss = adversarial + feature matching) vs torch autograd over the CPU oracle."""
    from neuralsvb_amd.modules import hifigan as H
    from oracle import modules_ref as R
    cfg = dict(HIFIGAN_CFG, upsample_initial_channel=32)
    torch.manual_seed(0)
    gen = H.HifiGanGenerator(cfg)
    sd = {k: v.clone() for k, v in gen.state_dict().items()}
    g_ = torch.Generator().manual_seed(1)
    mel = torch.randn(1, 80, 6, generator=g_) - 3
    f0 = 150 + 200 * torch.rand(1, 6, generator=g_)
    ri = torch.rand(1, 9, generator=g_)
    nz = torch.randn(1, 6 * 128, 9, generator=g_)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    wr = R.hifigan_generator(sdr, mel, f0, ri, nz, cfg)
    tgt = torch.randn(wr.shape, generator=g_) * 0.1
    ((wr - tgt) ** 2).mean().backward()
    gen = gen.to(dev).train()
    w = gen(mel.to(dev), f0.to(dev), rand_ini=ri.to(dev), noise=nz.to(dev))
    assert (w.detach().cpu() - wr.detach()).abs().max() < 2e-4
    ((w - tgt.to(dev)) ** 2).mean().backward()
    for k, p in gen.named_parameters():
        gr = sdr[k].grad
        rel = ((p.grad.cpu() - gr).abs().max() / gr.abs().max().clamp_min(1e-10)).item()
        assert rel < 5e-3, (k, rel)


def test_ingraph_mel_with_grad_matches_fused_kernel_and_oracle(dev):
    """D4 with a gradient: the two-GEMM differentiable mel equals the fused STFT kernel and the oracle
    (mel_utils.py:59-76), and d(mel)/d(wav) equals torch autograd through the oracle."""
    from neuralsvb_amd.modules.frontend import MelFrontend
    from oracle import frontend as ofe
    hp = dict(fft_size=512, hop_size=128, win_size=512, audio_num_mel_bins=80, fmin=50, fmax=12000, audio_sample_rate=24000)
    g_ = torch.Generator().manual_seed(3)
    y = (torch.randn(2, 1024, generator=g_) * 0.3).requires_grad_(True)
    ref = ofe.mel_spectrogram_ingraph(y, 512, 128, 512, 80, 50, 12000, 24000)
    dm = torch.randn(ref.shape, generator=g_)
    ref.backward(dm)
    fe = MelFrontend(hp, dev)
    yd = y.detach().to(dev).requires_grad_(True)
    m = fe.mel_spectrogram(yd)
    assert (m.detach().cpu() - ref.detach()).abs().max() < 2e-4
    with torch.no_grad():
        fused = fe.mel_spectrogram(yd.detach())
    assert (fused.cpu() - ref.detach()).abs().max() < 2e-4
    m.backward(dm.to(dev))
    assert ((yd.grad.cpu() - y.grad).abs().max() / y.grad.abs().max()).item() < 2e-3


@pytest.mark.parametrize("precision", PREC)
@pytest.mark.parametrize("name", ["mpd", "msd"])
def test_discriminators_train_mode_losses_and_gradients(dev, name, precision):
    """V3 / V4 in TRAIN mode + V5 against the unmodified reference (tests/golden/make_golden.py:hifigan_train_golden):
    discriminator pass (discriminator_loss -> every parameter gradient), generator pass (generator_loss + feature_loss ->
    d/d y_hat), and the spectral-norm power iteration of MSD scale 0 (u / v buffers after the four forwards)."""
    _slow(dev)
    _set_precision(precision)
    gtol = GRAD_TOL[precision]
    from neuralsvb_amd.modules import hifigan as H
    from tests.golden.make_golden import grad_digest
    d = np.load(os.path.join(G, "hifigan_disc.npz"))
    z = np.load(os.path.join(G, "hifigan_train.npz"))
    if name == "mpd":
        m = _load(H.MultiPeriodDiscriminator(), "MultiPeriodDiscriminator", "model_disc.mpd.")
    else:
        m = _load(H.MultiScaleDiscriminator(), "MultiScaleDiscriminator", "model_disc.msd.")
        sd = m.state_dict()
        for k in z.files:
            if k.startswith("msd.buf0."):
                sd[k[len("msd.buf0."):]].copy_(t(z[k]))
    m = m.to(dev).train()
    y, yh = t(d["y"]).to(dev), t(d["y_hat"]).to(dev)
    m.zero_grad()
    y_d_rs, y_d_gs, _, _ = m(y, yh)
    lr_, lg_ = H.discriminator_loss(y_d_rs, y_d_gs)
    (lr_ + lg_).backward()
    assert np.allclose([lr_.item(), lg_.item()], z[f"{name}.d_loss"], rtol=2e-4 if precision == "fp32" else 1e-3, atol=1e-6), (lr_.item(), lg_.item(), z[f"{name}.d_loss"])
    worst = 0.0
    for k, p in m.named_parameters():
        ref, got = z[f"{name}.dgrad.{k}"], grad_digest(p.grad.cpu())
        rel = abs(got[0] - ref[0]) / max(ref[0], 1e-12)
        worst = max(worst, rel)
        assert rel < gtol, (k, got[0], ref[0])
        rms = ref[0] / np.sqrt(p.numel())
        assert np.abs(got[1:] - ref[1:]).max() <= (gtol / 3e-3) * 2e-2 * max(rms, np.abs(ref[1:]).max()), k
    yh2 = yh.clone().requires_grad_(True)
    y_d_rs, y_d_gs, fmap_rs, fmap_gs = m(y, yh2)
    la, lf = H.generator_loss(y_d_gs), H.feature_loss(fmap_rs, fmap_gs)
    (la + lf).backward()
    assert np.allclose([la.item(), lf.item()], z[f"{name}.g_loss"], rtol=2e-4 if precision == "fp32" else 1e-3, atol=1e-6), (la.item(), lf.item(), z[f"{name}.g_loss"])
    gref = t(z[f"{name}.g_grad_yhat"])
    e_gy = ((yh2.grad.cpu() - gref).abs().max() / gref.abs().max()).item()
    assert e_gy < GY_TOL[precision], e_gy
    if name == "msd":
        sd = m.state_dict()
        for k in z.files:
            if k.startswith("msd.buf1."):
                assert np.abs(sd[k[len("msd.buf1."):]].cpu().numpy() - z[k]).max() < 2e-5, k
    print(f"[{precision}] {name} train mode: worst grad-norm rel err {worst:.3e}, d/d y_hat {e_gy:.3e}")


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("shape", [(128, 1, 15, 1), (128, 128, 41, 4), (64, 256, 5, 1), (1, 96, 3, 1)])
def test_spectral_norm_weight_matches_torch_formulation(dev, shape):
    """SF.spectral_norm_weight (csrc/spectral_norm.hip) against torch.nn.utils.spectral_norm's own formulation in torch ops
    (_SpectralConv1d._weight_torch): normalised weight, updated u / v buffers over two consecutive training-mode calls, the
    eval-mode weight, and the gradient w.r.t. weight_orig (u, v constants of the graph).  fp32 vs fp32: 2e-6 relative."""
    import copy
    from neuralsvb_amd.modules.hifigan import _SpectralConv1d
    cout, cin, k, groups = shape
    torch.manual_seed(7 + cout + k)
    ref = _SpectralConv1d(cin, cout, k, 1, k // 2, groups)
    mod = copy.deepcopy(ref).to(dev)
    g_ = torch.Generator().manual_seed(1)
    for it in range(2):
        ref.train()
        mod.train()
        wr = ref._weight_torch()
        wd = mod._weight()
        dy = torch.randn(wr.shape, generator=g_)
        (gr,) = torch.autograd.grad(wr, ref.weight_orig, dy)
        (gd,) = torch.autograd.grad(wd, mod.weight_orig, dy.to(dev))
        assert _rel(wd, wr) < 2e-6, it
        assert _rel(mod.weight_u, ref.weight_u) < 2e-6 and _rel(mod.weight_v, ref.weight_v) < 2e-6
        assert _rel(gd, gr) < 5e-6, it
    ref.eval()
    mod.eval()
    u0 = mod.weight_u.clone()
    assert _rel(mod._weight(), ref._weight_torch()) < 2e-6
    assert torch.equal(mod.weight_u, u0)                       # eval mode: no power iteration, buffers untouched


@pytest.mark.parametrize("precision", PREC)
@pytest.mark.parametrize("folded", [False, True])
def test_fused_resblock_node_equals_per_conv_nodes(dev, folded, precision):
    """ResBlock1 as ONE autograd node (SF.resblock1: the skip connections' gradient sums ride in the data-gradient convs'
    residual epilogue) against the per-conv nodes it replaces (reference modules/hifigan/hifigan.py:30-67): same kernels on the
    same operands, so output, input gradient and every weight / WeightNorm / bias gradient must be bit-identical -- with
    weight norm and with folded weights, kernel 7 / dilations (1, 3, 5), a sequence that is not a multiple of any tile."""
    import copy
    from neuralsvb_amd.modules import hifigan as H
    _set_precision(precision)
    torch.manual_seed(3)
    blk = H.ResBlock1(None, 24, 7, (1, 3, 5))
    for p in blk.parameters():
        p.data.mul_(8.0)                     # (init_std 0.01 would leave every gradient at rounding level)
    if folded:
        blk.remove_weight_norm()
    ref = copy.deepcopy(blk)
    blk, ref = blk.to(dev).train(), ref.to(dev).train()
    g_ = torch.Generator().manual_seed(5)
    x = torch.randn(2, 24, 157, generator=g_)
    dy = torch.randn(2, 24, 157, generator=g_)
    outs = []
    for m, fused in ((blk, True), (ref, False)):
        H.FUSED_RESBLOCK = fused
        try:
            xd = x.to(dev).requires_grad_(True)
            y = m(xd * 1.0)                # (a non-leaf input, as inside the generator)
            y.backward(dy.to(dev))
        finally:
            H.FUSED_RESBLOCK = True
        outs.append((y.detach().cpu(), xd.grad.cpu(), {k: p.grad.cpu() for k, p in m.named_parameters()}))
    (y1, dx1, g1), (y0, dx0, g0) = outs
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    assert set(g1) == set(g0) and len(g1) == (12 if folded else 18)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k
    assert float(dx0.abs().max()) > 0 and all(float(v.abs().max()) > 0 for v in g0.values())

"""Pins the CPU oracle (oracle/modules_ref.py, oracle/frontend.py) to golden vectors produced by the UNMODIFIED
reference (tests/golden/make_golden.py, run in the build container).  CPU only, no GPU, no reference needed."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import frontend as ofe
from oracle import modules_ref as R
from oracle import procedural

G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = json.load(open(os.path.join(G, "ref_state_keys.json")))
HP = json.load(open(os.path.join(G, "ref_hparams_vae_global_mle_eng.json")))
HIFIGAN_CFG = {"resblock": "1", "upsample_rates": [8, 4, 2, 2], "upsample_kernel_sizes": [16, 8, 4, 4],
               "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
               "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "use_pitch_embed": True,
               "audio_sample_rate": 24000, "hop_size": 128}


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol, what=""):
    a = a.detach().cpu() if isinstance(a, torch.Tensor) else t(a)
    b = t(b)
    err = (a.double() - b.double()).abs().max().item()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol}"


def test_f0_to_coarse_matches_reference():
    d = np.load(os.path.join(G, "f0_to_coarse.npz"))
    assert np.array_equal(ofe.f0_to_coarse(d["f0"]), d["coarse_np"])
    assert np.array_equal(ofe.f0_to_coarse(torch.from_numpy(d["f0"].astype(np.float32))).numpy(), d["coarse_torch"])


def test_mle_svb_vae_matches_reference():
    d = np.load(os.path.join(G, "vae_mle.npz"))
    sd = procedural.state_dict_for(KEYS["MleSVBVAE"], prefix="model.")
    with torch.no_grad():
        ret, ca, _ = R.mle_svb_vae(sd, t(d["mels"]), t(d["prof_mels"]), t(d["pitch"]), t(d["prof_pitch"]), t(d["spk"]),
                                   t(d["a2p_alignment"]), ["a2a", "p2p", "a2p"], t(d["eps_a2a"]), t(d["eps_p2p"]), HP,
                                   training=True)
    for k in ("h_pitch", "h_content", "h_style", "tgt_nonpadding"):
        close(ca[k], d[f"cond_a.{k}"], 2e-5, k)
    for way in ("a2a", "p2p"):
        for k in ("mel_out", "kl", "m_q", "logs_q", "z_q", "x_mask_sqz"):
            close(ret[way][k], d[f"{way}.{k}"], 2e-4 if k in ("z_q", "m_q", "logs_q") else 5e-5, f"{way}.{k}")
    close(ret["a2p"]["mel_out"], d["a2p.mel_out"], 5e-5, "a2p.mel_out")
    close(ret["a2p"]["mle"], d["a2p.mle"], 5e-5, "a2p.mle")
    with torch.no_grad():
        sm = R.ssim_map(t(d["a2a.mel_out"])[:, None] + 6.0, t(d["mels"])[:, None] + 6.0)
    close(sm, d["loss.ssim_map_a2a"], 2e-5, "ssim")


def test_mel_discriminator_matches_reference():
    d = np.load(os.path.join(G, "mel_disc.npz"))
    sd = procedural.state_dict_for(KEYS["Discriminator"], prefix="mel_disc.")
    with torch.no_grad():
        y, hs = R.mel_discriminator(sd, t(d["x"]), d["starts"])
    close(y, d["y"], 2e-5, "y")
    from tests.golden.make_golden import fmap_stats
    for i, h in enumerate(hs):
        np.testing.assert_allclose(fmap_stats(h), d["h_stats"][i], atol=2e-5, rtol=1e-4)


def test_hifigan_generator_matches_reference():
    d = np.load(os.path.join(G, "hifigan_gen.npz"))
    sd = procedural.state_dict_for(KEYS["HifiGanGenerator"], prefix="model_gen.")
    with torch.no_grad():
        wav = R.hifigan_generator(sd, t(d["mel"]), t(d["f0"]), t(d["rand_ini"]), t(d["noise"]), HIFIGAN_CFG)
    close(wav, d["wav"], 2e-5, "wav")


def test_spec2wav_plugin_golden_matches_oracle_generator():
    """The reference's vocoder plugin run (vocoders/hifigan.py:17-69, tests/golden/spec2wav.npz) is the generator applied to
    one clip with weight norm folded: the oracle's generator restatement reproduces it from the recorded draws."""
    d = np.load(os.path.join(G, "spec2wav.npz"))
    sd = procedural.state_dict_for(KEYS["HifiGanGenerator"], prefix="model_gen.")
    with torch.no_grad():
        wav = R.hifigan_generator(sd, t(d["mel"]).T[None], t(d["f0"])[None], t(d["rand_ini"]), t(d["noise"]), HIFIGAN_CFG)
    close(wav.reshape(-1), d["wav"], 2e-5, "spec2wav")


@pytest.mark.parametrize("name", ["mpd", "msd"])
def test_hifigan_discriminators_match_reference(name):
    d = np.load(os.path.join(G, "hifigan_disc.npz"))
    from tests.golden.make_golden import fmap_stats
    if name == "mpd":
        sd = procedural.state_dict_for(KEYS["MultiPeriodDiscriminator"], prefix="model_disc.mpd.")
        fn = R.multi_period_disc
    else:
        sd = procedural.state_dict_for(KEYS["MultiScaleDiscriminator"], prefix="model_disc.msd.")
        fn = R.multi_scale_disc
    with torch.no_grad():
        rs, gs, fr, fg = fn(sd, t(d["y"]), t(d["y_hat"]))
    for i, (a, b) in enumerate(zip(rs, gs)):
        close(a, d[f"{name}.y_d_r.{i}"], 3e-5 * max(1.0, np.abs(d[f"{name}.y_d_r.{i}"]).max()), f"{name}.r{i}")
        close(b, d[f"{name}.y_d_g.{i}"], 3e-5 * max(1.0, np.abs(d[f"{name}.y_d_g.{i}"]).max()), f"{name}.g{i}")
    st = np.stack([fmap_stats(x) for fm in fr for x in fm])
    np.testing.assert_allclose(st, d[f"{name}.fmap_r_stats"], atol=3e-5, rtol=2e-4)


def test_frontend_restatement_against_independent_implementations():
    """The librosa-0.8.0 front-end (`librosa.stft`, `librosa.filters.mel`; pinned by the reference's Requirements.txt:41, not
    vendored, no vectors in the reference) is restated in oracle/frontend.py from the published algorithm.  Two independent
    anchors exist in this image: torch.stft with librosa's conventions (centred, reflect padding, periodic Hann) for the STFT,
    and the example of librosa's own documentation for the Slaney filterbank (`librosa.filters.mel(22050, 2048)` prints
    `[[0., 0.016, ...` and with `fmax=8000` `[[0., 0.02, ...`) -- the latter only to the printed precision."""
    import torch
    from oracle import frontend as ofe
    rng = np.random.RandomState(0)
    for n_fft, hop, n in ((512, 128, 5000), (2048, 512, 9000), (1024, 256, 1024)):
        y = rng.randn(n).astype(np.float32)
        mine = ofe.librosa_stft(y, n_fft, hop, n_fft)
        ref = torch.stft(torch.from_numpy(y), n_fft, hop, n_fft, window=torch.hann_window(n_fft, periodic=True), center=True,
                         pad_mode="reflect", return_complex=True).numpy()
        assert mine.shape == ref.shape == (1 + n_fft // 2, 1 + n // hop)
        assert np.abs(mine - ref).max() <= 2e-6 * np.abs(ref).max()
    fb = ofe.librosa_mel_filterbank(22050, 2048)
    assert fb.shape == (128, 1025) and fb.dtype == np.float32
    assert round(float(fb[0, 1]), 3) == 0.016 and fb[0, 0] == 0 and fb[1, 0] == 0 and fb[-1, 0] == 0
    assert round(float(ofe.librosa_mel_filterbank(22050, 2048, fmax=8000)[0, 1]), 2) == 0.02
    # Slaney normalisation: every triangle has (continuous) area 1 in Hz, i.e. height 2 / (f_hi - f_lo); adjacent triangles
    # cross at their feet, so the columns between the first and last centre carry one or two filters only
    fb80 = ofe.librosa_mel_filterbank(24000, 512, 80, 50, 12000)
    nz = (fb80 > 0).sum(0)
    assert nz.max() <= 2 and fb80.min() >= 0 and (fb80.sum(1) > 0).all()

"""Generate the committed golden vectors by running the UNMODIFIED reference (/root/reference) on CPU.

Build-container only (the reference is not present on the GPU box).  Usage:  python tests/golden/make_golden.py
Writes tests/golden/ref_state_keys.json (state_dict key/shape lists of the reference modules at the real
hparams) and tests/golden/*.npz (inputs, injected randomness, reference outputs).  Weights are procedural
(oracle/procedural.py), so no state_dict is stored.  Every case that lets the reference draw random numbers seeds torch's global
generator first (CASE_SEEDS), so a second run of this script on the unmodified reference writes byte-identical files.
"""
import contextlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import procedural, ref_shims  # noqa: E402
sys.path.insert(0, HERE)
from detnpz import savez_det  # noqa: E402

HIFIGAN_CFG = {  # hop-128 NSF configuration (working assumption, configurable)
    "resblock": "1", "upsample_rates": [8, 4, 2, 2], "upsample_kernel_sizes": [16, 8, 4, 4],
    "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "use_pitch_embed": True,
    "audio_sample_rate": 24000, "hop_size": 128,
}


@contextlib.contextmanager
def record_rng(store):
    """Record every torch.randn_like / torch.rand / torch.randn draw the reference makes (injected later)."""
    o_rl, o_r, o_rn = torch.randn_like, torch.rand, torch.randn

    def rl(*a, **k):
        t = o_rl(*a, **k); store.append(("randn_like", t.clone())); return t

    def r(*a, **k):
        t = o_r(*a, **k); store.append(("rand", t.clone())); return t

    def rn(*a, **k):
        t = o_rn(*a, **k); store.append(("randn", t.clone())); return t
    torch.randn_like, torch.rand, torch.randn = rl, r, rn
    try:
        yield
    finally:
        torch.randn_like, torch.rand, torch.randn = o_rl, o_r, o_rn


def keys_of(module):
    return [[k, list(v.shape), str(v.dtype)] for k, v in module.state_dict().items()]


def load_procedural(module, prefix):
    ks = keys_of(module)
    sd = procedural.state_dict_for(ks, prefix=prefix)
    module.load_state_dict(sd, strict=True)
    return ks


def fmap_stats(t):
    t = t.detach().double()
    flat = t.flatten()
    idx = torch.linspace(0, flat.numel() - 1, 16).long()
    return np.concatenate([[t.mean().item(), t.abs().mean().item(), (t * t).mean().item()], flat[idx].numpy()])


def make_vae_inputs(B=2, T=64, lens=(64, 52), seed=0):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in ("mels", "prof_mels"):
        m = torch.randn(B, T, 80, generator=g) * 0.8 - 3.0
        for b, L in enumerate(lens):
            m[b, L:] = 0.0
        out[name] = m
    for name in ("pitch", "prof_pitch"):
        p = torch.randint(1, 256, (B, T), generator=g)
        for b, L in enumerate(lens):
            p[b, L:] = 0
        out[name] = p
    spk = torch.randn(B, 256, generator=g)
    out["spk"] = spk / spk.norm(dim=-1, keepdim=True)
    al = torch.stack([torch.clamp((torch.arange(T) * 0.97 + 1.5 * torch.sin(torch.arange(T) / 7.0)).round().long(), 0, L - 1)
                      for L in lens])
    out["a2p_alignment"] = al
    return out


CASE_SEEDS = {"vae_mle": 4101, "vae_mle_b16": 4102, "vae_mle_b16_grad": 4103, "hifigan_gen": 4104, "spec2wav": 4105,
              "hifigan_train": 4106}

BENCH_LENS = (1124, 1124, 1100, 1124, 1056, 1124, 1003, 1124, 1124, 960, 1124, 1088, 1124, 1124, 912, 1124)
BENCH_FRAME_STRIDE = 8


def vae_bench_shape(model):
    """configs[1] shape (B=16 clips x T=1124 frames, ragged lengths): the reference forward of all three ways.
    Inputs are regenerated from the seed by the test (make_vae_inputs is deterministic); the file keeps the injected
    N(0,1) draws, input checksums, every BENCH_FRAME_STRIDE-th frame of each way's mel_out, the per-clip latent
    statistics and the scalar terms."""
    inp = make_vae_inputs(B=16, T=1124, lens=BENCH_LENS, seed=21)
    rec = []
    torch.manual_seed(CASE_SEEDS["vae_mle_b16"])
    with record_rng(rec), torch.no_grad():
        out = model(amateur_mel=inp["mels"], prof_mel=inp["prof_mels"], amateur_pitch=inp["pitch"],
                    prof_pitch=inp["prof_pitch"], amateur_spk_id=inp["spk"], prof_spk_id=inp["spk"],
                    a2p_alignment=inp["a2p_alignment"], p2a_alignment=None, infer=False,
                    concurrent_ways=["a2a", "p2p", "a2p"], disable_map=False)
    assert [k for k, _ in rec] == ["randn_like", "randn_like"]
    save = {"eps_a2a": rec[0][1].numpy(), "eps_p2p": rec[1][1].numpy(), "lens": np.array(BENCH_LENS),
            "frame_stride": np.array(BENCH_FRAME_STRIDE),
            "input_checksum": np.array([float(inp[k].double().sum()) for k in ("mels", "prof_mels", "pitch", "prof_pitch", "spk",
                                                                                "a2p_alignment")])}
    for way in ("a2a", "p2p", "a2p"):
        save[f"{way}.mel_out"] = out[way]["mel_out"][:, ::BENCH_FRAME_STRIDE].numpy()
        save[f"{way}.mel_out_abs_mean"] = np.array(float(out[way]["mel_out"].double().abs().mean()))
        for k in ("kl", "m_q", "logs_q", "z_q", "mle"):
            if k in out[way] and isinstance(out[way][k], torch.Tensor):
                save[f"{way}.{k}"] = out[way][k].numpy()
    save.update(vae_bench_shape_gradients(model, inp))
    savez_det(os.path.join(HERE, "vae_mle_b16.npz"), **save)


BENCH_GRAD_PARAMS = ["vae_model.decoder.wn.in_layers.0.weight_v", "vae_model.decoder.wn.in_layers.3.weight_g",
                     "vae_model.decoder.wn.cond_layer.weight_v", "vae_model.decoder.wn.res_skip_layers.1.weight_v",
                     "vae_model.encoder.pre_net.0.weight", "vae_model.encoder.wn.in_layers.7.weight_v",
                     "vae_model.encoder.wn.cond_layer.weight_g", "vae_model.decoder.pre_net.0.weight",
                     "vae_model.decoder.out_proj.bias", "vae_model.encoder.poolings.3.weight", "vae_model.g_pre_net.0.weight",
                     "pitch_embed.weight", "pitch_encoder.conv.1.conv.conv.weight", "upsample_layer.0.1.weight",
                     "encoded_embed_proj.weight", "spk_embed_proj.weight"]


def vae_bench_shape_gradients(model, inp):
    """Backward tile / split-K choices are made per shape, so gradients are pinned AT the bench shape too: the reference's
    forward of the two training ways (a2a, p2p; phase 2) at B=16 x T=1124 in train mode, the scalar
    sum_way [ mean |mel_out - target| over all elements + kl ], backward, and a digest (l2 norm + 24 samples) of the gradient
    of a dozen parameters spread over the decoder / encoder gated stacks, their conditioning layers and the condition path."""
    rec = []
    torch.manual_seed(CASE_SEEDS["vae_mle_b16_grad"])
    model.zero_grad()
    with record_rng(rec):
        out = model(amateur_mel=inp["mels"], prof_mel=inp["prof_mels"], amateur_pitch=inp["pitch"],
                    prof_pitch=inp["prof_pitch"], amateur_spk_id=inp["spk"], prof_spk_id=inp["spk"],
                    a2p_alignment=inp["a2p_alignment"], p2a_alignment=None, infer=False,
                    concurrent_ways=["a2a", "p2p"], disable_map=True)
    assert [k for k, _ in rec] == ["randn_like", "randn_like"]
    tgt = {"a2a": inp["mels"], "p2p": inp["prof_mels"]}
    terms = {}
    loss = 0.0
    for way in ("a2a", "p2p"):
        terms[f"{way}.l1"] = (out[way]["mel_out"] - tgt[way]).abs().mean()
        terms[f"{way}.kl"] = out[way]["kl"]
        loss = loss + terms[f"{way}.l1"] + terms[f"{way}.kl"]
    loss.backward()
    params = dict(model.named_parameters())
    save = {"grad.eps_a2a": rec[0][1].numpy(), "grad.eps_p2p": rec[1][1].numpy(),
            "grad.terms": np.array([float(terms[k]) for k in ("a2a.l1", "a2a.kl", "p2p.l1", "p2p.kl")]),
            "grad.params": np.array(BENCH_GRAD_PARAMS)}
    for k in BENCH_GRAD_PARAMS:
        save[f"grad.{k}"] = grad_digest(params[k].grad)
    model.zero_grad()
    return save


def spec2wav_golden(gen):
    """V6: the reference's vocoder PLUGIN (vocoders/hifigan.py:17-69) end to end -- a checkpoint directory in its own layout
    (config.yaml + model_ckpt_steps_<N>.ckpt holding state_dict['model_gen']) is written with the procedural generator
    weights, `HifiGAN()` loads it through load_model (strict load, weight norm folded, eval) and `spec2wav(mel[T,80], f0=f0[T])`
    runs on one clip.  The three NSF draws are recorded for injection."""
    import tempfile
    import yaml
    from utils.hparams import hparams as ref_hp
    with tempfile.TemporaryDirectory() as d:
        cfg = dict(HIFIGAN_CFG)
        with open(os.path.join(d, "config.yaml"), "w") as f:
            yaml.safe_dump(cfg, f)
        torch.save({"state_dict": {"model_gen": gen.state_dict()}}, os.path.join(d, "model_ckpt_steps_7.ckpt"))
        old = {k: ref_hp.get(k) for k in ("vocoder_ckpt", "profile_infer", "vocoder_denoise_c")}
        ref_hp.update(vocoder_ckpt=d, profile_infer=False, vocoder_denoise_c=0.0)
        try:
            from vocoders.hifigan import HifiGAN
            voc = HifiGAN()
            g = torch.Generator().manual_seed(17)
            T = 40
            mel = (torch.randn(T, 80, generator=g) * 0.8 - 3.0).numpy()
            f0 = (110 + 260 * torch.rand(T, generator=g)).numpy()
            f0[5:9] = 0.0
            f0[33:] = 0.0
            rec = []
            torch.manual_seed(CASE_SEEDS["spec2wav"])
            with record_rng(rec):
                wav = voc.spec2wav(mel, f0=f0)
            assert [k for k, _ in rec] == ["rand", "randn_like", "randn_like"], [k for k, _ in rec]
        finally:
            for k, v in old.items():
                if v is None:
                    ref_hp.pop(k, None)
                else:
                    ref_hp[k] = v
    assert wav.shape == (T * HIFIGAN_CFG["hop_size"],) and wav.dtype == np.float32
    savez_det(os.path.join(HERE, "spec2wav.npz"), mel=mel, f0=f0, rand_ini=rec[0][1].numpy(), noise=rec[1][1].numpy(),
                        wav=wav)


def grad_digest(t):
    """[l2 norm, 24 evenly spaced elements] of a gradient tensor (float64)."""
    g = t.detach().double().flatten()
    idx = torch.linspace(0, g.numel() - 1, min(24, g.numel())).long()
    return np.concatenate([[g.norm().item()], g[idx].numpy()])


def hifigan_train_golden(mpd, msd, y, yh):
    """V3/V4 in TRAIN mode + V5: one discriminator pass (discriminator_loss, backward -> parameter gradients) and one
    generator pass (generator_loss + feature_loss, backward -> d/d y_hat), reference modules/hifigan/hifigan.py:202-365.
    MSD scale 0 runs its spectral-norm power iteration on every forward (two per MSD call); the u / v buffers are first
    converged with 10 train-mode forwards (procedural u, v start far from the dominant singular pair) and stored."""
    from modules.hifigan.hifigan import discriminator_loss, feature_loss, generator_loss
    mpd.train(); msd.train()
    torch.manual_seed(CASE_SEEDS["hifigan_train"])
    with torch.no_grad():
        for _ in range(10):
            msd(y, yh)
    save = {}
    for k, v in msd.state_dict().items():
        if k.endswith("weight_u") or k.endswith("weight_v") and v.dim() == 1:
            save[f"msd.buf0.{k}"] = v.numpy().copy()
    for name, d in (("mpd", mpd), ("msd", msd)):
        d.zero_grad()
        y_d_rs, y_d_gs, _, _ = d(y, yh.detach())
        lr_, lg_ = discriminator_loss(y_d_rs, y_d_gs)
        (lr_ + lg_).backward()
        save[f"{name}.d_loss"] = np.array([lr_.item(), lg_.item()])
        for k, p_ in d.named_parameters():
            save[f"{name}.dgrad.{k}"] = grad_digest(p_.grad)
        yh2 = yh.clone().requires_grad_(True)
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = d(y, yh2)
        la, lf = generator_loss(y_d_gs), feature_loss(fmap_rs, fmap_gs)
        (la + lf).backward()
        save[f"{name}.g_loss"] = np.array([la.item(), lf.item()])
        save[f"{name}.g_grad_yhat"] = yh2.grad.numpy()
    for k, v in msd.state_dict().items():
        if k.endswith("weight_u") or k.endswith("weight_v") and v.dim() == 1:
            save[f"msd.buf1.{k}"] = v.numpy().copy()
    savez_det(os.path.join(HERE, "hifigan_train.npz"), **save)


def main():
    torch.manual_seed(1234)
    torch.set_num_threads(8)
    hp = ref_shims.set_reference_hparams()
    keys = {}
    resolved = {k: v for k, v in hp.items() if isinstance(v, (int, float, str, bool, list, dict)) or v is None}
    with open(os.path.join(HERE, "ref_hparams_vae_global_mle_eng.json"), "w") as f:
        json.dump(resolved, f, indent=0, sort_keys=True)

    # ---------------- pitch bins (D3) ----------------
    from utils.pitch_utils import f0_to_coarse
    rng = np.random.RandomState(0)
    f0 = np.concatenate([np.zeros(16), rng.uniform(30, 1400, 2000), [50.0, 1100.0, 700.0, 1e-3]])
    savez_det(os.path.join(HERE, "f0_to_coarse.npz"), f0=f0, coarse_np=f0_to_coarse(f0.copy()),
                        coarse_torch=f0_to_coarse(torch.from_numpy(f0.astype(np.float32))).numpy())

    # ---------------- MleSVBVAE (M1-M8), real dims ----------------
    from modules.voice_conversion.svb_vae import MleSVBVAE
    model = MleSVBVAE(70)
    keys["MleSVBVAE"] = load_procedural(model, "model.")
    model.train()
    model.vc_asr.eval()
    inp = make_vae_inputs()
    rec = []
    torch.manual_seed(CASE_SEEDS["vae_mle"])
    with record_rng(rec):
        out = model(amateur_mel=inp["mels"], prof_mel=inp["prof_mels"], amateur_pitch=inp["pitch"],
                    prof_pitch=inp["prof_pitch"], amateur_spk_id=inp["spk"], prof_spk_id=inp["spk"],
                    a2p_alignment=inp["a2p_alignment"], p2a_alignment=None, infer=False,
                    concurrent_ways=["a2a", "p2p", "a2p"], disable_map=False)
    conds = model.prepare_condition(inp["mels"], inp["pitch"], spk_ids=inp["spk"])
    save = {k: v.numpy() for k, v in inp.items()}
    assert [k for k, _ in rec] == ["randn_like", "randn_like"], [k for k, _ in rec]
    save["eps_a2a"], save["eps_p2p"] = rec[0][1].numpy(), rec[1][1].numpy()
    for way in ("a2a", "p2p", "a2p"):
        for k, v in out[way].items():
            if isinstance(v, torch.Tensor):
                save[f"{way}.{k}"] = v.detach().numpy()
    for k in ("h_pitch", "h_content", "h_style", "tgt_nonpadding"):
        save[f"cond_a.{k}"] = conds[k].detach().numpy()
    # mel losses (L1): tasks/tts/fs2.py:158-175 call modules.commons.ssim.ssim on [B,1,T,80]+6
    from modules.commons.ssim import ssim
    mo, tg = out["a2a"]["mel_out"].detach(), inp["mels"]
    save["loss.ssim_map_a2a"] = ssim(mo[:, None] + 6.0, tg[:, None] + 6.0, size_average=False).numpy()
    savez_det(os.path.join(HERE, "vae_mle.npz"), **save)
    vae_bench_shape(model)

    # ---------------- mel discriminator (G1), eval mode (Dropout2d off), fixed windows ----------------
    from modules.fastspeech.multi_window_disc import Discriminator
    disc = Discriminator(time_lengths=[32, 64, 128], freq_length=80, hidden_size=128, kernel=(3, 3), cond_size=0,
                         norm_type="in", reduction="stack")
    keys["Discriminator"] = load_procedural(disc, "mel_disc.")
    disc.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 140, 80, generator=g) * 0.8 - 3.0
    x[1, 131:] = 0.0
    starts = [[7, 7], [40, 40], [3, 3]]
    o = disc(x, None, start_frames_wins=[list(s) for s in starts])
    savez_det(os.path.join(HERE, "mel_disc.npz"), x=x.numpy(), starts=np.array(starts), y=o["y"].detach().numpy(),
                        h_stats=np.stack([fmap_stats(h) for h in o["h"]]))

    # ---------------- NSF-HifiGAN generator (V1, V2) ----------------
    from modules.hifigan.hifigan import HifiGanGenerator, MultiPeriodDiscriminator, MultiScaleDiscriminator
    gen = HifiGanGenerator(HIFIGAN_CFG)
    keys["HifiGanGenerator"] = load_procedural(gen, "model_gen.")
    gen.eval()
    g = torch.Generator().manual_seed(9)
    mel = torch.randn(2, 80, 12, generator=g) * 0.8 - 3.0
    f0 = 120 + 300 * torch.rand(2, 12, generator=g)
    f0[0, 3:5] = 0.0
    f0[1, 9:] = 0.0
    rec = []
    torch.manual_seed(CASE_SEEDS["hifigan_gen"])
    with record_rng(rec), torch.no_grad():
        wav = gen(mel, f0)
    kinds = [k for k, _ in rec]
    assert kinds == ["rand", "randn_like", "randn_like"], kinds
    savez_det(os.path.join(HERE, "hifigan_gen.npz"), mel=mel.numpy(), f0=f0.numpy(), rand_ini=rec[0][1].numpy(),
                        noise=rec[1][1].numpy(), wav=wav.numpy())

    spec2wav_golden(gen)

    # ---------------- MPD / MSD (V3, V4), eval mode (no spectral-norm power iteration) ----------------
    mpd, msd = MultiPeriodDiscriminator(), MultiScaleDiscriminator()
    keys["MultiPeriodDiscriminator"] = load_procedural(mpd, "model_disc.mpd.")
    keys["MultiScaleDiscriminator"] = load_procedural(msd, "model_disc.msd.")
    mpd.eval(); msd.eval()
    g = torch.Generator().manual_seed(13)
    y = torch.tanh(torch.randn(2, 1, 2101, generator=g))
    yh = torch.tanh(torch.randn(2, 1, 2101, generator=g))
    save = dict(y=y.numpy(), y_hat=yh.numpy())
    with torch.no_grad():
        for name, d in (("mpd", mpd), ("msd", msd)):
            y_d_rs, y_d_gs, fmap_rs, fmap_gs = d(y, yh)
            for i, (a, b) in enumerate(zip(y_d_rs, y_d_gs)):
                save[f"{name}.y_d_r.{i}"], save[f"{name}.y_d_g.{i}"] = a.numpy(), b.numpy()
            save[f"{name}.fmap_r_stats"] = np.stack([fmap_stats(t) for fm in fmap_rs for t in fm])
            save[f"{name}.fmap_g_stats"] = np.stack([fmap_stats(t) for fm in fmap_gs for t in fm])
            save[f"{name}.fmap_shapes"] = np.array([list(t.shape) + [0] * (4 - t.dim()) for fm in fmap_rs for t in fm])
    savez_det(os.path.join(HERE, "hifigan_disc.npz"), **save)
    hifigan_train_golden(mpd, msd, y, yh)

    with open(os.path.join(HERE, "ref_state_keys.json"), "w") as f:
        json.dump(keys, f)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()

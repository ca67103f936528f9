"""Synthetic fixed-length paired clips in the reference's binary dataset format (SURVEY §8d "Synthetic inputs").

There is no dataset and no pretrained checkpoint in the build environment, so benchmarks, smoke() and the
step-level tests feed the *unmodified driver path* (tasks/run.py -> Trainer -> task -> dataset) with:
  clip  = sum of harmonics of a vibrato f0 contour + noise, amplitude <= 0.9;
  prof  = the same contour under a smooth +-3 % time warp (so a2p_f0_alignment is known analytically);
  mel   = the front-end under test (HIP `svb_stft_mel` on the GPU; tests may inject the numpy oracle);
  pitch = f0_to_coarse of the analytic f0 (numpy semantics, as the binarizer does).
"""
import json
import os

import numpy as np
import torch

from . import pitch_utils
from .indexed_datasets import IndexedDatasetBuilder


def f0_contour(t, base=220.0):
    f0 = base * 2 ** ((3 * np.sin(2 * np.pi * 0.25 * t) + 0.3 * np.sin(2 * np.pi * 5.5 * t)) / 12)
    gap = (np.floor(t / 0.5) % 10) == 7                     # ~10 % unvoiced gaps
    return np.where(gap, 0.0, f0)


def make_clip(seconds, sr, seed, warp=False, base=220.0):
    rng = np.random.RandomState(seed)
    n = int(round(seconds * sr))
    t = np.arange(n) / sr
    tw = t + (0.03 * seconds / (2 * np.pi)) * np.sin(2 * np.pi * t / seconds) if warp else t
    f0 = f0_contour(tw, base)
    phase = 2 * np.pi * np.cumsum(np.where(f0 > 0, f0, 0.0)) / sr
    nh = 1 + seed % 9
    w = sum(np.sin(h * phase) / h for h in range(1, nh + 1)) * (f0 > 0)
    noise = rng.randn(n)
    noise = np.convolve(noise, np.ones(8) / 8, mode="same") * 10 ** (-30 / 20) * 3
    w = 0.9 * w / max(1e-6, np.abs(w).max()) * 0.8 + noise
    return np.clip(w, -0.9, 0.9).astype(np.float32), tw


def mel_fn_hip(hp, device):
    """Front-end under test: the HIP STFT+mel kernel (offline mode, D2)."""
    from .. import kernels as K
    from ..modules.frontend import mel_filterbank, hann_window
    basis = mel_filterbank(hp["audio_sample_rate"], hp["fft_size"], hp["audio_num_mel_bins"], hp["fmin"], hp["fmax"]).to(device)
    win = hann_window(hp["win_size"], hp["fft_size"]).to(device)

    def fn(wavs):                                            # [B, N] numpy -> [B, T, 80] numpy
        x = torch.from_numpy(wavs).to(device)
        return K.stft_mel(x, win, basis, hp["fft_size"], hp["hop_size"], 0, 1e-10).cpu().numpy()
    return fn


def build_items(n_items, seconds, hp, mel_fn, seed0=0):
    """seconds: one clip length for all items, or a sequence (cycled) of per-item lengths (ragged batches)."""
    sr, hop = hp["audio_sample_rate"], hp["hop_size"]
    ragged = not np.isscalar(seconds)
    secs = [float(seconds[i % len(seconds)]) for i in range(n_items)] if ragged else [float(seconds)] * n_items
    wa, wp, twp = [], [], []
    for i in range(n_items):
        a, _ = make_clip(secs[i], sr, seed0 + i, warp=False, base=180.0 + 15 * (i % 8))
        p, tw = make_clip(secs[i], sr, seed0 + i, warp=True, base=180.0 + 15 * (i % 8))
        wa.append(a); wp.append(p); twp.append(tw)
    if ragged:
        mel_a, mel_p = [mel_fn(w[None])[0] for w in wa], [mel_fn(w[None])[0] for w in wp]
    else:
        mel_a, mel_p = mel_fn(np.stack(wa)), mel_fn(np.stack(wp))
    items = []
    rng = np.random.RandomState(seed0 + 77)
    for i in range(n_items):
        T = mel_a[i].shape[0]
        tt = (np.arange(T) * hop) / sr
        f0_a = f0_contour(tt, 180.0 + 15 * (i % 8))
        idx = np.minimum(np.arange(T) * hop, len(twp[i]) - 1)
        f0_p = f0_contour(twp[i][idx], 180.0 + 15 * (i % 8))
        # frame j of the professional clip sits at warped time tw -> amateur frame index
        align = np.clip(np.rint(twp[i][idx] * sr / hop), 0, T - 1).astype(np.int64)
        emb = rng.randn(5, 256)
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
        items.append({"item_name": f"synth_{seed0 + i:05d}", "txt": "synthetic", "mel": mel_a[i], "prof_mel": mel_p[i],
                      "f0": f0_a, "prof_f0": f0_p, "pitch": pitch_utils.f0_to_coarse(f0_a),
                      "prof_pitch": pitch_utils.f0_to_coarse(f0_p), "a2p_f0_alignment": align,
                      "multi_spk_emb": emb.astype(np.float32), "len": T, "sec": secs[i]})
    return items


def write_binary_dataset(data_dir, hp, mel_fn, n_train=16, n_valid=2, seconds=6.0, n_phones=60):
    os.makedirs(data_dir, exist_ok=True)
    with open(os.path.join(data_dir, "phone_set.json"), "w") as f:
        json.dump([f"ph{i}" for i in range(n_phones)], f)
    f0s = []
    for prefix, n, seed0 in (("train", n_train, 0), ("valid", n_valid, 1000), ("test", n_valid, 2000)):
        items = build_items(n, seconds, hp, mel_fn, seed0)
        b = IndexedDatasetBuilder(os.path.join(data_dir, prefix))
        for it in items:
            b.add_item(it)
            if prefix == "train":
                f0s.append(it["f0"][it["f0"] > 0])
        b.finalize()
        np.save(os.path.join(data_dir, f"{prefix}_lengths.npy"),
                np.array([max(len(it["mel"]), len(it["prof_mel"])) for it in items]))
    f0s = np.concatenate(f0s)
    np.save(os.path.join(data_dir, "train_f0s_mean_std.npy"), np.array([f0s.mean(), f0s.std()]))


def write_fake_asr_ckpt(ckpt_dir, dict_size, hp):
    """Random-init frozen PPG extractor in the reference's checkpoint layout (loader: utils/ckpt_utils.py:28-43)."""
    from ..modules.vc_asr import VCASR
    os.makedirs(ckpt_dir, exist_ok=True)
    g = torch.Generator().manual_seed(1234)
    m = VCASR(dict_size, hp["audio_num_mel_bins"], hp)
    sd = m.state_dict()
    for k, v in sd.items():
        if v.is_floating_point() and v.dim() >= 2 and "asr_decoder" in k:
            v.copy_(torch.randn(v.shape, generator=g) * 0.02)
    torch.save({"state_dict": {"model": sd}}, os.path.join(ckpt_dir, "model_ckpt_steps_1.ckpt"))


def write_vocoder_dataset(data_dir, hp, mel_fn, n_train=8, n_valid=2, seconds=1.0):
    """Synthetic clips for the vocoder task (tasks/vocoder_dataset.py item schema): wav, its front-end mel, analytic f0."""
    os.makedirs(data_dir, exist_ok=True)
    sr, hop = hp["audio_sample_rate"], hp["hop_size"]
    for prefix, n, seed0 in (("train", n_train, 0), ("valid", n_valid, 1000), ("test", n_valid, 2000)):
        b = IndexedDatasetBuilder(os.path.join(data_dir, prefix))
        lens = []
        for i in range(n):
            wav, tw = make_clip(seconds, sr, seed0 + i, base=170.0 + 20 * (i % 6))
            mel = mel_fn(wav[None])[0]
            T = mel.shape[0]
            f0 = f0_contour((np.arange(T) * hop) / sr, 170.0 + 20 * (i % 6))
            pad = np.zeros(max(0, T * hop - len(wav)), np.float32)
            b.add_item({"item_name": f"voc_{seed0 + i:05d}", "mel": mel, "wav": np.concatenate([wav, pad])[:T * hop],
                        "f0": f0.astype(np.float32), "len": T})
            lens.append(T)
        b.finalize()
        np.save(os.path.join(data_dir, f"{prefix}_lengths.npy"), np.array(lens))

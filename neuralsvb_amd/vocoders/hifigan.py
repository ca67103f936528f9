"""NSF-HifiGAN vocoder plugin on the MI355X.

Drop-in for reference vocoders/hifigan.py (load_model :17-33, HifiGAN :40-69) and the `wav2spec` it inherits from
vocoders/pwg.py:106-122: newest `model_ckpt_steps_*.ckpt` under `hparams['vocoder_ckpt']` + its `config.yaml`
(or legacy `generator_v1` + `config.json`), strict load of `state_dict['model_gen']`, weight norm folded, eval mode.
`spec2wav` also accepts a batch ([B,T,80], [B,T]) -- BASELINE config #5; the reference is batch-1 only.
"""
import glob
import json
import os
import re

import numpy as np
import torch
import yaml

from ..modules.frontend import MelFrontend
from ..modules.hifigan import HifiGanGenerator
from ..utils.hparams import hparams
from .base_vocoder import BaseVocoder, register_vocoder


def load_model(config_path, checkpoint_path, device):
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    if config_path.endswith(".yaml"):
        with open(config_path) as f:
            config = yaml.safe_load(f)
        state = ckpt["state_dict"]["model_gen"]
    else:
        with open(config_path) as f:
            config = json.load(f)
        state = ckpt["generator"]
    model = HifiGanGenerator(config)
    model.load_state_dict(state, strict=True)
    model.remove_weight_norm()
    model = model.eval().to(device)
    print(f"| Loaded model parameters from {checkpoint_path}.")
    return model, config, device


@register_vocoder
class HifiGAN(BaseVocoder):
    def __init__(self, device=None):
        base_dir = hparams["vocoder_ckpt"]
        device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        config_path = f"{base_dir}/config.yaml"
        if os.path.exists(config_path):
            ckpt = sorted(glob.glob(f"{base_dir}/model_ckpt_steps_*.ckpt"),
                          key=lambda x: int(re.findall(rf"{base_dir}/model_ckpt_steps_(\d+).ckpt", x)[0]))[-1]
            print("| load HifiGAN: ", ckpt)
        else:
            config_path, ckpt = f"{base_dir}/config.json", f"{base_dir}/generator_v1"
        self.model, self.config, self.device = load_model(config_path, ckpt, device)

    @torch.no_grad()
    def spec2wav(self, mel, **kwargs):
        """mel [T,80] (or [B,T,80]), f0=[T] (or [B,T]) -> np.float32 wav [T*hop] (or [B, T*hop]).
        Optional rand_ini [B,9] / noise [B,T*hop,9]: the NSF source's draws (SineGen's initial phases and additive noise,
        source.py:84-86,125-127) supplied by the caller instead of drawn -- what a parity test injects."""
        mel = torch.as_tensor(np.asarray(mel) if not isinstance(mel, torch.Tensor) else mel, dtype=torch.float32)
        single = mel.dim() == 2
        c = (mel[None] if single else mel).transpose(2, 1).to(self.device)
        f0 = kwargs.get("f0")
        if f0 is not None:
            f0 = torch.as_tensor(np.asarray(f0) if not isinstance(f0, torch.Tensor) else f0, dtype=torch.float32)
            f0 = (f0[None] if single else f0).to(self.device)
            inj = {k: torch.as_tensor(kwargs[k], dtype=torch.float32).to(self.device) for k in ("rand_ini", "noise")
                   if kwargs.get(k) is not None}
            y = self.model(c, f0, **inj)
        else:
            y = self.model(c)
        y = y[:, 0].cpu().numpy()
        return y[0] if single else y

    @staticmethod
    def wav2spec(wav_fn, return_linear=False):
        """PWG.wav2spec (vocoders/pwg.py:106-122) on the HIP front-end.  wav_fn: path to a wav at the configured sample
        rate, or a float array."""
        if isinstance(wav_fn, str):
            from scipy.io import wavfile
            sr, data = wavfile.read(wav_fn)
            assert sr == hparams["audio_sample_rate"], "resampling is not part of the hot path: provide audio at audio_sample_rate"
            wav = data.astype(np.float32) / (32768.0 if data.dtype == np.int16 else 1.0)
        else:
            wav = np.asarray(wav_fn, dtype=np.float32)
        fe = MelFrontend(hparams, torch.device("cuda"))
        w, mel = fe.wav2mel(torch.from_numpy(wav))
        return w.cpu().numpy(), mel.cpu().numpy()

"""V6: the vocoder PLUGIN (`vocoders.hifigan.HifiGAN`: load_model + spec2wav, reference vocoders/hifigan.py:17-69) against the
unmodified reference's own plugin run on a checkpoint directory in the reference layout (tests/golden/make_golden.py:
spec2wav_golden -> tests/golden/spec2wav.npz)."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import procedural
from tests.test_oracle_golden import HIFIGAN_CFG

G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = json.load(open(os.path.join(G, "ref_state_keys.json")))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_spec2wav_matches_reference_plugin(gpu_only, tmp_path, precision):
    """Stated waveform tolerance (|wav| <= 1): 2e-5 abs with conv_precision fp32 (measured 1.3e-6), 5e-5 with bf16x3 (3.9e-6)."""
    from neuralsvb_amd import functional as SF
    from neuralsvb_amd.utils.hparams import hparams
    SF.set_precision(precision)
    from neuralsvb_amd.vocoders.base_vocoder import get_vocoder_cls
    d = np.load(os.path.join(G, "spec2wav.npz"))
    # a checkpoint directory exactly as the reference's trainer leaves it: config.yaml + model_ckpt_steps_<N>.ckpt
    with open(tmp_path / "config.yaml", "w") as f:
        yaml.safe_dump(dict(HIFIGAN_CFG), f)
    torch.save({"state_dict": {"model_gen": procedural.state_dict_for(KEYS["HifiGanGenerator"], prefix="model_gen.")}},
               tmp_path / "model_ckpt_steps_3.ckpt")
    torch.save({"state_dict": {"model_gen": procedural.state_dict_for(KEYS["HifiGanGenerator"], prefix="model_gen.")}},
               tmp_path / "model_ckpt_steps_7.ckpt")                       # (the newest one is the one that must be loaded)
    old = dict(hparams)
    hparams.update(vocoder="vocoders.hifigan.HifiGAN", vocoder_ckpt=str(tmp_path), audio_sample_rate=HIFIGAN_CFG["audio_sample_rate"])
    try:
        voc = get_vocoder_cls(hparams)()                                   # the reference's dotted path resolves to the plugin
        wav = voc.spec2wav(d["mel"], f0=d["f0"], rand_ini=torch.from_numpy(d["rand_ini"]), noise=torch.from_numpy(d["noise"]))
    finally:
        hparams.clear()
        hparams.update(old)
    assert isinstance(wav, np.ndarray) and wav.dtype == np.float32 and wav.shape == d["wav"].shape
    err = np.abs(wav - d["wav"]).max()
    print(f"[{precision}] spec2wav max abs error vs the reference plugin: {err:.3e}")
    assert err < (2e-5 if precision == "fp32" else 5e-5)     # |wav| <= 1; same bounds as the generator golden (NSF phase rounding)

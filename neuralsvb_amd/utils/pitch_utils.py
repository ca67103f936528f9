"""Host-side pitch helpers with the reference's semantics (utils/pitch_utils.py:130-195).  The bulk bin
quantisation has a bit-exact HIP twin (`svb_f0_to_coarse_f64`, neuralsvb_amd.kernels.f0_to_coarse)."""
import numpy as np
import torch

f0_bin = 256
f0_max = 1100.0
f0_min = 50.0
f0_mel_min = 1127 * np.log(1 + f0_min / 700)
f0_mel_max = 1127 * np.log(1 + f0_max / 700)


def f0_to_coarse(f0):
    """ndarray -> rint (half-to-even); torch tensor on the GPU -> HIP kernel; CPU tensor -> (x+0.5).long()."""
    if isinstance(f0, torch.Tensor):
        if f0.is_cuda:
            from .. import kernels
            return kernels.f0_to_coarse(f0.contiguous())
        mel = 1127 * (1 + f0 / 700).log()
        mel = torch.where(mel > 0, (mel - f0_mel_min) * (f0_bin - 2) / (f0_mel_max - f0_mel_min) + 1, mel)
        return (mel.clamp(min=1, max=f0_bin - 1) + 0.5).long()
    mel = 1127 * np.log(1 + np.asarray(f0, dtype=np.float64) / 700)
    mel = np.where(mel > 0, (mel - f0_mel_min) * (f0_bin - 2) / (f0_mel_max - f0_mel_min) + 1, mel)
    return np.rint(np.clip(mel, 1, f0_bin - 1)).astype(np.int64)


def norm_interp_f0(f0, hparams):
    """standardise / log2, zero the unvoiced frames, then linearly interpolate them (numpy path of :160-176)."""
    f0 = np.asarray(f0, dtype=np.float64).copy()
    uv = f0 == 0
    if hparams["pitch_norm"] == "standard":
        f0 = (f0 - hparams["f0_mean"]) / hparams["f0_std"]
    elif hparams["pitch_norm"] == "log":
        f0 = np.log2(f0 + 1e-8)
    if hparams["use_uv"]:
        f0[uv] = 0
    if uv.all():
        f0[uv] = 0
    elif uv.any():
        f0[uv] = np.interp(np.where(uv)[0], np.where(~uv)[0], f0[~uv])
    return f0, uv


def denorm_f0(f0, uv, hparams, pitch_padding=None, min=None, max=None):
    if hparams["pitch_norm"] == "standard":
        f0 = f0 * hparams["f0_std"] + hparams["f0_mean"]
    elif hparams["pitch_norm"] == "log":
        f0 = 2 ** f0
    f0 = f0.clamp(min=0 if min is None else min, max=f0_max if max is None else max)
    if uv is not None and hparams["use_uv"]:
        f0 = torch.where(uv > 0, torch.zeros_like(f0), f0)
    if pitch_padding is not None:
        f0 = torch.where(pitch_padding, torch.zeros_like(f0), f0)
    return f0

"""TEST INFRASTRUCTURE ONLY -- op-level CPU oracle (plain torch fp32 / fp64 on the host).

These are the exact stock torch op sequences the reference executes at the cited lines; the HIP
kernels in neuralsvb_amd/csrc are checked against them in tests/.  Product code never imports this.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def weight_norm(v, g):
    """torch.nn.utils.weight_norm(dim=0): w = g * v / ||v|| over all dims but 0
    (reference modules/fastspeech/fs2_vae.py:42,48,58; modules/hifigan/hifigan.py:33-50)."""
    norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g.view_as(norm) / norm)


def conv1d(x, w, b=None, stride=1, pad=0, dil=1, groups=1):
    return F.conv1d(x, w, b, stride, pad, dil, groups)


def conv_transpose1d(x, w, b=None, stride=1, pad=0, dil=1, groups=1, output_padding=0):
    return F.conv_transpose1d(x, w, b, stride, pad, output_padding, groups, dil)


def lrelu_gate(t, slope):
    return torch.where(t > 0, torch.ones_like(t), torch.full_like(t, slope))


def wn_gate(xin, g_slice):
    """fused_add_tanh_sigmoid_multiply, reference modules/fastspeech/fs2_vae.py:10-16."""
    c = xin.shape[1] // 2
    a = xin + g_slice
    return torch.tanh(a[:, :c]) * torch.sigmoid(a[:, c:])


def layernorm(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, eps)


def sine_source(f0_frames, rand_ini, noise, lin_w, lin_b, upp, sr, sine_amp=0.1, noise_std=0.003):
    """SineGen.forward + SourceModuleHnNSF.forward restated with injected randomness
    (reference modules/parallel_wavegan/models/source.py:44-74,104-137,385-398 and the nearest
    upsample of modules/hifigan/hifigan.py:113,147).  float32 cumsums exactly as the reference.

    f0_frames [B, frames] -> (merged [B, L], sine_waves [B, L, H], uv [B, L])"""
    B, frames = f0_frames.shape
    H = rand_ini.shape[1]
    f0 = f0_frames.repeat_interleave(upp, dim=1)[:, :, None]                 # nearest upsample, [B, L, 1]
    f0_buf = torch.zeros(B, f0.shape[1], H)
    f0_buf[:, :, 0] = f0[:, :, 0]
    for idx in range(H - 1):
        f0_buf[:, :, idx + 1] = f0_buf[:, :, 0] * (idx + 2)                  # source.py:114-117
    rad = (f0_buf / sr) % 1                                                  # :50
    rand_ini = rand_ini.clone()
    rand_ini[:, 0] = 0                                                       # :55 (no phase noise on the fundamental)
    rad[:, 0, :] = rad[:, 0, :] + rand_ini                                   # :53-56
    tmp_over_one = torch.cumsum(rad, 1) % 1                                  # :66
    over_idx = (tmp_over_one[:, 1:, :] - tmp_over_one[:, :-1, :]) < 0        # :67-68
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over_idx * -1.0                                        # :69-70
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi)          # :72-73
    sine_waves = sines * sine_amp                                            # :120
    uv = (f0 > 0).float()                                                    # :38-42
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3                     # :131
    sine_waves = sine_waves * uv + noise_amp * noise                         # :132-136
    merged = torch.tanh(F.linear(sine_waves, lin_w.view(1, H), lin_b.view(1)))  # :394
    return merged[:, :, 0], sine_waves, uv[:, :, 0]


def sine_source_f64(f0_frames, rand_ini, noise, lin_w, lin_b, upp, sr, sine_amp=0.1, noise_std=0.003):
    """Same signal with the phase accumulated in float64 (the mathematically exact phase of the
    reference's fp32 `rad` values): used to bound the *reference's own* fp32 cumsum drift in tests."""
    B, frames = f0_frames.shape
    H = rand_ini.shape[1]
    f0 = f0_frames.repeat_interleave(upp, dim=1)[:, :, None]
    f0_buf = torch.zeros(B, f0.shape[1], H)
    f0_buf[:, :, 0] = f0[:, :, 0]
    for idx in range(H - 1):
        f0_buf[:, :, idx + 1] = f0_buf[:, :, 0] * (idx + 2)
    rad = ((f0_buf / sr) % 1).double()
    rand_ini = rand_ini.clone()
    rand_ini[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + rand_ini.double()
    phase = torch.cumsum(rad, 1) % 1.0
    sines = torch.sin((phase.float()) * 2 * np.pi)
    uv = (f0 > 0).float()
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    sine_waves = sines * sine_amp * uv + noise_amp * noise
    merged = torch.tanh(F.linear(sine_waves, lin_w.view(1, H), lin_b.view(1)))
    return merged[:, :, 0], sine_waves, uv[:, :, 0]

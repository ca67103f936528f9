"""Mel front-end on the MI355X: STFT + mel filterbank + log in one HIP kernel (`svb_stft_mel`).

Replaces, for this path, `data_gen/tts/data_gen_utils.py:93-147 process_utterance` (offline, via
`vocoders/pwg.py:106-122 PWG.wav2spec`) and `modules/hifigan/mel_utils.py:45-79 mel_spectrogram` (in-graph).
The window and the Slaney mel filterbank are built on the host once per configuration (they are the
constants librosa 0.8.0 would produce -- `scipy.signal.get_window('hann', N, fftbins=True)` and
`librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` with htk=False, norm='slaney').
"""
import functools

import numpy as np
import torch

from .. import kernels as K


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    log_part = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log_part, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)


@functools.lru_cache(maxsize=16)
def _mel_filterbank_np(sr, n_fft, n_mels, fmin, fmax):
    fmax = sr / 2.0 if fmax is None or fmax == -1 else float(fmax)
    fmin = 0.0 if fmin == -1 else float(fmin)
    bins = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
    lower = (bins[None, :] - edges[:-2, None]) / (edges[1:-1] - edges[:-2])[:, None]
    upper = (edges[2:, None] - bins[None, :]) / (edges[2:] - edges[1:-1])[:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return w.astype(np.float32)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    return torch.from_numpy(_mel_filterbank_np(int(sr), int(n_fft), int(n_mels), fmin, fmax).copy())


def hann_window(win_length, n_fft=None):
    """periodic hann, centre-padded to n_fft (librosa.util.pad_center)."""
    n_fft = win_length if n_fft is None else n_fft
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
    lp = (n_fft - win_length) // 2
    return torch.from_numpy(np.pad(w, (lp, n_fft - win_length - lp)).astype(np.float32))


class MelFrontend:
    """Holds the device-resident window + filterbank for one audio config and runs the kernel."""

    def __init__(self, hp, device):
        self.hp, self.device = hp, device
        self.n_fft, self.hop, self.n_mels = hp["fft_size"], hp["hop_size"], hp["audio_num_mel_bins"]
        self.window = hann_window(hp["win_size"], self.n_fft).to(device)
        self.basis = mel_filterbank(hp["audio_sample_rate"], self.n_fft, self.n_mels, hp["fmin"], hp["fmax"]).to(device)

    def wav2mel(self, wav, eps=1e-10):
        """Offline front-end (D2).  wav [B, N] or [N] float32 -> (wav padded/cut to T*hop, mel [B, T, n_mels] log10),
        T = 1 + N // hop  (frame rule of librosa.stft(center=True) + utils/audio.py:67-76 trimming)."""
        squeeze = wav.dim() == 1
        wav = wav[None] if squeeze else wav
        wav = wav.to(self.device, torch.float32).contiguous()
        mel = K.stft_mel(wav, self.window, self.basis, self.n_fft, self.hop, 0, eps)
        T = mel.shape[1]
        n = wav.shape[1]
        r_pad = (n // self.hop + 1) * self.hop - n
        wav = torch.nn.functional.pad(wav, (0, r_pad))[:, :T * self.hop]
        return (wav[0], mel[0]) if squeeze else (wav, mel)

    def mel_spectrogram(self, y):
        """In-graph variant (D4): y [B, N] -> [B, n_mels, N // hop] natural-log mel.

        Without grad: the fused STFT+mel+log kernel.  With grad (vocoder mel loss): the same arithmetic as two GEMMs on the
        implicit-GEMM conv kernel -- frames [B, n_fft, F] x windowed DFT basis [2*(n_fft/2+1), n_fft], then the mel basis
        -- so the backward also runs on the HIP kernels (reference modules/hifigan/mel_utils.py:59-76)."""
        y = y.to(self.device, torch.float32)
        if not (torch.is_grad_enabled() and y.requires_grad):
            return K.stft_mel(y.contiguous(), self.window, self.basis, self.n_fft, self.hop, 1, 1e-5)
        from .. import functional as SF
        if not hasattr(self, "_dft"):
            n = torch.arange(self.n_fft, dtype=torch.float64)
            k = torch.arange(self.n_fft // 2 + 1, dtype=torch.float64)[:, None]
            ang = 2.0 * np.pi * k * n[None, :] / self.n_fft
            w = self.window.double().cpu()[None, :]
            self._dft = torch.cat([torch.cos(ang) * w, -torch.sin(ang) * w], 0).float().to(self.device)[:, :, None].contiguous()
        p = (self.n_fft - self.hop) // 2
        yp = torch.nn.functional.pad(y.clamp(min=-1.0, max=1.0)[:, None], (p, p), mode="reflect")[:, 0]
        frames = yp.unfold(-1, self.n_fft, self.hop).transpose(1, 2).contiguous()       # [B, n_fft, F]
        spec = SF.conv1d(frames, self._dft)                                              # [B, 2*(n_fft/2+1), F]
        nb = self.n_fft // 2 + 1
        mag = torch.sqrt(spec[:, :nb] ** 2 + spec[:, nb:] ** 2 + 1e-9)
        mel = SF.conv1d(mag.contiguous(), self.basis[:, :, None].contiguous())
        return torch.log(torch.clamp(mel, min=1e-5))

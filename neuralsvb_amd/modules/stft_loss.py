"""Multi-resolution STFT loss (reference modules/parallel_wavegan/losses/stft_loss.py:12-31 `stft`, :34-76 spectral
convergence + log-magnitude, :79-152 STFTLoss / MultiResolutionSTFTLoss): optional term of the vocoder generator pass
(`use_ms_stft`, egs/egs_bases/tts/vocoder/hifigan.yaml:15).

The reference calls `torch.stft` without `return_complex` (fails on torch >= 2); the same magnitudes are taken from the
complex result here.  The three resolutions (FFT 1024/2048/512, hop 120/240/50, window 600/1200/240) are not the mel
front-end's geometry, so this optional loss stays on torch's FFT (plumbing, not a hot kernel).
"""
import torch
import torch.nn.functional as F
from torch import nn


def stft_magnitude(x, fft_size, hop_size, win_length, window):
    """x [B, T] -> [B, frames, fft_size//2+1]: sqrt(clamp(re^2 + im^2, 1e-7))  (stft_loss.py:12-31)."""
    s = torch.stft(x, fft_size, hop_size, win_length, window, return_complex=True)
    return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=1e-7)).transpose(2, 1)


class MultiResolutionSTFTLoss(nn.Module):
    def __init__(self, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240)):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.cfg = list(zip(fft_sizes, hop_sizes, win_lengths))
        for i, (_, _, wl) in enumerate(self.cfg):
            self.register_buffer(f"window_{i}", torch.hann_window(wl), persistent=False)

    def forward(self, x, y):
        """x predicted, y ground truth, both [B, T] -> (spectral convergence, log-magnitude), averaged over resolutions."""
        sc = mag = 0.0
        for i, (fs, hs, wl) in enumerate(self.cfg):
            w = getattr(self, f"window_{i}")
            xm, ym = stft_magnitude(x, fs, hs, wl, w), stft_magnitude(y, fs, hs, wl, w)
            sc = sc + torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
            mag = mag + F.l1_loss(torch.log(ym), torch.log(xm))
        n = len(self.cfg)
        return sc / n, mag / n

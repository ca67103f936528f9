"""NSF-HifiGAN: generator (with the NSF harmonic source), multi-period and multi-scale discriminators, GAN losses.

Drop-in for reference modules/hifigan/hifigan.py (ResBlock1 :30-67, HifiGanGenerator :104-178, DiscriminatorP
:181-223, MultiPeriodDiscriminator :226-250, DiscriminatorS :253-286, MultiScaleDiscriminator :289-325, losses
:328-365) and modules/parallel_wavegan/models/source.py (SineGen :7-137, SourceModuleHnNSF :351-398): same config
dict `h`, call signatures, returned structures and state_dict keys (`weight_g/weight_v`, 4-D MPD weights, the
spectral-norm buffers `weight_orig/weight_u/weight_v` of MSD scale 0, `m_source.l_linear.*`, un-normalised
`noise_convs.*`).

MI355X mapping: every Conv1d / ConvTranspose1d runs on the HIP implicit-GEMM kernel with the surrounding
LeakyReLU fused on the operand load, the residual adds fused in the epilogue, WeightNorm fused in the weight pack;
the MPD's (k,1) Conv2d stacks run as 1-D convs over a period-major layout [B*p, C, T/p] (so the conv axis is
contiguous); the NSF source is one HIP kernel (phase in fp64 per frame).  tanh / AvgPool1d / spectral-norm power
iteration / the sum over the three resblocks stay on torch-ROCm ops.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as SF
from .. import kernels as K
from .layers import Conv1d, ConvTranspose1d

LRELU_SLOPE = 0.1
FUSED_RESBLOCK = True       # hparam `fused_resblock` (HifiGanTask): False = one autograd node per conv (the A/B form)


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class ResBlock1(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([Conv1d(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d),
                                            weight_norm=True, init_std=0.01) for d in dilation])
        self.convs2 = nn.ModuleList([Conv1d(channels, channels, kernel_size, dilation=1, padding=get_padding(kernel_size, 1),
                                            weight_norm=True, init_std=0.01) for _ in dilation])

    def forward(self, x):
        if FUSED_RESBLOCK and torch.is_grad_enabled() and x.requires_grad and all(c.precision is None for c in
                                                                                   list(self.convs1) + list(self.convs2)):
            return SF.resblock1(x, self.convs1, self.convs2, LRELU_SLOPE)      # one autograd node, skips summed in conv epilogues
        for c1, c2 in zip(self.convs1, self.convs2):
            xt = c1(x, in_slope=LRELU_SLOPE)                     # c1(leaky_relu(x))
            x = c2(xt, in_slope=LRELU_SLOPE, residual=x)         # c2(leaky_relu(xt)) + x
        return x

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            l.remove_weight_norm()


class ResBlock2(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.convs = nn.ModuleList([Conv1d(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d),
                                           weight_norm=True, init_std=0.01) for d in dilation])

    def forward(self, x):
        for c in self.convs:
            x = c(x, in_slope=LRELU_SLOPE, residual=x)
        return x

    def remove_weight_norm(self):
        for l in self.convs:
            l.remove_weight_norm()


class SourceModuleHnNSF(nn.Module):
    """f0 (per frame, Hz, 0 = unvoiced) -> harmonic excitation [B, 1, frames*upp]  (source.py:104-137,385-398)."""

    def __init__(self, sampling_rate, harmonic_num=0, sine_amp=0.1, add_noise_std=0.003, voiced_threshod=0):
        super().__init__()
        self.sampling_rate, self.dim = float(sampling_rate), harmonic_num + 1
        self.sine_amp, self.noise_std = sine_amp, add_noise_std
        self.l_linear = nn.Linear(harmonic_num + 1, 1)

    def forward(self, f0_frames, upp, rand_ini=None, noise=None):
        B, frames = f0_frames.shape
        dev = f0_frames.device
        if rand_ini is None:
            rand_ini = torch.rand(B, self.dim, device=dev)      # source.py:53-54
        rand_ini = rand_ini.clone()
        rand_ini[:, 0] = 0                                       # :55
        if noise is None:
            noise = torch.randn(B, frames * upp, self.dim, device=dev)   # :132
        f0c = f0_frames.detach().float().contiguous()
        train_lin = torch.is_grad_enabled() and self.l_linear.weight.requires_grad
        if train_lin:   # keep the 9-harmonic tensor so autograd reaches l_linear (vocoder training)
            _, sw, _ = K.nsf_source(f0c, rand_ini, noise.contiguous(), self.l_linear.weight.detach().view(-1).contiguous(),
                                    self.l_linear.bias.detach(), upp, self.sampling_rate, self.sine_amp, self.noise_std,
                                    want_sine_waves=True)
            return torch.tanh(self.l_linear(sw)).transpose(1, 2)
        merged, _, _ = K.nsf_source(f0c, rand_ini, noise.contiguous(), self.l_linear.weight.detach().view(-1).contiguous(),
                                    self.l_linear.bias.detach(), upp, self.sampling_rate, self.sine_amp, self.noise_std)
        return merged[:, None, :]


class HifiGanGenerator(nn.Module):
    def __init__(self, h, c_out=1):
        super().__init__()
        self.h = h
        self.num_kernels = len(h["resblock_kernel_sizes"])
        self.num_upsamples = len(h["upsample_rates"])
        self.use_pitch_embed = bool(h["use_pitch_embed"])
        self.upp = int(np.prod(h["upsample_rates"]))
        if self.use_pitch_embed:
            self.harmonic_num = 8
            self.m_source = SourceModuleHnNSF(sampling_rate=h["audio_sample_rate"], harmonic_num=self.harmonic_num)
            self.noise_convs = nn.ModuleList()
        c0 = h["upsample_initial_channel"]
        self.conv_pre = Conv1d(80, c0, 7, 1, padding=3, weight_norm=True)
        resblock = ResBlock1 if h["resblock"] == "1" else ResBlock2
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            c_cur = c0 // (2 ** (i + 1))
            self.ups.append(ConvTranspose1d(c_cur * 2, c_cur, k, u, padding=(k - u) // 2, weight_norm=True, init_std=0.01))
            if self.use_pitch_embed:
                if i + 1 < len(h["upsample_rates"]):
                    s = int(np.prod(h["upsample_rates"][i + 1:]))
                    self.noise_convs.append(Conv1d(1, c_cur, s * 2, stride=s, padding=s // 2))
                else:
                    self.noise_convs.append(Conv1d(1, c_cur, 1))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
                self.resblocks.append(resblock(h, ch, k, d))
        self.conv_post = Conv1d(ch, c_out, 7, 1, padding=3, weight_norm=True, init_std=0.01)

    def forward(self, x, f0=None, rand_ini=None, noise=None):
        """x [B,80,T] mel, f0 [B,T] Hz -> wav [B,1,T*hop]  (hifigan.py:144-169); rand_ini/noise inject the source's draws."""
        har = self.m_source(f0, self.upp, rand_ini, noise) if f0 is not None else None
        x = self.conv_pre(x.contiguous())
        for i in range(self.num_upsamples):
            x = self.ups[i](x, in_slope=LRELU_SLOPE)                       # ups(leaky_relu(x))
            if har is not None:
                x = self.noise_convs[i](har, residual=x)                   # x + noise_conv(har_source)
            rs = [self.resblocks[i * self.num_kernels + j](x) for j in range(self.num_kernels)]
            x = SF.mean_of(rs) if len(rs) > 1 else rs[0]                  # (xs + ...) / num_kernels, one pass
        x = self.conv_post(x, in_slope=0.01)                               # F.leaky_relu default slope (:165)
        return torch.tanh(x)

    def remove_weight_norm(self):
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()


class _ConvK1(nn.Module):
    """weight_norm(Conv2d(cin, cout, (k,1), (s,1), padding=(p,0))) parameters (4-D, reference layout); computed on the
    reference's own [B, C, H, period] planes as a dilation-`period` 1-D conv over the flattened [H*period] axis -- strided
    layers through a row space-to-depth (functional.period_strided_conv)."""

    def __init__(self, cin, cout, k, stride, padding):
        super().__init__()
        ref = nn.Conv2d(cin, cout, (k, 1), (stride, 1), padding=(padding, 0))
        self.stride, self.padding = stride, padding
        self.weight_g = nn.Parameter(ref.weight.data.flatten(1).norm(dim=1).view(-1, 1, 1, 1))
        self.weight_v = nn.Parameter(ref.weight.data.clone())
        self.bias = ref.bias

    def forward(self, x, H, period, out_act=SF.ACT_NONE):
        """x [B, Cin, H*period] -> ([B, Cout, H_out*period], H_out)."""
        return SF.period_strided_conv(x, H, period, self.weight_v, self.weight_g, self.bias, self.stride, self.padding,
                                      out_act=out_act, out_slope=LRELU_SLOPE)


class DiscriminatorP(nn.Module):
    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False, use_cond=False, c_in=1):
        super().__init__()
        if use_cond or use_spectral_norm:
            raise NotImplementedError("hifigan.yaml: use_cond false, MPD is weight-normalised")
        self.period = period
        chans = [c_in, 32, 128, 512, 1024]
        self.convs = nn.ModuleList([_ConvK1(chans[i], chans[i + 1], kernel_size, stride, get_padding(5, 1)) for i in range(4)]
                                   + [_ConvK1(1024, 1024, kernel_size, 1, 2)])
        self.conv_post = _ConvK1(1024, 1, 3, 1, 1)

    def forward(self, x, mel=None):
        """x [B,1,T] -> (flattened logits [B, H'*p], fmaps [B,C,H,p])  (hifigan.py:202-223).  Every feature map lives in the
        reference's layout: no period-major copy, and a clip is H*p positions long for the conv tiles instead of H."""
        b, c, t = x.shape
        p = self.period
        if t % p != 0:
            x = F.pad(x, (0, p - (t % p)), "reflect")
            t = x.shape[-1]
        h, H = x, t // p
        fmap = []
        for l in self.convs:
            h, H = l(h, H, p, out_act=SF.ACT_LRELU)
            fmap.append(h.view(b, h.shape[1], H, p))
        h, H = self.conv_post(h, H, p)
        fmap.append(h.view(b, 1, H, p))
        return h.reshape(b, -1), fmap


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self, use_cond=False, c_in=1):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorP(p, use_cond=use_cond, c_in=c_in) for p in (2, 3, 5, 7, 11)])

    def forward(self, y, y_hat, mel=None):
        """Real and generated audio go through each period discriminator as ONE stacked batch (every layer is a
        weight-normalised conv + LeakyReLU, i.e. per clip), then the outputs are split: identical results, half the
        launches, twice the work per launch."""
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        B = y.shape[0]
        both = torch.cat([y, y_hat], 0)
        mel2 = None if mel is None else torch.cat([mel, mel], 0)
        for d in self.discriminators:
            o, fm = d(both, mel2)
            r, g = SF.split_batch_halves(o)
            y_d_rs.append(r); y_d_gs.append(g)
            fmap_rs.append([f[:B] for f in fm]); fmap_gs.append([f[B:] for f in fm])
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs


FUSED_SPECTRAL_NORM = True      # hparam `fused_spectral_norm` (HifiGanTask): False = torch.nn.utils.spectral_norm's op-by-op formulation


class _SpectralConv1d(nn.Module):
    """spectral_norm(Conv1d) state: weight_orig (param), weight_u / weight_v (buffers), bias  (SURVEY Appendix A.12)."""

    def __init__(self, cin, cout, k, stride, padding, groups=1):
        super().__init__()
        ref = nn.Conv1d(cin, cout, k, stride, padding=padding, groups=groups)
        self.stride, self.padding, self.groups = stride, padding, groups
        self.weight_orig = nn.Parameter(ref.weight.data.clone())
        self.bias = ref.bias
        wm = self.weight_orig.data.flatten(1)
        u = F.normalize(torch.randn(wm.shape[0]), dim=0, eps=1e-12)
        v = F.normalize(torch.randn(wm.shape[1]), dim=0, eps=1e-12)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)

    def _weight(self):
        """weight_orig / sigma after one power iteration (training) -- one fused op (SF.spectral_norm_weight: 3 launches where
        torch.nn.utils.spectral_norm's formulation issues 15, and 2 instead of ~12 in backward)."""
        if not FUSED_SPECTRAL_NORM:
            return self._weight_torch()
        return SF.spectral_norm_weight(self.weight_orig, self.weight_u, self.weight_v, self.training)

    def _weight_torch(self):
        """The same in torch ops, as torch.nn.utils.spectral_norm computes it (kept for the parity test)."""
        wm = self.weight_orig.flatten(1)
        if self.training:   # one power iteration per forward, buffers updated in place (torch.nn.utils.spectral_norm)
            with torch.no_grad():
                v = F.normalize(torch.mv(wm.t(), self.weight_u), dim=0, eps=1e-12)
                u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
                self.weight_v.copy_(v)
                self.weight_u.copy_(u)
        u, v = self.weight_u.clone(), self.weight_v.clone()
        sigma = torch.dot(u, torch.mv(wm, v))
        return self.weight_orig / sigma

    def forward(self, x, out_act=SF.ACT_NONE):
        return SF.conv1d(x, self._weight(), self.bias, self.stride, self.padding, 1, self.groups, out_act=out_act,
                         out_slope=LRELU_SLOPE)


class _WNConv1d(Conv1d):
    def __init__(self, cin, cout, k, stride, padding, groups=1):
        super().__init__(cin, cout, k, stride, padding, 1, groups, weight_norm=True)

    def forward(self, x, out_act=SF.ACT_NONE):
        return super().forward(x, out_act=out_act, out_slope=LRELU_SLOPE)


class DiscriminatorS(nn.Module):
    def __init__(self, use_spectral_norm=False, use_cond=False, upsample_rates=None, c_in=1):
        super().__init__()
        if use_cond:
            raise NotImplementedError("hifigan.yaml: use_cond false")
        mk = _SpectralConv1d if use_spectral_norm else _WNConv1d
        cfg = [(c_in, 128, 15, 1, 7, 1), (128, 128, 41, 2, 20, 4), (128, 256, 41, 2, 20, 16), (256, 512, 41, 4, 20, 16),
               (512, 1024, 41, 4, 20, 16), (1024, 1024, 41, 1, 20, 16), (1024, 1024, 5, 1, 2, 1)]
        self.convs = nn.ModuleList([mk(*c) for c in cfg])
        self.conv_post = mk(1024, 1, 3, 1, 1)

    def forward(self, x, mel=None):
        fmap = []
        for l in self.convs:
            x = l(x, out_act=SF.ACT_LRELU)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class MultiScaleDiscriminator(nn.Module):
    def __init__(self, use_cond=False, c_in=1):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorS(use_spectral_norm=True, c_in=c_in), DiscriminatorS(c_in=c_in),
                                             DiscriminatorS(c_in=c_in)])
        self.meanpools = nn.ModuleList([nn.AvgPool1d(4, 2, padding=1), nn.AvgPool1d(4, 2, padding=1)])

    def forward(self, y, y_hat, mel=None):
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        B = y.shape[0]
        for i, d in enumerate(self.discriminators):
            if i != 0:
                y, y_hat = self.meanpools[i - 1](y), self.meanpools[i - 1](y_hat)
            if i == 0:
                # spectral norm runs one power iteration per forward call (buffers updated in place): the reference's
                # real and generated calls see different sigma estimates, so this scale keeps its two separate calls
                r, fr = d(y.contiguous(), mel)
                g, fg = d(y_hat.contiguous(), mel)
            else:   # weight-normalised scales: real and generated stacked into one batch (per-clip layers only)
                o, fm = d(torch.cat([y, y_hat], 0), None if mel is None else torch.cat([mel, mel], 0))
                r, g = SF.split_batch_halves(o)
                fr, fg = [f[:B] for f in fm], [f[B:] for f in fm]
            y_d_rs.append(r); fmap_rs.append(fr); y_d_gs.append(g); fmap_gs.append(fg)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs


def feature_loss(fmap_r, fmap_g):
    """2 * sum over the feature-map pairs of mean(|r - g|)  (hifigan.py:328-335)."""
    pairs = [(rl, gl) for dr, dg in zip(fmap_r, fmap_g) for rl, gl in zip(dr, dg)]
    if SF.FUSED_FEATURE_LOSS and pairs and all(r.dtype == torch.float32 and r.shape == g.shape for r, g in pairs):
        # all pairs in one multi-tensor launch per direction (~430 stock launches per generator pass otherwise)
        return SF.l1_mean_pairs([r for r, _ in pairs], [g for _, g in pairs], 2.0)
    loss = 0
    for rl, gl in pairs:
        loss = loss + torch.mean(torch.abs(rl - gl))
    return loss * 2


def _fused_gan_terms(outs):
    return SF.FUSED_GAN_LOSS and outs and all(o.dtype == torch.float32 and o.numel() > 0 for o in outs)


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    """(mean over discriminators of mean((1 - D(x))^2), ... of mean(D(G(s))^2))  (hifigan.py:338-353)."""
    n = len(disc_real_outputs)
    if _fused_gan_terms(list(disc_real_outputs) + list(disc_generated_outputs)):
        # one multi-tensor launch per term list and direction (4 stock launches per discriminator and term otherwise)
        return (SF.mean_sq_to_targets(list(disc_real_outputs), [1.0] * n, 1.0 / n),
                SF.mean_sq_to_targets(list(disc_generated_outputs), [0.0] * len(disc_generated_outputs), 1.0 / n))
    r = sum(torch.mean((1 - dr) ** 2) for dr in disc_real_outputs) / n
    g = sum(torch.mean(dg ** 2) for dg in disc_generated_outputs) / n
    return r, g


def generator_loss(disc_outputs):
    """mean over discriminators of mean((1 - D(G(s)))^2)  (hifigan.py:356-365)."""
    if _fused_gan_terms(list(disc_outputs)):
        return SF.mean_sq_to_targets(list(disc_outputs), [1.0] * len(disc_outputs), 1.0 / len(disc_outputs))
    return sum(torch.mean((1 - dg) ** 2) for dg in disc_outputs) / len(disc_outputs)

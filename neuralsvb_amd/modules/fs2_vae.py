"""Gated conv VAE with a *global* latent: WN, GlobalFVAEEncoder/Decoder, GlobalFVAE, GlobalLatentMap.

Host-side mirror of reference modules/fastspeech/fs2_vae.py (WN :19-100, FVAEEncoder :103-127, FVAEDecoder :130-151,
FVAE :154-206) and modules/voice_conversion/vae_models.py (TMPFVAE :12-48, GlobalFVAEEncoder :81-105,
GlobalFVAEDecoder :107-131, GlobalFVAE :133-147, GlobalLatentMap :149-172): same constructor arguments, attribute
names and state_dict keys; the convolutions, WeightNorm, gate and res/skip updates run on the HIP kernels.
Masks are carried as [B, T] (the kernels broadcast over channels).
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as SF
from .layers import Conv1d, ConvTranspose1d


FUSED_HEAD = True        # GlobalFVAEEncoder: latent head (mean, split, draw, guard, KL) as one HIP pass per direction


class WN(nn.Module):
    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1 and hidden_channels % 2 == 0
        if p_dropout != 0:
            raise NotImplementedError("the hot path uses p_dropout=0 (fs2_vae.py:115,143)")
        self.hidden_channels, self.kernel_size, self.dilation_rate = hidden_channels, kernel_size, dilation_rate
        self.n_layers, self.gin_channels = n_layers, gin_channels
        self.precision = None
        self.in_layers, self.res_skip_layers = nn.ModuleList(), nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = Conv1d(gin_channels, 2 * hidden_channels * n_layers, 1, weight_norm=True)
        for i in range(n_layers):
            d = dilation_rate ** i
            self.in_layers.append(Conv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=d,
                                         padding=(kernel_size * d - d) // 2, weight_norm=True))
            rc = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(Conv1d(hidden_channels, rc, 1, weight_norm=True))

    @staticmethod
    def _p(c):
        return (c.weight_v, c.weight_g, c.bias) if c.is_weight_norm else (c.weight, None, c.bias)

    def forward(self, x, x_mask=None, g=None):
        """x [B,C,T]; x_mask [B,T] (or None); g [B,gin,T].  Returns output * mask  (fs2_vae.py:61-91)."""
        layers = [self._p(a) + self._p(b) for a, b in zip(self.in_layers, self.res_skip_layers)]
        cond = self._p(self.cond_layer) if (g is not None and self.gin_channels != 0) else None
        with SF.precision_scope(self.precision):
            return SF.wn_stack(x, x_mask, g if cond is not None else None, cond, layers, self.kernel_size,
                               self.dilation_rate)

    def remove_weight_norm(self):
        for m in list(self.in_layers) + list(self.res_skip_layers) + ([self.cond_layer] if self.gin_channels else []):
            m.remove_weight_norm()


class GlobalFVAEEncoder(nn.Module):
    def __init__(self, in_channels, hidden_channels, latent_channels, kernel_size, n_layers, gin_channels=0, p_dropout=0,
                 strides=(4,)):
        super().__init__()
        self.strides, self.hidden_size, self.latent_channels = list(strides), hidden_channels, latent_channels
        self.pre_net = nn.Sequential(*[
            Conv1d(in_channels if i == 0 else hidden_channels, hidden_channels, s * 2, stride=s, padding=s // 2)
            for i, s in enumerate(self.strides)])
        self.wn = WN(hidden_channels, kernel_size, 1, n_layers, gin_channels, p_dropout)
        self.out_proj = Conv1d(hidden_channels, latent_channels * 2, 1)
        c2 = latent_channels * 2
        self._head_kl = None
        self.poolings = nn.Sequential(
            Conv1d(c2, c2, 3, stride=2), nn.ReLU(), nn.BatchNorm1d(c2),
            Conv1d(c2, c2, 3, stride=2), nn.ReLU(), nn.BatchNorm1d(c2),
            Conv1d(c2, c2, 3, stride=2))

    def forward(self, x, x_mask, g, eps=None, groups=1):
        """x [B,80,T], x_mask [B,T], g [B,gin,T/4]; eps: optional injected N(0,1) draw (vae_models.py:96-105).
        groups > 1: the batch is `groups` independent calls stacked along dim 0 (the a2a and p2p ways run as one wide
        launch sequence); the train-mode BatchNorms then normalise -- and update their running statistics -- per group,
        in order, exactly as the separate calls would."""
        n = len(self.pre_net)
        m = None
        for i, (conv, s) in enumerate(zip(self.pre_net, self.strides)):
            if i == n - 1:   # `x * x_mask` (vae_models.py:98-99) fused into the last pre-net conv's epilogue
                m = x_mask[:, ::int(np.prod(self.strides))][:, :conv_len(x.shape[-1], s)].contiguous()
                x = conv(x, mask=m)
            else:
                x = conv(x)
        x = self.wn(x, m, g)                      # output already masked; the reference's extra `* x_mask` is idempotent
        x = self.out_proj(x)
        p = self.poolings
        x = bn_groups(p[2], p[0](x, out_act=SF.ACT_RELU), groups)
        x = bn_groups(p[5], p[3](x, out_act=SF.ACT_RELU), groups)
        x = p[6](x)
        self._head_kl = None
        if FUSED_HEAD:
            # time mean, (m_q, logs_q) split, re-parameterised draw, positivity guard and the masked KL mean in one pass
            # (logs_q comes back already guarded; GlobalFVAE.forward picks the KL up from _head_kl)
            if eps is None:
                eps = torch.randn((x.shape[0], self.latent_channels, 1), device=x.device, dtype=x.dtype)
            z, m_q, logs_q, self._head_kl = SF.vae_head(x, eps, m, groups)
            return z, m_q, logs_q, m[:, None, :]
        x = x.mean(-1, keepdim=True)
        m_q, logs_q = torch.split(x, self.latent_channels, dim=1)
        if eps is None:
            eps = torch.randn_like(m_q)
        z = m_q + eps * torch.exp(logs_q)
        return z, m_q, logs_q, m[:, None, :]


def bn_groups(bn, x, groups):
    """BatchNorm over each of `groups` equal batch slices separately (see GlobalFVAEEncoder.forward)."""
    return SF.batch_norm_nct(bn, x, groups)


def conv_len(t, s):
    return (t + 2 * (s // 2) - (2 * s - 1) - 1) // s + 1


class GlobalFVAEDecoder(nn.Module):
    def __init__(self, latent_channels, hidden_channels, out_channels, kernel_size, n_layers, gin_channels=0, p_dropout=0,
                 strides=(4,)):
        super().__init__()
        self.strides, self.hidden_size = list(strides), hidden_channels
        self.pre_net = nn.Sequential(*[
            ConvTranspose1d(latent_channels if i == 0 else hidden_channels, hidden_channels, s, stride=s)
            for i, s in enumerate(self.strides)])
        self.wn = WN(hidden_channels, kernel_size, 1, n_layers, gin_channels, p_dropout)
        self.out_proj = Conv1d(hidden_channels, out_channels, 1)

    def forward(self, z, x_mask, g):
        """z [B,L,1]; x_mask [B,T]; g [B,gin,T]  (vae_models.py:124-131)."""
        x = z.repeat(1, 1, g.shape[-1] // 4)
        n = len(self.pre_net)
        for i, ct in enumerate(self.pre_net):
            x = ct(x, mask=x_mask if i == n - 1 else None)
        x = self.wn(x, x_mask, g)
        return self.out_proj(x)


class GlobalFVAE(nn.Module):
    def __init__(self, in_out_channels, hidden_channels, latent_size, kernel_size, enc_n_layers, dec_n_layers,
                 gin_channels, strides, use_prior_glow=False, **_):
        super().__init__()
        if use_prior_glow:
            raise NotImplementedError("vae_global_mle_eng uses use_prior_glow=False (svb_vae.py:187)")
        self.strides, self.hidden_size, self.latent_size = list(strides), hidden_channels, latent_size
        self.g_pre_net = nn.Sequential(*[Conv1d(gin_channels, gin_channels, s * 2, stride=s, padding=s // 2)
                                         for s in self.strides])
        self.encoder = GlobalFVAEEncoder(in_out_channels, hidden_channels, latent_size, kernel_size, enc_n_layers,
                                         gin_channels, strides=strides)
        self.decoder = GlobalFVAEDecoder(latent_size, hidden_channels, in_out_channels, kernel_size, dec_n_layers,
                                         gin_channels, strides=strides)

    def forward(self, x=None, x_mask=None, g=None, infer=False, eps=None, groups=1):
        """x [B,80,T]; x_mask [B,T]; g [B,gin,T]  ->  (x_recon, kl, z_p=None, m_q, logs_q, x_mask_sqz, z_q)
        (TMPFVAE.forward, vae_models.py:12-41).  groups > 1: `groups` stacked independent calls, kl is [groups]."""
        g_sqz = g
        for c in self.g_pre_net:
            g_sqz = c(g_sqz)
        if infer:
            raise NotImplementedError("run_model always passes infer=False (svb_vae_task.py:148; SURVEY Appendix A.1)")
        z_q, m_q, logs_q, mask_sqz = self.encoder(x, x_mask, g_sqz, eps, groups)
        x_recon = self.decoder(z_q, x_mask, g)
        if self.encoder._head_kl is not None:
            kl, self.encoder._head_kl = self.encoder._head_kl, None
            return x_recon, (kl[0] if groups == 1 else kl), None, m_q, logs_q, mask_sqz, z_q
        with torch.no_grad():  # positivity guard of vae_models.py:24-30 (unconditional select: no device->host sync)
            bad = ~(logs_q.exp() > 0)
        logs_q = torch.where(bad, torch.zeros_like(logs_q), logs_q)
        kl = 0.5 * (torch.exp(2 * logs_q) + m_q ** 2 - 1.0) - logs_q    # KL(N(m, e^logs) || N(0,1))
        if groups == 1:
            loss_kl = (kl * mask_sqz).sum() / mask_sqz.sum() / z_q.shape[1]
        else:
            loss_kl = ((kl * mask_sqz).reshape(groups, -1).sum(1) / mask_sqz.reshape(groups, -1).sum(1)) / z_q.shape[1]
        return x_recon, loss_kl, None, m_q, logs_q, mask_sqz, z_q


class GlobalLatentMap(nn.Module):
    """vae_models.py:149-172 -- 1x1 convs + BatchNorm on [B, 128, 1] (the phase-3 latent map).  The parameter containers are the
    reference's (`convs.{0,3,6}` / `spk_proj.{0,2}` nn.Conv1d, `convs.{1,4}` nn.BatchNorm1d: same state_dict); the arithmetic runs on
    the HIP conv (ReLU and the `x + spk_proj(...)` add in its epilogue) and the HIP BatchNorm kernel -- no stock-torch op on the
    path."""

    def __init__(self, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.convs = nn.Sequential(
            nn.Conv1d(hidden_size, hidden_size, 1), nn.BatchNorm1d(hidden_size), nn.ReLU(inplace=True),
            nn.Conv1d(hidden_size, hidden_size, 1), nn.BatchNorm1d(hidden_size), nn.ReLU(inplace=True),
            nn.Conv1d(hidden_size, hidden_size, 1))
        self.spk_proj = nn.Sequential(nn.Conv1d(256, hidden_size, 1), nn.ReLU(inplace=True),
                                      nn.Conv1d(hidden_size, hidden_size, 1))

    @staticmethod
    def _c(conv, x, **epi):
        return SF.conv1d(x, conv.weight, conv.bias, **epi)

    def forward(self, x, spk_emb):
        c, s = self.convs, self.spk_proj
        h = self._c(s[0], spk_emb[:, :, :x.shape[-1]].contiguous(), out_act=SF.ACT_RELU)
        h = self._c(s[2], h, residual=x.contiguous())                          # x + spk_proj(spk)
        h = SF.batch_norm_nct(c[1], self._c(c[0], h))
        h = SF.batch_norm_nct(c[4], self._c(c[3], h, in_slope=0.0))            # ReLU applied on the operand load of the next conv
        return self._c(c[6], h, in_slope=0.0)

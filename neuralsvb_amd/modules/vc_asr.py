"""Frozen PPG / content encoder (VCASR, encoder side only) in NCT layout.

Host-side mirror of reference modules/voice_conversion/vc_modules.py:56-80 (VCASR), modules/fastspeech/pe.py:7-41
(Prenet), modules/fastspeech/conformer/conformer.py:9-53 (ConformerLayers), conformer/layers.py (EncoderLayer,
ConvolutionModule, MultiLayeredConv1d), modules/commons/espnet_transformer_attn.py:106-186 and
espnet_positional_embedding.py:89-112.  Same state_dict keys (incl. the never-executed asr_decoder / token_embed,
kept as opaque frozen parameters so reference checkpoints load strictly).

Forward-only (the task freezes it, svb_vae_task.py:558-561) and eval-mode (BatchNorm running stats, no dropout).
All dense projections / FFN / k5 convs, the five LayerNorms per block, the rel-pos softmax glue and the conv module's
GLU + depthwise k31 + BatchNorm + Swish chain run on the HIP kernels; the three attention matrix products stay on rocBLAS.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as SF
from .. import kernels as K
from .layers import Conv1d, LinearNCT, LayerNormNCT, attach_opaque


FOLD_RESIDUALS = True    # EncoderLayer: residual adds / macaron 0.5 in the conv epilogues (False: explicit element-wise ops)


class Prenet(nn.Module):
    def __init__(self, in_dim=80, out_dim=256, kernel=5, n_layers=3, strides=None):
        super().__init__()
        self.strides = strides if strides is not None else [1] * n_layers
        layers = []
        for l in range(n_layers):
            layers.append(nn.Sequential(Conv1d(in_dim, out_dim, kernel, stride=self.strides[l], padding=kernel // 2),
                                        nn.ReLU(), nn.BatchNorm1d(out_dim)))
            in_dim = out_dim
        self.layers = nn.ModuleList(layers)
        self.out_proj = LinearNCT(out_dim, out_dim, xavier=False)

    def forward(self, mel_nct, nonpad):
        """mel_nct [B,80,T]; nonpad [B,T] float.  Returns ([B,H,T'], nonpad' [B,T'])  (pe.py:23-41)."""
        x = mel_nct
        for i, l in enumerate(self.layers):
            nonpad = nonpad[:, ::self.strides[i]]
            x = l[0](x, out_act=SF.ACT_RELU)
            x = SF.batch_norm_nct(l[2], x, mask=nonpad)           # bn(x) * nonpadding, one kernel
        nonpad = nonpad.contiguous()
        return self.out_proj(x, mask=nonpad), nonpad


POS_IN_KERNEL = True     # frozen encoder, d_k = 64: position scores computed inside the attention kernel (K.relpos_attention_pos)
FUSE_QKV = True          # frozen encoder: one D -> 4D projection instead of three + an add (see RelPositionMultiHeadedAttention.forward)


class RelPositionMultiHeadedAttention(nn.Module):
    def __init__(self, n_head, n_feat):
        super().__init__()
        self.d_k, self.h = n_feat // n_head, n_head
        self.linear_q, self.linear_k = LinearNCT(n_feat, n_feat, xavier=False), LinearNCT(n_feat, n_feat, xavier=False)
        self.linear_v, self.linear_out = LinearNCT(n_feat, n_feat, xavier=False), LinearNCT(n_feat, n_feat, xavier=False)
        self.linear_pos = LinearNCT(n_feat, n_feat, bias=False, xavier=False)
        self.pos_bias_u = nn.Parameter(torch.empty(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.empty(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)

    def _fused_qkv(self):
        """([4D, D, 1], [4D]) = rows of linear_q, linear_k, linear_v, linear_q again with `+ pos_bias_v` folded into its bias;
        rebuilt when any of the (frozen) tensors changes."""
        ts = (self.linear_q.weight, self.linear_k.weight, self.linear_v.weight, self.linear_q.bias, self.linear_k.bias,
              self.linear_v.bias, self.pos_bias_v)
        key = tuple((t._version, t.data_ptr()) for t in ts)
        c = getattr(self, "_qkv_cache", None)
        if c is None or c[0] != key:
            with torch.no_grad():
                w4 = torch.cat([ts[0], ts[1], ts[2], ts[0]], 0)[:, :, None].contiguous()
                b4 = torch.cat([ts[3], ts[4], ts[5], ts[3] + ts[6].reshape(-1)], 0).contiguous()
            D = ts[0].shape[0]
            c = self._qkv_cache = (key, w4, b4, w4[:3 * D].contiguous(), b4[:3 * D].contiguous())
        return c[1], c[2], c[3], c[4]

    def _pos_proj(self, pos_emb):
        """linear_pos(pos_emb) of the frozen encoder, cached per (length, weights): the position embedding is a constant."""
        w = self.linear_pos.weight
        key = (w._version, w.data_ptr(), pos_emb.data_ptr(), tuple(pos_emb.shape))
        c = getattr(self, "_pos_cache", None)
        if c is None or c[0] != key:
            with torch.no_grad():
                c = self._pos_cache = (key, self.linear_pos(pos_emb), pos_emb)      # (pos_emb kept alive: its address keys the entry)
        return c[1]

    def _pos_table(self, pos_emb):
        """(pt_hi, pt_lo): the position projection in the attention kernel's operand layout (K.relpos_pos_table), cached with it."""
        p = self._pos_proj(pos_emb)
        c = getattr(self, "_pt_cache", None)
        if c is None or c[0] is not p:
            c = self._pt_cache = (p,) + K.relpos_pos_table(p, self.h)
        return c[1], c[2]

    def forward(self, x, pos_emb, mask, residual=None):
        """x [B,D,T]; pos_emb [1,D,T]; mask [B,T] bool (True = keep)  (espnet_transformer_attn.py:150-186).
        residual: added to the result in the output projection's epilogue (the block's `x + self_attn(...)`)."""
        B, D, T = x.shape
        h, dk = self.h, self.d_k
        frozen = not (torch.is_grad_enabled() and (x.requires_grad or self.pos_bias_u.requires_grad
                                                   or self.linear_q.weight.requires_grad))
        if frozen and FUSE_QKV and POS_IN_KERNEL and dk == 64:
            # ... and the position scores inside the attention kernel: no [B,h,T,T] tensor, no GEMM (round 4)
            w4, b4, w3, b3 = self._fused_qkv()
            with SF.precision_scope(self.linear_q.precision):
                y = SF.conv1d(x, w3, b3)                                            # [B, 3D, T]
            pt_hi, pt_lo = self._pos_table(pos_emb)
            o = K.relpos_attention_pos(y[:, :D], y[:, D:2 * D], y[:, 2 * D:], self.pos_bias_u.contiguous(),
                                       self.pos_bias_v.contiguous(), pt_hi, pt_lo, mask.float().contiguous(), 1.0 / math.sqrt(dk), h)
            return self.linear_out(o, residual=residual)
        if frozen and FUSE_QKV:
            # frozen encoder: q, k, v and q + pos_bias_v as ONE 1x1 conv D -> 4D (the three projections read the same x; the
            # fourth block is linear_q again with pos_bias_v folded into its bias), the position projection of the (constant)
            # position embedding cached per length
            w4, b4, _, _ = self._fused_qkv()
            with SF.precision_scope(self.linear_q.precision):
                y = SF.conv1d(x, w4, b4)                                            # [B, 4D, T]
            q, k, v, qv = y[:, :D], y[:, D:2 * D], y[:, 2 * D:3 * D], y[:, 3 * D:]
            p = self._pos_proj(pos_emb).view(1, h, dk, T)
            q4 = q.view(B, h, dk, T)
            q_v = qv.view(B, h, dk, T).transpose(-1, -2)
        else:
            q, k, v = self.linear_q(x), self.linear_k(x), self.linear_v(x)              # [B, D, T]
            p = self.linear_pos(pos_emb).view(1, h, dk, T)
            q4 = q.view(B, h, dk, T)
            q_v = (q4 + self.pos_bias_v[None, :, :, None]).transpose(-1, -2)
        bd = torch.matmul(q_v, p)                                   # [B,h,T,T] position scores, unshifted
        scale = 1.0 / math.sqrt(dk)
        if dk == 64 and frozen:
            # frozen encoder (the hot path): content scores, rel_shift (:125-148), scale, key mask, softmax and the value
            # product in ONE kernel -- neither `ac` nor `attn` ([B,h,T,T] each) is ever written
            o = K.relpos_attention(q, k, v, self.pos_bias_u.contiguous(), bd, mask.float().contiguous(), scale, h)
        else:
            q_u = (q4 + self.pos_bias_u[None, :, :, None]).transpose(-1, -2)
            ac = torch.matmul(q_u, k.view(B, h, dk, T))             # [B,h,T,T]
            # rel_shift + add + scale + key mask + softmax + mask: one HIP kernel, one pass over [B,h,T,T]
            attn = K.relpos_softmax(ac, bd.expand(B, h, T, T).contiguous(), mask.float().contiguous(), scale)
            o = torch.matmul(v.view(B, h, dk, T), attn.transpose(-1, -2)).reshape(B, D, T)
        return self.linear_out(o, residual=residual)


class MultiLayeredConv1d(nn.Module):
    def __init__(self, in_chans, hidden_chans, kernel_size):
        super().__init__()
        self.w_1 = Conv1d(in_chans, hidden_chans, kernel_size, padding=(kernel_size - 1) // 2)
        self.w_2 = Conv1d(hidden_chans, in_chans, kernel_size, padding=(kernel_size - 1) // 2)

        self._half = None

    def _half_w2(self):
        """(0.5 * w_2.weight, 0.5 * w_2.bias) of a frozen layer, cached: halving is exact in binary floating point, so
        conv(h, 0.5 W, 0.5 b) + x == x + 0.5 * conv(h, W, b) bit for bit (up to the order of the final add)."""
        w, b = self.w_2.weight, self.w_2.bias
        key = (w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr()))
        if self._half is None or self._half[0] != key:
            with torch.no_grad():
                self._half = (key, (w * 0.5).contiguous(), None if b is None else (b * 0.5).contiguous())
        return self._half[1], self._half[2]

    def forward(self, x, half_residual=None):
        """half_residual (frozen, no-grad use only): returns half_residual + 0.5 * ffn(x) (the macaron block's update,
        conformer/layers.py:216-222,243-247) with the scale folded into w_2 and the add into its epilogue."""
        h = self.w_1(x, out_act=SF.ACT_RELU)
        if half_residual is None:
            return self.w_2(h)
        c = self.w_2
        if torch.is_grad_enabled() and (c.weight.requires_grad or (c.bias is not None and c.bias.requires_grad)):
            return half_residual + 0.5 * self.w_2(h)      # trainable (ASR pre-training): the folded copy would cut the gradient
        w, b = self._half_w2()
        with SF.precision_scope(c.precision):
            return SF.conv1d(h, w, b, c.stride, c.padding, c.dilation, c.groups, residual=half_residual)


class ConvolutionModule(nn.Module):
    def __init__(self, channels, kernel_size):
        super().__init__()
        self.pointwise_conv1 = Conv1d(channels, 2 * channels, 1)
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, padding=(kernel_size - 1) // 2, groups=channels)
        self.norm = nn.BatchNorm1d(channels)
        self.pointwise_conv2 = Conv1d(channels, channels, 1)

    def forward(self, x, residual=None):
        y = self.pointwise_conv1(x)
        if self.training or self.norm.training:          # (never on the hot path: VCASR pins eval mode)
            x = F.glu(y, dim=1)
            x = self.norm(self.depthwise_conv(x))
            x = x * torch.sigmoid(x)
        else:                                            # GLU + depthwise k31 + folded BatchNorm + Swish: one HIP kernel
            dw, bn = self.depthwise_conv, self.norm
            x = K.glu_dwconv_bn_swish(y.contiguous(), dw.weight, dw.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      bn.eps)
        return self.pointwise_conv2(x, residual=residual)


class EncoderLayer(nn.Module):
    """Macaron conformer block, normalize_before, eval mode (conformer/layers.py:182-258)."""

    def __init__(self, size, num_heads, kernel_size):
        super().__init__()
        self.self_attn = RelPositionMultiHeadedAttention(num_heads, size)
        self.feed_forward = MultiLayeredConv1d(size, size * 4, 1)
        self.feed_forward_macaron = MultiLayeredConv1d(size, size * 4, 1)
        self.conv_module = ConvolutionModule(size, kernel_size)
        self.norm_ff, self.norm_mha = LayerNormNCT(size), LayerNormNCT(size)
        self.norm_ff_macaron = LayerNormNCT(size)
        self.norm_conv, self.norm_final = LayerNormNCT(size), LayerNormNCT(size)

    def forward(self, x, pos_emb, mask):
        if not FOLD_RESIDUALS:
            x = x + 0.5 * self.feed_forward_macaron(self.norm_ff_macaron(x))
            x = x + self.self_attn(self.norm_mha(x), pos_emb, mask)
            x = x + self.conv_module(self.norm_conv(x))
            x = x + 0.5 * self.feed_forward(self.norm_ff(x))
            return self.norm_final(x)
        # (forward-only frozen encoder) every residual add -- and the macaron halves' 0.5 -- rides in the epilogue of the
        # sub-block's last conv (6 element-wise launches per block less)
        x = self.feed_forward_macaron(self.norm_ff_macaron(x), half_residual=x)
        x = self.self_attn(self.norm_mha(x), pos_emb, mask, residual=x)
        x = self.conv_module(self.norm_conv(x), residual=x)
        x = self.feed_forward(self.norm_ff(x), half_residual=x)
        return self.norm_final(x)


class ConformerLayers(nn.Module):
    def __init__(self, hidden_size, num_layers, kernel_size=31, num_heads=4, use_last_norm=True):
        super().__init__()
        self.hidden_size = hidden_size
        self.encoder_layers = nn.ModuleList([EncoderLayer(hidden_size, num_heads, kernel_size) for _ in range(num_layers)])
        self.layer_norm = LayerNormNCT(hidden_size) if use_last_norm else LinearNCT(hidden_size, hidden_size, xavier=False)
        self._pe = None

    def _pos_emb(self, T, device):
        """First T rows of the reversed 5000-long table (espnet_positional_embedding.py:23-46,107-112), as [1,D,T]."""
        if self._pe is None or self._pe.device != device:
            D, max_len = self.hidden_size, 5000
            pos = torch.arange(max_len - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
            div = torch.exp(torch.arange(0, D, 2, dtype=torch.float32) * -(math.log(10000.0) / D))
            pe = torch.zeros(max_len, D)
            pe[:, 0::2] = torch.sin(pos * div)
            pe[:, 1::2] = torch.cos(pos * div)
            self._pe = pe.t().contiguous().to(device)              # [D, 5000]
        c = getattr(self, "_pe_T", None)
        if c is None or c[0] != (T, self._pe.data_ptr()):
            c = self._pe_T = ((T, self._pe.data_ptr()), self._pe[None, :, :T].contiguous())     # (one copy per length, reused:
        return c[1]                                                   # the attention layers key their cached projection on it)

    def forward(self, x):
        """x [B,D,T] -> [B,D,T]  (conformer.py:37-53)."""
        nonpadding = x.abs().sum(1) > 0                             # [B,T]
        pos_emb = self._pos_emb(x.shape[-1], x.device)
        x = x * math.sqrt(self.hidden_size)
        for l in self.encoder_layers:
            x = l(x, pos_emb, nonpadding)
        return self.layer_norm(x) * nonpadding[:, None, :].float()


def _asr_decoder_spec(hidden, n_layers, dict_size, ffn_kernel=9):
    """state_dict keys of TransformerASRDecoder (modules/asr/seq2seq.py:11-33) -- stored, never executed on the path."""
    spec = [("asr_decoder.embed_positions._float_tensor", (1,), True)]
    for i in range(n_layers):
        p = f"asr_decoder.layers.{i}.op."
        spec += [(p + "layer_norm1.weight", (hidden,), False), (p + "layer_norm1.bias", (hidden,), False),
                 (p + "self_attn.in_proj_weight", (3 * hidden, hidden), False),
                 (p + "self_attn.out_proj.weight", (hidden, hidden), False),
                 (p + "layer_norm2.weight", (hidden,), False), (p + "layer_norm2.bias", (hidden,), False),
                 (p + "encoder_attn.in_proj_weight", (3 * hidden, hidden), False),
                 (p + "encoder_attn.out_proj.weight", (hidden, hidden), False),
                 (p + "layer_norm3.weight", (hidden,), False), (p + "layer_norm3.bias", (hidden,), False),
                 (p + "ffn.ffn_1.1.weight", (4 * hidden, hidden, ffn_kernel), False),
                 (p + "ffn.ffn_1.1.bias", (4 * hidden,), False),
                 (p + "ffn.ffn_2.weight", (hidden, 4 * hidden), False), (p + "ffn.ffn_2.bias", (hidden,), False)]
    spec += [("asr_decoder.layer_norm.weight", (hidden,), False), ("asr_decoder.layer_norm.bias", (hidden,), False),
             ("asr_decoder.project_out_dim.weight", (dict_size, hidden), False)]
    return spec


class VCASR(nn.Module):
    def __init__(self, dict_size, n_mel_bins, hparams):
        super().__init__()
        self.hidden_size = hparams["hidden_size"]
        if hparams["asr_enc_type"] != "conformer":
            raise NotImplementedError("vae_global_mle_eng uses asr_enc_type=conformer (vc_ppg.yaml)")
        self.mel_prenet = Prenet(n_mel_bins, self.hidden_size, strides=hparams["mel_strides"])
        self.content_encoder = ConformerLayers(self.hidden_size, hparams["asr_enc_layers"], 31,
                                               use_last_norm=hparams["asr_last_norm"])
        self.token_embed = nn.Embedding(dict_size, self.hidden_size, padding_idx=0)
        for key, shape, is_buf in _asr_decoder_spec(self.hidden_size, hparams["asr_dec_layers"], dict_size,
                                                    hparams.get("dec_ffn_kernel_size", 9)):
            attach_opaque(self, key, shape, buffer=is_buf)

    def train(self, mode=True):
        return super().train(False)          # always eval: frozen pretrained extractor (svb_vae_task.py:559)

    @torch.no_grad()
    def forward(self, mel_input, prev_tokens=None):
        """mel_input [B,T,80] -> {'h_content': [B, H, T/2] (NCT)}  (vc_modules.py:75-80, encoder only)."""
        if prev_tokens is not None:
            raise NotImplementedError("the ASR decoder is not on the hot path")
        nonpad = (mel_input.abs().sum(-1) != 0).float()
        x, _ = self.mel_prenet(mel_input.transpose(1, 2).contiguous(), nonpad)
        return {"h_content": self.content_encoder(x)}

"""Parameter-holding building blocks.  Parameter names and shapes are the reference's (so checkpoints load both
ways); the compute goes through neuralsvb_amd.functional (HIP kernels).

Everything here works on [B, C, T] ("NCT") tensors -- the layout the reference's Conv1d uses -- so the conv
kernels read time-contiguous rows; nn.Linear layers of the reference become 1x1 convs over that layout.
"""
import math

import torch
from torch import nn

from .. import functional as SF


class Conv1d(nn.Module):
    """torch.nn.Conv1d / weight_norm(Conv1d) parameter layout: weight [Cout, Cin/groups, k] (+bias) or
    weight_g [Cout,1,1] + weight_v (+bias)  (reference: modules/fastspeech/fs2_vae.py:44-59)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 weight_norm=False, init_std=None):
        super().__init__()
        ref = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        if init_std is not None:
            ref.weight.data.normal_(0.0, init_std)
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.is_weight_norm = weight_norm
        self.precision = None           # None: the global conv_precision; "fp32" pins this layer (set_layer_precision)
        if weight_norm:
            w = ref.weight.data
            self.weight_g = nn.Parameter(w.flatten(1).norm(dim=1).view(-1, 1, 1))
            self.weight_v = nn.Parameter(w.clone())
        else:
            self.weight = ref.weight
        self.bias = ref.bias if bias else None

    def forward(self, x, in_slope=None, out_act=SF.ACT_NONE, out_slope=0.0, residual=None, mask=None):
        with SF.precision_scope(self.precision):
            return self._forward(x, in_slope, out_act, out_slope, residual, mask)

    def _forward(self, x, in_slope, out_act, out_slope, residual, mask):
        if self.is_weight_norm:
            return SF.conv1d(x, self.weight_v, self.bias, self.stride, self.padding, self.dilation, self.groups,
                             weight_g=self.weight_g, in_slope=in_slope, out_act=out_act, out_slope=out_slope,
                             residual=residual, mask=mask)
        return SF.conv1d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups,
                         in_slope=in_slope, out_act=out_act, out_slope=out_slope, residual=residual, mask=mask)

    def remove_weight_norm(self):
        """Fold g*v/||v|| into a plain weight (reference vocoders/hifigan.py:29)."""
        if not self.is_weight_norm:
            return
        v, g = self.weight_v.data, self.weight_g.data
        w = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))
        del self.weight_g, self.weight_v
        self.weight = nn.Parameter(w)
        self.is_weight_norm = False


class ConvTranspose1d(nn.Module):
    """torch.nn.ConvTranspose1d layout: weight [Cin, Cout, k]; weight_norm (dim 0) -> weight_g [Cin,1,1]
    (reference modules/hifigan/hifigan.py:124-125, vae_models.py:115-120)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, weight_norm=False,
                 init_std=None):
        super().__init__()
        ref = nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        if init_std is not None:
            ref.weight.data.normal_(0.0, init_std)
        self.stride, self.padding = stride, padding
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.is_weight_norm = weight_norm
        self.precision = None
        if weight_norm:
            w = ref.weight.data
            self.weight_g = nn.Parameter(w.flatten(1).norm(dim=1).view(-1, 1, 1))
            self.weight_v = nn.Parameter(w.clone())
        else:
            self.weight = ref.weight
        self.bias = ref.bias if bias else None

    def forward(self, x, in_slope=None, mask=None):
        with SF.precision_scope(self.precision):
            return self._forward(x, in_slope, mask)

    def _forward(self, x, in_slope, mask):
        if self.is_weight_norm:
            return SF.conv_transpose1d(x, self.weight_v, self.bias, self.stride, self.padding, weight_g=self.weight_g,
                                       in_slope=in_slope, mask=mask)
        return SF.conv_transpose1d(x, self.weight, self.bias, self.stride, self.padding, in_slope=in_slope, mask=mask)

    def remove_weight_norm(self):
        if not self.is_weight_norm:
            return
        v, g = self.weight_v.data, self.weight_g.data
        w = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))
        del self.weight_g, self.weight_v
        self.weight = nn.Parameter(w)
        self.is_weight_norm = False


class LinearNCT(nn.Module):
    """nn.Linear parameters (weight [out, in], bias) applied over the channel dim of [B, C, T] as a 1x1 conv.
    xavier-uniform init as the reference's `Linear` factory (modules/commons/common_layers.py:81-86)."""

    def __init__(self, in_features, out_features, bias=True, xavier=True):
        super().__init__()
        ref = nn.Linear(in_features, out_features, bias)
        if xavier:
            nn.init.xavier_uniform_(ref.weight)
            if bias:
                nn.init.constant_(ref.bias, 0.0)
        self.weight = ref.weight
        self.bias = ref.bias if bias else None
        self.precision = None

    def forward(self, x, out_act=SF.ACT_NONE, mask=None, residual=None):
        with SF.precision_scope(self.precision):
            return SF.conv1d(x, self.weight[:, :, None], self.bias, out_act=out_act, mask=mask, residual=residual)


class LayerNormNCT(nn.Module):
    """nn.LayerNorm(C) parameters, applied over the channel dim of [B, C, T] (frozen PPG encoder; forward only)."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.eps = eps

    def forward(self, x):
        return SF.layer_norm_nct(x, self.weight, self.bias, self.eps)


def attach_opaque(root, dotted, shape, buffer=False, dtype=torch.float32):
    """Register a parameter/buffer under a dotted state_dict key without building its module (weights that live in
    the reference checkpoint but are never executed on the hot path, e.g. vc_asr.asr_decoder.*)."""
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, nn.Module())
        m = getattr(m, p)
    t = torch.zeros(shape, dtype=dtype)
    if buffer:
        m.register_buffer(parts[-1], t)
    else:
        m.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


def set_layer_precision(root, rules):
    """rules: {module-name prefix: "fp32" | "bf16x3" | None}.  Pins the conv arithmetic of every layer below a prefix
    (longest prefix wins); layers that match no prefix follow the global `conv_precision`."""
    n = 0
    for name, m in root.named_modules():
        if not hasattr(m, "precision"):
            continue
        best = None
        for pre in rules:
            if (name == pre or name.startswith(pre + ".") or pre == "") and (best is None or len(pre) > len(best)):
                best = pre
        if best is not None:
            m.precision = rules[best]
            n += 1
    return n

"""Multi-window mel-spectrogram critic (the GAN discriminator of the vae_global_mle_eng step).

Drop-in for reference modules/fastspeech/multi_window_disc.py:154-199 `Discriminator` as built by
tasks/tts/fs2_adv.py:22-35 (uncond, norm 'in', reduction 'stack'): identical state_dict keys
(`discriminator.conv_layers.<w>.model.<b>.{0,3}.*`, `...adv_layer.*`), call signature and result dict.
Semantics kept: one np.random window start per window length shared by the whole batch (:144-148), x_len from
non-zero frames (:190), Dropout2d(0.25) on whole channels in train mode (:23), `y=None` when a clip is shorter
than a window (:140-142).

Every block runs on the HIP kernels: space-to-depth planes -> tap-table implicit-GEMM conv with the LeakyReLU in its
epilogue -> one crop + Dropout2d-factor + InstanceNorm2d pass (SF.critic_block); the score layer is one reduction kernel
(SF.plane_score).  Only the 'bn' norm variant (not used by vae_global_mle_eng) keeps torch's BatchNorm2d.
"""
import numpy as np
import torch
from torch import nn

from .. import functional as SF


def _adv_score(adv_layer, h):
    """tower.adv_layer(h.flatten(1)) -- nn.Linear(C*H*W, 1), multi_window_disc.py:62-64 -- as one row-reduction kernel
    over the channel-major feature maps (rocBLAS picks a 120 us tile GEMM for this GEMV)."""
    return SF.plane_score(h, adv_layer.weight, adv_layer.bias)


def _block(blk, h, planes=None, s2d_out=False):
    """Conv2d 3x3 s2 p1 -> LeakyReLU(0.2) -> Dropout2d [-> InstanceNorm2d | BatchNorm2d] (multi_window_disc.py:14-31).
    planes / s2d_out: the block reads / writes the space-to-depth layout of the conv directly (SF.critic_block).
    Odd planes (a window length such as 20 or 36 reaches them at the 2nd / 3rd block, or an odd number of mel bins) and
    kernels other than 3x3 are legal in the reference: they take the general im2col conv + the per-(clip, channel) factor of
    Dropout2d + the stock norm module."""
    conv, drop = blk[0], blk[2]
    norm = blk[3] if len(blk) > 3 else None
    p = drop.p if drop.training else 0.0
    N, C, H, W = planes if planes is not None else h.shape
    if not (_fast_plane(conv, H, W)):
        assert planes is None and not s2d_out
        h = SF.conv2d_lrelu(h, conv.weight, conv.bias, 2, tuple(conv.padding), 0.2)
        if p:
            h = h * SF.dropout2d_keep(h.size(0), h.size(1), p, h.device)[:, :, None, None]
        return norm(h) if norm is not None else h
    if norm is None or isinstance(norm, nn.InstanceNorm2d):
        gamma, beta, eps = (norm.weight, norm.bias, norm.eps) if norm is not None else (None, None, 1e-5)
        return SF.critic_block(h, conv.weight, conv.bias, 0.2, p, gamma, beta, eps, planes=planes, s2d_out=s2d_out)
    assert planes is None and not s2d_out
    return norm(SF.critic_block(h, conv.weight, conv.bias, 0.2, p, None, None))


def _fast_plane(conv, H, W):
    return tuple(conv.kernel_size) == (3, 3) and H % 2 == 0 and W % 2 == 0


FUSED_CROP = True        # forward_many: crop + stack + space-to-depth of all windows in one pass per direction


def _tower(tower, h, want_fmaps, fmaps, planes=None):
    """The three blocks of one window's tower.  Without feature maps (and without BatchNorm2d) the blocks hand each other the
    conv's space-to-depth layout directly; the last block returns the plain feature map for the score layer.
    planes: `h` already is the first block's space-to-depth input (SF.window_crop_s2d)."""
    blocks = list(tower.model)
    chain = not want_fmaps and all(len(b) <= 3 or isinstance(b[3], nn.InstanceNorm2d) for b in blocks)
    for i, blk in enumerate(blocks):
        if chain and i + 1 < len(blocks):
            N, C, H, W = planes if planes is not None else h.shape
            if _fast_plane(blk[0], H, W) and (H // 2) % 2 == 0 and (W // 2) % 2 == 0:
                h, planes = _block(blk, h, planes, s2d_out=True)
                continue
        h = _block(blk, h, planes)
        planes = None
        if want_fmaps:
            fmaps.append(h)
    return h


def _tower_score(tower, x4, planes):
    """The whole tower + score layer as one autograd node (SF.critic_tower) when every block chains through the conv's
    space-to-depth layout (3x3 kernels, planes even down to the last block, InstanceNorm2d or no norm); None otherwise."""
    if not SF.FUSE_CRITIC_TOWER:
        return None
    blocks, (N, C, H, W) = list(tower.model), planes
    spec = []
    for i, blk in enumerate(blocks):
        conv, drop = blk[0], blk[2]
        norm = blk[3] if len(blk) > 3 else None
        if not _fast_plane(conv, H, W) or (norm is not None and not isinstance(norm, nn.InstanceNorm2d)):
            return None
        if i + 1 < len(blocks) and ((H // 2) % 2 or (W // 2) % 2):
            return None
        spec.append((conv.weight, conv.bias, drop.p if drop.training else 0.0, norm.weight if norm is not None else None,
                     norm.bias if norm is not None else None, norm.eps if norm is not None else 1e-5))
        H, W = H // 2, W // 2
    return SF.critic_tower(x4, planes, spec, tower.adv_layer.weight, tower.adv_layer.bias)


def _critic_tower(time_length, freq_length, kernel, c_in, hidden, norm_type, reduction):
    """One window's conv tower as bare containers named like the reference's Discriminator2DFactory (:6-44)."""
    tower = nn.Module()
    blocks = []
    for b in range(3):
        mods = [nn.Conv2d(c_in if b == 0 else hidden, hidden, kernel, (2, 2), (kernel[0] // 2, kernel[1] // 2)),
                nn.LeakyReLU(0.2), nn.Dropout2d(0.25)]
        if b > 0 and norm_type == "in":
            mods.append(nn.InstanceNorm2d(hidden, affine=True))
        elif b > 0 and norm_type == "bn":
            mods.append(nn.BatchNorm2d(hidden, 0.8))
        blocks.append(nn.Sequential(*mods))
    tower.model = nn.ModuleList(blocks)
    t8, f8 = time_length // 8, (freq_length + 7) // 8
    tower.adv_layer = nn.Linear(hidden * f8 * (t8 if reduction != "none" else 1), 1)
    return tower


class Discriminator(nn.Module):
    def __init__(self, time_lengths=(32, 64, 128), freq_length=80, cond_size=0, kernel=(3, 3), c_in=1, hidden_size=128,
                 norm_type="bn", reduction="sum", uncond_disc=True):
        super().__init__()
        if cond_size > 0 or not uncond_disc:
            raise NotImplementedError("vae_global_mle_eng uses the unconditional critic only (use_cond_disc: false)")
        if reduction not in ("sum", "stack"):
            raise NotImplementedError(reduction)
        self.time_lengths, self.reduction, self.norm_type = list(time_lengths), reduction, norm_type
        self.discriminator = nn.Module()
        self.discriminator.conv_layers = nn.ModuleList(
            [_critic_tower(tl, freq_length, kernel, c_in, hidden_size, norm_type, reduction) for tl in self.time_lengths])

    def forward(self, x, cond=None, start_frames_wins=None, starts_dev=None, longest=None, want_fmaps=True):
        """x [B,T,n_mels] (or [B,1,T,n_mels]).  Returns {'y': [B,1,W] or None, 'y_c': None, 'h': fmaps,
        'start_frames_wins': [[s]*B per window]}.

        `longest` (max number of non-zero frames over the batch) and `start_frames_wins` may be supplied by a caller
        that already knows them on the host (the task draws the step's window starts up front, in the reference's
        order), which removes the device->host sync of :190.  `starts_dev` (int64 [n_windows], device) makes the window
        crop a device-side gather so that the call can be replayed from a captured hipGraph with fresh starts.
        `want_fmaps=False` (the training task: it only reads 'y') returns 'h': [] and lets the blocks of a tower hand each
        other the conv's own input layout."""
        if x.dim() == 3:
            x = x[:, None]
        if longest is None:
            longest = int(x.sum([1, -1]).ne(0).int().sum(-1).max().item())
        starts = list(start_frames_wins) if start_frames_wins is not None else [None] * len(self.time_lengths)
        scores, fmaps = [], []
        for w, (tower, wl) in enumerate(zip(self.discriminator.conv_layers, self.time_lengths)):
            if longest - wl < 0:
                continue
            if starts[w] is None:
                starts[w] = [int(np.random.randint(low=0, high=longest - wl + 1))] * x.size(0)
            if starts_dev is not None:
                h = x.index_select(2, starts_dev[w] + torch.arange(wl, device=x.device))
            else:
                s = starts[w][0]
                h = x[:, :, s:s + wl]
            h = _tower(tower, h, want_fmaps, fmaps)
            scores.append(_adv_score(tower.adv_layer, h))
        y = None
        if len(scores) == len(self.time_lengths):
            y = torch.stack(scores, -1) if self.reduction == "stack" else sum(scores)
        return {"y": y, "y_c": None, "h": fmaps, "start_frames_wins": starts}

    def forward_many(self, calls, want_fmaps=True):
        """Several independent critic calls -- [(x, start_frames_wins, starts_dev, longest), ...] with equal shapes and all
        windows present -- as ONE pass per tower over the stacked crops [n*B,1,wl,80].  Every layer of the tower is
        per-clip (Conv2d, LeakyReLU, Dropout2d, InstanceNorm2d), so each call's scores equal those of a separate
        forward() (with Dropout2d off; with it on only the order of the mask draws differs).  Returns one result dict per
        call ('h' holds the stacked feature maps and is shared)."""
        xs = [c[0][:, None] if c[0].dim() == 3 else c[0] for c in calls]
        B = xs[0].size(0)
        scores, fmaps = [], []
        fused = None
        if (FUSED_CROP and all(c[2] is None for c in calls) and all(x.dim() == 4 and x.size(1) == 1 for x in xs)
                and xs[0].size(3) % 2 == 0 and all(wl % 2 == 0 for wl in self.time_lengths) and len(xs) <= 8
                and all(tuple(t.model[0][0].kernel_size) == (3, 3) for t in self.discriminator.conv_layers)
                and all(len(b) <= 3 or isinstance(b[3], nn.InstanceNorm2d) for t in self.discriminator.conv_layers
                        for b in t.model)):
            fused = SF.window_crop_s2d([x[:, 0] for x in xs], self.time_lengths,
                                       [[c[1][w][0] for c in calls] for w in range(len(self.time_lengths))])
        with SF.dropout2d_pool(*self._keep_fields(len(xs) * B), xs[0].device):
            return self._forward_many_towers(calls, xs, B, fused, want_fmaps)

    def _keep_fields(self, n):
        """(shapes, p) of the Dropout2d fields a stacked pass over all towers draws, in order (one p for all blocks, else none)."""
        shapes, ps = [], set()
        for tower in self.discriminator.conv_layers:
            for blk in tower.model:
                if blk[2].training and blk[2].p:
                    shapes.append((n, blk[0].out_channels))
                    ps.add(blk[2].p)
        return (shapes, ps.pop()) if len(ps) == 1 else ([], 0.0)

    def _forward_many_towers(self, calls, xs, B, fused, want_fmaps):
        scores, fmaps = [], []
        for w, (tower, wl) in enumerate(zip(self.discriminator.conv_layers, self.time_lengths)):
            if fused is not None:
                x4, planes = fused[w]
                if want_fmaps:         # (feature maps are per block: the first block runs from the planes, unchained)
                    h = _block(tower.model[0], x4, planes)
                    fmaps.append(h)
                    for blk in list(tower.model)[1:]:
                        h = _block(blk, h)
                        fmaps.append(h)
                else:
                    y = _tower_score(tower, x4, planes)
                    if y is not None:
                        scores.append(y)
                        continue
                    h = _tower(tower, x4, want_fmaps, fmaps, planes=planes)
                scores.append(_adv_score(tower.adv_layer, h))
                continue
            crops = []
            for x, (_, starts, starts_dev, _) in zip(xs, calls):
                if starts_dev is not None:
                    crops.append(x.index_select(2, starts_dev[w] + torch.arange(wl, device=x.device)))
                else:
                    s = starts[w][0]
                    crops.append(x[:, :, s:s + wl])
            h = torch.cat(crops, 0)
            h = _tower(tower, h, want_fmaps, fmaps)
            scores.append(_adv_score(tower.adv_layer, h))
        y = torch.stack(scores, -1) if self.reduction == "stack" else sum(scores)
        return [{"y": y[i * B:(i + 1) * B], "y_c": None, "h": fmaps, "start_frames_wins": list(c[1]), "y_all": y}
                for i, c in enumerate(calls)]

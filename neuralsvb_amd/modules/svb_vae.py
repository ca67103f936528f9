"""MleSVBVAE -- the vae_global_mle_eng model.

Host-side mirror of reference modules/voice_conversion/svb_vae.py (SVBVAE :13-162, GlobalSVBVAE :172-248,
MleSVBVAE :251-312) and modules/commons/common_layers.py (ConvStacks :672-707, ConvBlock :739-773): same
constructor (`MleSVBVAE(dict_size)` + hparams), call signature, returned dict-of-dicts and state_dict keys.
Internally everything is kept in [B, C, T] so the HIP conv kernels read time-contiguous rows; the `[B, T, C]`
tensors of the reference API are produced as (free) transposed views at the boundary.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as SF
from .fs2_vae import GlobalFVAE, GlobalLatentMap, bn_groups
from .layers import Conv1d, LinearNCT
from .vc_asr import VCASR


SPLIT_STACKED = True     # stacked ways: per-way mel_out views through SF.split_stacked_ways (one gradient buffer for both)
PPG_SIDE_STREAM = True   # prepare_condition: the frozen PPG encoder (no grad, T/2 frames: it under-fills the chip) runs on its own
                         # stream beside the pitch encoder -- the two branches only meet at the conditioning projection
_PPG_STREAMS = {}
PPG_GRAPH = False        # hparam `ppg_graph`: the look-ahead run of the frozen PPG encoder (prefetch_content) replayed from a captured
                         # hipGraph per input shape instead of being issued launch by launch -- ~75 launches / 1.6 ms of HOST time per
                         # step become one replay.  Frozen weights, persistent packed images and table-driven tiles make the
                         # captured sequence the eager one: identical results.
PPG_GRAPH_MAX_SHAPES = 8
FUSED_GN = True          # ConvBlock: GroupNorm + ReLU + residual as one HIP pass per direction


class _ConvNorm(nn.Module):          # reference common_layers.ConvNorm: holds `.conv`
    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = Conv1d(cin, cout, k, padding=(k - 1) // 2)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain("linear"))


class ConvBlock(nn.Module):
    """conv k5 -> GroupNorm(C/16) -> ReLU  (common_layers.py:739-773, norm='gn', dropout=0)."""

    def __init__(self, n_chans, kernel_size):
        super().__init__()
        self.conv = _ConvNorm(n_chans, n_chans, kernel_size)
        self.norm = nn.GroupNorm(n_chans // 16, n_chans)

    def forward(self, x, residual=None):
        """residual: added to the block's output (ConvStacks' `x + f(x)`) inside the GroupNorm+ReLU pass."""
        h = self.conv.conv(x)
        n = self.norm
        if FUSED_GN and n.affine and n.num_channels // n.num_groups <= 64:
            return SF.group_norm_relu(h, n.weight, n.bias, n.num_groups, n.eps, residual=residual)
        y = F.relu(n(h))
        return y if residual is None else residual + y


class ConvStacks(nn.Module):
    def __init__(self, idim, n_chans, odim, n_layers=5, kernel_size=5):
        super().__init__()
        self.in_proj = LinearNCT(idim, n_chans)
        self.conv = nn.ModuleList([ConvBlock(n_chans, kernel_size) for _ in range(n_layers)])
        self.out_proj = LinearNCT(n_chans, odim)

    def forward(self, x):
        """x [B,C,T] -> [B,odim,T]   (common_layers.py:688-707, res=True)."""
        x = self.in_proj(x)
        for f in self.conv:
            x = f(x, residual=x)
        return self.out_proj(x)


class MleSVBVAE(nn.Module):
    def __init__(self, dict_size, hparams=None):
        super().__init__()
        if hparams is None:
            from neuralsvb_amd.utils.hparams import hparams as _hp
            hparams = _hp
        self.hp = hp = hparams
        self.hidden_size = H = hp["hidden_size"]
        self.c_content = self.c_out = hp["audio_num_mel_bins"]
        self.pitch_embed = nn.Embedding(300, H, padding_idx=0)
        nn.init.normal_(self.pitch_embed.weight, mean=0, std=H ** -0.5)
        nn.init.constant_(self.pitch_embed.weight[0], 0)
        self.pitch_encoder = ConvStacks(idim=H, n_chans=H, odim=H, n_layers=3)
        self.vc_asr = VCASR(dict_size, self.c_content, hp)
        ups = []
        for scale in hp["mel_strides"]:
            if scale > 1:
                ups.append(nn.Sequential(nn.Upsample(scale_factor=scale, mode="nearest"),
                                         Conv1d(H, H, scale * 2 + 1, padding=scale), nn.ReLU(), nn.BatchNorm1d(H)))
        ups.append(Conv1d(H, H, 5, padding=2))
        self.upsample_layer = nn.Sequential(*ups)
        self.spk_embed_proj = nn.Linear(256, H)
        nn.init.xavier_uniform_(self.spk_embed_proj.weight)
        nn.init.constant_(self.spk_embed_proj.bias, 0.0)
        self.encoded_embed_proj = LinearNCT(3 * H, H)
        self.vae_model = GlobalFVAE(in_out_channels=self.c_content, hidden_channels=hp["fvae_enc_dec_hidden"],
                                    latent_size=hp["latent_size"], kernel_size=hp["fvae_kernel_size"],
                                    enc_n_layers=hp["fvae_enc_n_layers"], dec_n_layers=hp["fvae_dec_n_layers"],
                                    gin_channels=H, use_prior_glow=False, strides=[4])
        self.z_mapping_function = GlobalLatentMap(hp["latent_size"])
        self.stack_ways = bool(hp.get("stack_ways", True))     # see forward()

    # ---- conditioning (svb_vae.py:60-86); all tensors NCT --------------------------------------------------------
    # ---- the frozen PPG encoder one step ahead ----------------------------------------------------------------------------
    @staticmethod
    def _content_key(*mels):
        return tuple((m.data_ptr(), tuple(m.shape), m._version) for m in mels)

    def prefetch_content(self, amateur_mel, prof_mel):
        """Run the frozen PPG encoder on the mels of the NEXT batch now, on its side stream (the Trainer calls this between the
        forward and the backward of the current step).  Its output depends on the batch alone -- the encoder never trains
        (svb_vae.py:66-72) -- but everything in the next forward waits for it: 1.8 ms at the head of the step's dependency chain
        that now run beside this step's backward instead.  The result is consumed (once) by the forward that finds the same
        mel tensors, unchanged; any other forward simply computes it."""
        if not (PPG_SIDE_STREAM and amateur_mel.is_cuda and not SF.CAPTURING):
            return
        stacked = self.stack_ways and amateur_mel.shape == prof_mel.shape
        jobs = [((amateur_mel, prof_mel), None)] if stacked else [((amateur_mel,), None), ((prof_mel,), None)]
        cur = torch.cuda.current_stream(amateur_mel.device)
        side = _PPG_STREAMS.get(amateur_mel.device.index)
        if side is None:
            side = _PPG_STREAMS[amateur_mel.device.index] = torch.cuda.Stream(amateur_mel.device)
        side.wait_stream(cur)
        cache = self.__dict__.setdefault("_content_cache", {})
        cache.clear()
        with torch.no_grad(), torch.cuda.stream(side):
            for mels, _ in jobs:
                for m in mels:
                    m.record_stream(side)
                h = self._ppg_graph_run(mels, side) if PPG_GRAPH else None
                if h is None:
                    x = torch.cat(list(mels)) if len(mels) > 1 else mels[0]
                    h = self.vc_asr(x)["h_content"].detach()
                cache[self._content_key(*mels)] = (h, mels)

    def _ppg_graph_run(self, mels, side):
        """The frozen encoder on cat(mels) through a hipGraph captured once per input shape (static input / output buffers; the
        current stream is `side`).  Returns the (static) output tensor, or None when this shape is not graphed (first sight of
        a shape runs eagerly -- lazy caches fill, tiles outside the table are measured --, the second captures; more than
        PPG_GRAPH_MAX_SHAPES shapes, or a capture that fails, fall back to eager launches for good).

        The graph's output buffer is overwritten by the NEXT replay; callers get a copy of it."""
        st = self.__dict__.setdefault("_ppg_graphs", {"seen": {}, "graphs": {}, "off": False})
        if st["off"] or self.vc_asr.training:
            return None
        key = tuple((tuple(m.shape), m.dtype) for m in mels)
        ent = st["graphs"].get(key)
        if ent is None:
            st["seen"][key] = st["seen"].get(key, 0) + 1
            if st["seen"][key] < 2 or len(st["graphs"]) >= PPG_GRAPH_MAX_SHAPES:
                return None
            try:
                x_static = torch.cat(list(mels)) if len(mels) > 1 else mels[0].clone()
                g = torch.cuda.CUDAGraph()
                side.synchronize()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    h_static = self.vc_asr(x_static)["h_content"].detach()
                ent = st["graphs"][key] = (g, x_static, h_static)
            except Exception as e:            # a sync or an unsupported call inside the encoder: eager launches from here on
                st["off"] = True
                import warnings
                warnings.warn(f"PPG encoder graph capture failed ({e!r}); falling back to eager launches")
                return None
        g, x_static, h_static = ent
        if len(mels) > 1:
            torch.cat(list(mels), out=x_static)
        else:
            x_static.copy_(mels[0])
        g.replay()
        # (a copy, not the static buffer: two jobs of one prefetch may share a shape key -- `stack_ways: false` with equally long
        #  amateur / professional mels -- and a consumer may save the tensor for its weight gradient; the next replay overwrites
        #  the static one.  One 18 MB copy on the PPG stream.)
        return h_static.clone()

    def _cached_content(self, key):
        cache = self.__dict__.get("_content_cache")
        return cache.pop(key, None) if cache and key is not None else None

    def prepare_condition(self, mels_content, pitch, spk_ids, groups=1, content_key=None):
        T = pitch.shape[1]
        pe = self.pitch_embed          # nn.Embedding(300, H, padding_idx=0) + transpose, as one gather into [B,H,T]
        side = None
        hit = self._cached_content(content_key)
        if hit is not None:
            # computed one step ahead on the PPG stream (prefetch_content): the current stream waits where it is first read
            h, side = hit[0], _PPG_STREAMS[mels_content.device.index]
            cur = torch.cuda.current_stream(mels_content.device)
        elif PPG_SIDE_STREAM and mels_content.is_cuda and not SF.CAPTURING:
            # independent branches (svb_vae.py:66-72): PPG encoder on a side stream that first catches up with the producers of
            # the mel, pitch encoder on the current one; the current stream waits where the content features are first read
            cur = torch.cuda.current_stream(mels_content.device)
            side = _PPG_STREAMS.get(mels_content.device.index)
            if side is None:
                side = _PPG_STREAMS[mels_content.device.index] = torch.cuda.Stream(mels_content.device)
            side.wait_stream(cur)
            mels_content.record_stream(side)
            with torch.cuda.stream(side):
                h = self.vc_asr(mels_content)["h_content"].detach()
        h_pitch = self.pitch_encoder(SF.embedding_nct(pitch, pe.weight, pe.padding_idx))
        if side is not None:
            cur.wait_stream(side)
            h.record_stream(cur)
        else:
            h = self.vc_asr(mels_content)["h_content"].detach()
        for m in self.upsample_layer:
            if isinstance(m, nn.Sequential):
                h = SF.upsample_nearest_nct(h, int(m[0].scale_factor))
                h = bn_groups(m[3], m[1](h, out_act=SF.ACT_RELU), groups)
            else:
                h = m(h)
        h_content = h[:, :, :mels_content.shape[1]]
        sp = self.spk_embed_proj      # nn.Linear(256, H) parameters (reference state_dict), computed as a 1x1 HIP conv on [B,256,1]
        h_style = SF.conv1d(spk_ids[:, :, None].contiguous(), sp.weight[:, :, None], sp.bias).expand(-1, -1, T)
        return {"h_pitch": h_pitch, "h_content": h_content, "h_style": h_style, "tgt_nonpadding": (pitch > 0).float()}

    def _cond_sum(self, h_pitch, h_content, h_style):
        return self.encoded_embed_proj(torch.cat([h_pitch, h_content, h_style], 1))

    def normal_vae(self, tgt_mel, c, eps=None, groups=1):
        cond = self._cond_sum(c["h_pitch"], c["h_content"], c["h_style"])
        mel_out, kl, z_p, m_q, logs_q, mask_sqz, z_q = self.vae_model(
            tgt_mel.transpose(1, 2).contiguous(), c["tgt_nonpadding"], g=cond, eps=eps, groups=groups)
        out = {"mel_out": mel_out.transpose(1, 2), "kl": kl, "z_p": z_p, "m_q": m_q, "logs_q": logs_q,
               "x_mask_sqz": mask_sqz, "z_q": z_q}
        if groups > 1 and SPLIT_STACKED:
            # the ways' [B,T,80] outputs as views whose gradients are assembled in one [groups*B,80,T] buffer
            out["_mel_ways"] = SF.split_stacked_ways(mel_out, groups)
        return out

    def forward(self, amateur_mel=None, prof_mel=None, amateur_pitch=None, prof_pitch=None, amateur_spk_id=None,
                prof_spk_id=None, a2p_alignment=None, p2a_alignment=None, infer=False, disable_map=False, **kwargs):
        """Same contract as reference svb_vae.py:258-312.  Extra (optional) kwargs: eps_a2a / eps_p2p inject the
        N(0,1) draws of the encoder (vae_models.py:104) for parity tests."""
        ways = kwargs["concurrent_ways"]
        ret = {}
        eps_a, eps_p = kwargs.get("eps_a2a"), kwargs.get("eps_p2p")
        stacked = (self.stack_ways and amateur_mel.shape == prof_mel.shape and amateur_pitch.shape == prof_pitch.shape
                   and (eps_a is None) == (eps_p is None))
        if stacked:
            # The two voices share every weight and are independent per clip: run them as ONE launch sequence on the
            # stacked batch [amateur; professional] (half the launches, twice the work per launch).  The only
            # batch-coupled layers on the path, the train-mode BatchNorms, are applied per half (bn_groups).
            B = amateur_mel.shape[0]
            c2 = self.prepare_condition(torch.cat([amateur_mel, prof_mel]), torch.cat([amateur_pitch, prof_pitch]),
                                        torch.cat([amateur_spk_id, prof_spk_id]), groups=2,
                                        content_key=self._content_key(amateur_mel, prof_mel))
            ca, cp = {k: v[:B] for k, v in c2.items()}, {k: v[B:] for k, v in c2.items()}
        else:
            ca = self.prepare_condition(amateur_mel, amateur_pitch, amateur_spk_id, content_key=self._content_key(amateur_mel))
            cp = self.prepare_condition(prof_mel, prof_pitch, prof_spk_id, content_key=self._content_key(prof_mel))
        self._last_conds = (ca, cp)
        self._last_stacked_kl = None
        if stacked and "a2a" in ways and "p2p" in ways:
            o2 = self.normal_vae(torch.cat([amateur_mel, prof_mel]), c2, None if eps_a is None else torch.cat([eps_a, eps_p]),
                                 groups=2)
            self._last_stacked_kl = o2["kl"]          # [2]: both ways' KL terms as one vector (the task sums from it)
            mel_ways = o2.pop("_mel_ways", None)
            for i, way in enumerate(("a2a", "p2p")):
                ret[way] = {k: (None if v is None else (v[i] if k == "kl" else v[i * B:(i + 1) * B])) for k, v in o2.items()}
                if mel_ways is not None:
                    ret[way]["mel_out"] = mel_ways[i]
        else:
            if "a2a" in ways:
                ret["a2a"] = self.normal_vae(amateur_mel, ca, eps_a)
            if "p2p" in ways:
                ret["p2p"] = self.normal_vae(prof_mel, cp, eps_p)
        if "a2p" in ways:
            out = {}
            z_a = ret["a2a"]["z_q"]
            m_p, logs_p = ret["p2p"]["m_q"], ret["p2p"]["logs_q"]
            if disable_map:
                print("here disable map!!!")
                z_map = z_a
            else:
                z_map = self.z_mapping_function(z_a, ca["h_style"])
            prof_dist = torch.distributions.Normal(m_p, logs_p.exp(), validate_args=False)   # no host-side constraint check (sync)
            out["mle"] = -prof_dist.log_prob(z_map).sum() / z_map.shape[0] / z_map.shape[1]
            idx = a2p_alignment[:, None, :].expand(-1, self.hidden_size, -1)          # gather along time (svb_vae.py:285,298)
            cond = self._cond_sum(cp["h_pitch"], torch.gather(ca["h_content"], 2, idx),
                                  ca["h_style"][:, :, :1].expand(-1, -1, cp["h_pitch"].shape[2]))
            mel = self.vae_model.decoder(z_map, cp["tgt_nonpadding"], g=cond)
            out["mel_out"] = mel.transpose(1, 2)
            out["logs_amateur_zq"], out["logs_prof_zq"] = ret["a2a"]["z_q"], ret["p2p"]["z_q"]
            ret["a2p"] = out
        return ret

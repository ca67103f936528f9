"""SVBVAEMleTask -- the vae_global_mle_eng training / validation / inference task.

Drop-in for `tasks.singing.svb_vae_task.SVBVAEMleTask` (reference tasks/singing/svb_vae_task.py:543-726) with its
inherited pieces flattened into one class: three optimizers (generator / mel-discriminator / latent map,
:84-118), three phases (:585-595), `run_model` (:120-165), LS-GAN helper steps (tasks/singing/svb_para.py:118-170),
mel losses (tasks/tts/fs2.py:143-175, SSIM of modules/commons/ssim.py:320-351), gradient clipping and scheduler
stepping hooks (:390-404).  The Trainer-facing hook API (tasks/base_task.py:131-355) is unchanged.

Differences that do not change results: loss terms stay on the GPU (no `.item()` per term per step -- they are
converted when something logs them), and the NaN/Inf guards of :665-672 are expressed with torch.where so the
step never synchronises the host.
"""
import json
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as SF
from ..modules.mel_disc import Discriminator
from ..modules.svb_vae import MleSVBVAE
from ..utils import ckpt_utils
from ..utils.hparams import hparams
from ..utils.schedulers import NoneSchedule, RSQRTSchedule
from .base_task import BaseTask, data_loader


class _Terms:
    """The loss terms of one optimizer pass as a few device vectors instead of one 0-d tensor per term.

    A piece is a 1-D tensor of raw terms with, per element, a key (None = slot that is neither logged nor summed), the
    factor its logged value carries (`scale`: the reference logs `l1 * lambda`, `kl * lambda_kl`, ...) and its weight in the
    total (`weight`: loss_weights of svb_vae_task.py:673-674).  total = dot(cat(pieces), scale*weight) and every logged
    value = one `cat * scale` launch + views, instead of ~3 launches forward and ~3 backward per term."""

    _cache = {}

    def __init__(self):
        self.vecs, self.keys, self.scale, self.weight = [], [], [], []

    def add(self, vec, keys, scale, weight=None, guard=False):
        vec = vec.reshape(-1)
        if guard:        # a non-finite KL / MLE term contributes no gradient (svb_vae_task.py:665-672)
            vec = torch.where(torch.isfinite(vec), vec, vec.detach())
        self.vecs.append(vec)
        self.keys += list(keys)
        self.scale += [float(x) for x in scale]
        self.weight += [1.0] * len(keys) if weight is None else [float(x) for x in weight]

    def __len__(self):
        return len(self.keys)

    def _const(self, vals, device):
        key = (tuple(vals), str(device))
        t = _Terms._cache.get(key)
        if t is None:
            t = _Terms._cache[key] = torch.tensor(vals, dtype=torch.float32, device=device)
        return t

    def total_and_logs(self):
        cat = self.vecs[0] if len(self.vecs) == 1 else torch.cat(self.vecs)
        total = torch.dot(cat, self._const([s * w for s, w in zip(self.scale, self.weight)], cat.device))
        logged = cat.detach() * self._const(self.scale, cat.device)
        return total, {k: logged[i] for i, k in enumerate(self.keys) if k is not None}


class SVBVAEMleTask(BaseTask):
    def __init__(self):
        super().__init__()
        from .dataset import MultiSpkEmbDataset
        self.dataset_cls = MultiSpkEmbDataset
        self.mse_loss_fn = nn.MSELoss()
        self._w_cache, self._lw_cache = {}, {}       # per-step target weights / per-pass loss-weight vectors
        self.loss_and_lambda = {}
        for part in hparams["mel_loss"].split("|"):
            name, _, lbd = part.partition(":")
            self.loss_and_lambda[name] = float(lbd) if lbd else 1.0
        self.vocoder = None
        self.model_out = self.model_out_gt = None
        self._step_rand = None          # host randoms of the current step, drawn up front by begin_step()
        self._rand_dev = None           # their device copies (static buffers; hipGraph replay reads them)

    # ------------------------------------------------------------------ model / optimizers
    def build_tts_model(self):
        phone_list = json.load(open(os.path.join(hparams["binary_data_dir"], "phone_set.json")))
        self.model = MleSVBVAE(len(phone_list) + 10, hparams)

    def build_disc_model(self):
        self.disc_params = []
        if hparams["mel_gan"]:
            self.mel_disc = Discriminator(time_lengths=[32, 64, 128][:hparams["disc_win_num"]], freq_length=80,
                                          hidden_size=hparams["mel_disc_hidden_size"], kernel=(3, 3),
                                          cond_size=hparams["hidden_size"] if hparams["use_cond_disc"] else 0,
                                          norm_type=hparams["disc_norm"], reduction=hparams["disc_reduction"])
            self.disc_params = list(self.mel_disc.parameters())

    def build_model(self):
        SF.set_precision(hparams.get("conv_precision", "fp32"))
        SF.STACK_EXECUTOR = bool(hparams.get("wn_stack_executor", True))
        SF.TOWER_EXECUTOR = bool(hparams.get("tower_executor", True))       # (A/B switch: the critic towers as one C-ABI call per direction)
        SF.S2_REGISTERED = bool(hparams.get("critic_s2_registered", True))
        from ..modules import vc_asr as _va
        _va.FUSE_QKV = bool(hparams.get("ppg_fuse_qkv", True))
        _va.POS_IN_KERNEL = bool(hparams.get("ppg_pos_in_kernel", True))
        from ..modules import svb_vae as _svb
        _svb.PPG_SIDE_STREAM = bool(hparams.get("overlap_ppg_encoder", True)) and os.environ.get("SVB_PPG_SIDE", "1") != "0"
        _svb.PPG_GRAPH = bool(hparams.get("ppg_graph", False))
        self.build_tts_model()
        if hparams.get("pretrain_asr_ckpt"):
            ckpt_utils.load_ckpt(self.model.vc_asr, hparams["pretrain_asr_ckpt"], model_name="model",
                                 force=hparams.get("pretrain_asr_required", True))
        self.model.vc_asr.eval()
        for p in self.model.vc_asr.parameters():
            p.requires_grad = False
        if hparams.get("load_ckpt", "") != "":
            self.load_ckpt(hparams["load_ckpt"], strict=False)
        self.build_disc_model()
        self.gen_params = [p for n, p in self.model.named_parameters()
                           if "vc_asr" not in n and "z_mapping_function" not in n]
        self.mapping_params = list(self.model.z_mapping_function.parameters())
        return self.model

    def configure_optimizers(self):
        betas = (hparams["optimizer_adam_beta1"], hparams["optimizer_adam_beta2"])
        fused = all(p.is_cuda for p in self.gen_params)
        opt_gen = torch.optim.AdamW(self.gen_params, lr=hparams["lr"], betas=betas, weight_decay=hparams["weight_decay"],
                                    fused=fused)
        opt_disc = torch.optim.AdamW(self.disc_params, lr=hparams["disc_lr"], betas=betas, fused=fused,
                                     **hparams["discriminator_optimizer_params"]) if len(self.disc_params) > 0 else None
        opt_map = torch.optim.AdamW(self.mapping_params, lr=hparams["map_lr"], betas=betas,
                                    weight_decay=hparams["weight_decay"], fused=fused)
        gen_sched = RSQRTSchedule(opt_gen, hparams) if hparams["scheduler"] == "rsqrt" else NoneSchedule(opt_gen, hparams)
        self.scheduler = {
            "gen": gen_sched,
            "disc": torch.optim.lr_scheduler.StepLR(optimizer=opt_disc, **hparams["discriminator_scheduler_params"])
            if opt_disc is not None else None,
            "map": torch.optim.lr_scheduler.StepLR(optimizer=opt_map, **hparams["map_scheduler_params"]),
        }
        return [opt_gen, opt_disc, opt_map]

    # ------------------------------------------------------------------ losses (tasks/tts/fs2.py:143-175)
    @staticmethod
    def weights_nonzero_speech(target):
        return target.abs().sum(-1, keepdim=True).ne(0).float().expand(-1, -1, target.size(-1))

    def _speech_weights(self, target):
        """(weights_nonzero_speech(target), its sum) -- the L1 and the SSIM term of a way weigh the same target, and the a2p way
        shares the professional target with p2p: computed once per target and step instead of once per term."""
        key = (target.data_ptr(), tuple(target.shape), target._version)
        hit = self._w_cache.get(key)
        if hit is None:
            w = self.weights_nonzero_speech(target)
            hit = self._w_cache[key] = (w, w.sum(), target)      # (holding `target` keeps its address from being reused)
        return hit[:2]

    def l1_loss(self, out, target):
        w, wsum = self._speech_weights(target)
        return ((out - target).abs() * w).sum() / wsum

    def ssim_loss(self, out, target, bias=6.0):
        w, wsum = self._speech_weights(target)
        s = 1 - SF.ssim_map(out, target, bias)
        return (s * w).sum() / wsum

    def add_mel_loss(self, mel_out, target, losses, postfix=""):
        for name, lbd in self.loss_and_lambda.items():
            if name == "l1":
                l = self.l1_loss(mel_out, target)
            elif name == "ssim":
                l = self.ssim_loss(mel_out, target)
            else:
                raise NotImplementedError(name)
            losses[f"{name}{postfix}"] = l * lbd

    def add_mel_terms(self, mel_out, target, terms, postfix=""):
        """add_mel_loss into a _Terms: L1 and SSIM of a way come from ONE fused pass (SF.mel_loss) when those are the
        configured terms (mel_loss: "ssim:0.5|l1:0.5"); anything else takes the per-term path."""
        names = list(self.loss_and_lambda)
        if hparams.get("fused_mel_loss", True) and names and all(n in ("l1", "ssim") for n in names):
            v = SF.mel_loss(mel_out, target, 6.0, l1="l1" in names, ssim="ssim" in names)
            lam = self.loss_and_lambda
            terms.add(v, [f"l1{postfix}" if "l1" in lam else None, f"ssim{postfix}" if "ssim" in lam else None, None],
                      [lam.get("l1", 0.0), lam.get("ssim", 0.0), 0.0], [1.0, 1.0, 0.0])
            return
        losses = {}
        self.add_mel_loss(mel_out, target, losses, postfix)
        for k, v in losses.items():
            terms.add(v, [k], [1.0])

    @staticmethod
    def get_corresponding_gtmel(way, sample):
        return sample["mels"] if way in ("a2a", "p2a") else sample["prof_mels"]

    # ------------------------------------------------------------------ host randoms of a step
    def _disc_plan(self, global_step):
        """[(optimizer_idx, way), ...] of the critic calls one training step makes, in execution order."""
        disc_start = hparams["mel_gan"] and global_step > hparams["disc_start_steps"] and hparams["lambda_mel_adv"] > 0
        phase, ways = self.phase_of(global_step)
        plan = []
        if phase in (1, 2) and disc_start:
            plan += [(0, w) for w in ways]
            if global_step % hparams["disc_interval"] == 0:
                plan += [(1, w) for w in ways for _ in ("real", "fake")]
        elif phase == 3 and not hparams["cross_way_no_disc_loss"]:
            plan += [(2, w) for w in ways]
        return phase, plan

    def begin_step(self, sample, global_step, stage_device=False):
        """Draw the step's host-side randoms up front, in the reference's order: the speaker-embedding pick of run_model
        (svb_vae_task.py:124), then per critic call one window start per window length (multi_window_disc.py:144-148,
        high = longest - window + 1).  `longest` comes from the batch's host-side lengths, so neither the draw nor the
        critic needs a device->host sync.  With stage_device the values are also written to static device tensors
        (read by the captured hipGraphs)."""
        phase, plan = self._disc_plan(global_step)
        runs_model = phase in (1, 2) or phase == 3
        rand = {"spk": int(np.random.randint(1, sample["multi_spk_emb"].shape[1])) if runs_model else 0, "disc": [],
                "cursor": 0, "structure": []}
        wins = self.mel_disc.time_lengths if hparams["mel_gan"] else []
        for _, way in plan:
            lens = sample["mel_lengths"] if way in ("a2a", "p2a") else sample["prof_mel_lengths"]
            longest = int(lens.max())
            starts = [[int(np.random.randint(low=0, high=longest - wl + 1))] * sample["mels"].shape[0] if longest - wl >= 0
                      else None for wl in wins]
            rand["disc"].append({"starts": starts, "longest": longest})
            rand["structure"].append(tuple(s is not None for s in starts))
        self._step_rand = rand
        if stage_device:
            # static device tensors (a captured hipGraph has their addresses baked in), refilled by ONE async copy from a
            # ring of pinned host rows: a pageable source would make the copy wait for the stream -- a full device sync
            # per step (round 4: the host needed 16 ms per "graph" step for exactly that reason)
            dev = self.model.z_mapping_function.parameters().__next__().device
            n_calls, n_win = 8, max(len(wins), 1)
            n = n_calls * n_win
            if self._rand_dev is None:
                flat = torch.zeros(n + 1, dtype=torch.long, device=dev)
                self._rand_dev = {"flat": flat, "starts": flat[:n].view(n_calls, n_win), "spk": flat[n:n + 1]}
                if dev.type == "cuda":
                    self._rand_ring = (torch.zeros(32, n + 1, dtype=torch.long).pin_memory(), [None] * 32, [0])
            if dev.type == "cuda":
                ring, events, cursor = self._rand_ring
                slot = cursor[0] % ring.shape[0]
                cursor[0] += 1
                if events[slot] is not None:
                    events[slot].synchronize()        # (only ever waits when the host is 32 steps ahead of the device)
                host = ring[slot]
            else:
                host = torch.zeros(n + 1, dtype=torch.long)
            host.zero_()
            for i, d in enumerate(rand["disc"][:n_calls]):
                for w, s in enumerate(d["starts"]):
                    host[i * n_win + w] = s[0] if s is not None else 0
            host[n] = rand["spk"]
            self._rand_dev["flat"].copy_(host, non_blocking=True)
            if dev.type == "cuda":
                ev = torch.cuda.Event()
                ev.record()
                events[slot] = ev
            rand["dev"] = True
        return rand

    def end_step(self):
        self._step_rand = None

    def prefetch(self, sample):
        """Called by the Trainer with the NEXT step's batch (already on the device) while the current step is between its
        forward and its backward: the frozen PPG encoder of that batch runs now, beside this step's backward."""
        if "mels" in sample and "prof_mels" in sample:
            self.model.prefetch_content(sample["mels"], sample["prof_mels"])

    def graph_key(self, global_step):
        """Everything host-side that shapes the launch sequence of a step (a captured graph is only replayed for an equal key)."""
        phase, plan = self._disc_plan(global_step)
        r = self._step_rand
        return (phase, tuple(plan), tuple(r["structure"]) if r else None)

    independent_critic_pass = 1     # optimizer index of the pass the Trainer may run on its own stream (see critic_barrier)
    critic_barrier = None           # set by the Trainer: makes the current stream wait for a critic pass still in flight

    def _critic_many(self, xs):
        """The critic on several mel batches at once (one stacked pass per tower) when the step's window starts were drawn
        up front and every window fits; otherwise one call after the other.  -> list of y."""
        if self.critic_barrier is not None:
            self.critic_barrier()
        r = self._step_rand
        n = len(xs)
        # (stacking is only an identity for per-clip layers: with disc_norm 'bn' the BatchNorm2d statistics would mix the
        # real and generated batches, so those configurations keep the reference's separate calls)
        if (hparams.get("stack_critic_calls", True) and self.mel_disc.norm_type == "in" and r is not None
                and r["cursor"] + n <= len(r["disc"]) and n > 1
                and all(x.shape == xs[0].shape for x in xs)
                and all(all(s is not None for s in d["starts"]) for d in r["disc"][r["cursor"]:r["cursor"] + n])):
            calls = []
            for x in xs:
                d = r["disc"][r["cursor"]]
                dev = self._rand_dev["starts"][r["cursor"]] if r.get("dev") else None
                r["cursor"] += 1
                calls.append((x, d["starts"], dev, d["longest"]))
            outs = self.mel_disc.forward_many(calls, want_fmaps=False)
            self._y_stacked = outs[0].get("y_all")       # [n*B,1,W]: the calls' scores in one tensor (see _adv_terms)
            return [o["y"] for o in outs]
        self._y_stacked = None
        return [self._critic(x)["y"] for x in xs]

    def _adv_terms(self, ys, keys, targets, weights, terms):
        """LS-GAN terms mse(y_i, target_i) of several critic calls (svb_para.py:118-170).  When the calls ran as one stacked
        pass their scores are one tensor and the terms one [n]-vector: 3 launches instead of 2 per term (ones_like + mse)."""
        keep = [i for i, y in enumerate(ys) if y is not None]
        if not keep:
            return
        ya = getattr(self, "_y_stacked", None)
        if ya is not None and len(keep) == len(ys) and ya.shape[0] == len(ys) * ys[0].shape[0]:
            n = len(ys)
            tv = terms._const(targets, ya.device)
            terms.add((ya.reshape(n, -1) - tv[:, None]).square().mean(1), keys, [1.0] * n, weights)
            return
        for i in keep:
            y = ys[i]
            t = torch.ones_like(y) if targets[i] == 1.0 else torch.zeros_like(y)
            terms.add(self.mse_loss_fn(y, t), [keys[i]], [1.0], [weights[i]])

    def _critic(self, x):
        if self.critic_barrier is not None:
            self.critic_barrier()
        r = self._step_rand
        if r is None or r["cursor"] >= len(r["disc"]):
            return self.mel_disc(x, None, want_fmaps=False)
        d = r["disc"][r["cursor"]]
        dev = self._rand_dev["starts"][r["cursor"]] if r.get("dev") else None
        r["cursor"] += 1
        return self.mel_disc(x, None, start_frames_wins=d["starts"], starts_dev=dev, longest=d["longest"], want_fmaps=False)

    # ------------------------------------------------------------------ model run (svb_vae_task.py:120-165)
    def run_model(self, model, sample, concurrent_ways, return_output=False, infer=False, disable_map=False, terms=None,
                  **inject):
        if model.vc_asr.training:          # (pinned to eval by VCASR.train(); walking its ~120 submodules every step cost 0.3 ms)
            model.vc_asr.eval()
        r = self._step_rand
        if infer:
            spk = sample["multi_spk_emb"][:, 0, :]
        elif r is not None and r.get("dev"):
            spk = sample["multi_spk_emb"].index_select(1, self._rand_dev["spk"])[:, 0, :]
        else:
            idx = r["spk"] if r is not None else np.random.randint(1, sample["multi_spk_emb"].shape[1])
            spk = sample["multi_spk_emb"][:, idx, :]
        output = model(amateur_mel=sample["mels"], prof_mel=sample["prof_mels"], amateur_pitch=sample["pitch"],
                       prof_pitch=sample["prof_pitch"], amateur_spk_id=spk, prof_spk_id=spk,
                       a2p_alignment=sample["a2p_f0_alignment"], p2a_alignment=None, infer=False,
                       concurrent_ways=concurrent_ways, disable_map=disable_map, **inject)
        if terms is not None:            # training: the terms go into a few vectors (see _Terms)
            kl_vec = getattr(model, "_last_stacked_kl", None)
            kl_ways = [w for w in concurrent_ways if "kl" in output[w]]
            if kl_vec is not None and kl_ways == ["a2a", "p2p"]:
                terms.add(kl_vec, ["a2a_kl", "p2p_kl"], [hparams["lambda_kl"]] * 2, guard=True)
            else:
                for way in kl_ways:
                    terms.add(output[way]["kl"], [f"{way}_kl"], [hparams["lambda_kl"]], guard=True)
            for way in concurrent_ways:
                if way not in ("a2a", "p2p") and hparams["cross_way_no_recon_loss"]:
                    continue
                self.add_mel_terms(output[way]["mel_out"], self.get_corresponding_gtmel(way, sample), terms, postfix=way)
            return (terms, output) if return_output else terms
        losses = {}
        for way in concurrent_ways:
            if "kl" in output[way]:
                losses[f"{way}_kl"] = output[way]["kl"] * hparams["lambda_kl"]
            if way not in ("a2a", "p2p") and hparams["cross_way_no_recon_loss"]:
                continue
            self.add_mel_loss(output[way]["mel_out"], self.get_corresponding_gtmel(way, sample), losses, postfix=way)
        return (losses, output) if return_output else losses

    # ------------------------------------------------------------------ GAN helpers (svb_para.py:118-170)
    def gen_cheat_disc(self, way, model_out, log_outputs, loss_weights):
        p_ = self._critic(model_out[way]["mel_out"])["y"]
        if p_ is not None:
            log_outputs[f"{way}_a"] = self.mse_loss_fn(p_, torch.ones_like(p_))
            loss_weights[f"{way}_a"] = hparams["lambda_mel_adv"]

    def disc_judge_gen(self, way, sample, log_outputs):
        p = self._critic(self.get_corresponding_gtmel(way, sample))["y"]
        p_ = self._critic(self.model_out_gt[way]["mel_out"])["y"]
        if p_ is not None:
            log_outputs[f"{way}_r"] = self.mse_loss_fn(p, torch.ones_like(p))
            log_outputs[f"{way}_f"] = self.mse_loss_fn(p_, torch.zeros_like(p_))

    def phase_of(self, step):
        if step <= hparams["phase_1_steps"]:
            return 1, hparams["phase_1_concurrent_ways"].split(",")
        if step <= hparams["phase_2_steps"]:
            return 2, hparams["phase_2_concurrent_ways"].split(",")
        return 3, hparams["phase_3_concurrent_ways"].split(",")

    # ------------------------------------------------------------------ the step (svb_vae_task.py:579-676)
    def _training_step(self, sample, batch_idx, optimizer_idx):
        self._w_cache = {}
        terms = _Terms()
        disc_start = hparams["mel_gan"] and self.global_step > hparams["disc_start_steps"] and hparams["lambda_mel_adv"] > 0
        phase, ways = self.phase_of(self.global_step)
        lam_adv = hparams["lambda_mel_adv"]
        if optimizer_idx == 0:
            if phase in (1, 2):
                if self.model.z_mapping_function.training:
                    self.model.z_mapping_function.eval()
                _, model_out = self.run_model(self.model, sample, ways, return_output=True, terms=terms)
                self.model_out = {w: {k: v.detach() for k, v in o.items() if isinstance(v, torch.Tensor)}
                                  for w, o in model_out.items()}
                self.model_out_gt = self.model_out
                if self._step_rand is not None and self._step_rand.get("dev"):
                    # hipGraph mode: the critic pass (possibly another graph) reads the generated mels from fixed buffers
                    if not hasattr(self, "_gen_mel_buf"):
                        self._gen_mel_buf = {}
                    if self.critic_barrier is not None:      # the previous step's critic pass may still read these buffers
                        self.critic_barrier()
                    gt = {}
                    for w, o in self.model_out.items():
                        # one buffer per (way, shape), never freed: a captured graph has the address baked in, and the
                        # Trainer keeps graphs of several batch shapes alive at once
                        bkey = (w, tuple(o["mel_out"].shape))
                        buf = self._gen_mel_buf.get(bkey)
                        if buf is None:
                            buf = self._gen_mel_buf[bkey] = torch.empty_like(o["mel_out"])
                        buf.copy_(o["mel_out"])
                        gt[w] = {"mel_out": buf}
                    self.model_out_gt = gt
                if disc_start:       # gen_cheat_disc (svb_para.py:118-131), all ways in one critic pass
                    ys = self._critic_many([model_out[w]["mel_out"] for w in ways])
                    self._adv_terms(ys, [f"{w}_a" for w in ways], [1.0] * len(ways), [lam_adv] * len(ways), terms)
        elif optimizer_idx == 1:
            if phase in (1, 2):
                if self.model.z_mapping_function.training:
                    self.model.z_mapping_function.eval()
                if disc_start and self.global_step % hparams["disc_interval"] == 0:
                    xs = [x for way in ways for x in (self.get_corresponding_gtmel(way, sample),
                                                      self.model_out_gt[way]["mel_out"])]
                    if self.critic_barrier is not None and xs[0].is_cuda:
                        # (this pass may run on the Trainer's critic stream: its inputs were produced on the compute stream
                        #  and are dropped there when the next step replaces them)
                        for x in xs:
                            x.record_stream(torch.cuda.current_stream(x.device))
                    ys = self._critic_many(xs)   # disc_judge_gen (svb_para.py:133-170): real, fake per way, one pass
                    self._adv_terms(ys, [f"{w}_{rf}" for w in ways for rf in ("r", "f")], [1.0, 0.0] * len(ways),
                                    [1.0] * len(ys), terms)
        elif optimizer_idx == 2:
            if phase == 3:
                self.model.eval()
                self.model.z_mapping_function.train()
                _, model_out = self.run_model(self.model, sample, ["a2a", "p2p"] + ways, return_output=True, terms=terms)
                for way in ways:
                    cross = model_out[way]
                    terms.add(cross["mle"], [f"{way}_mle"], [1.0], [hparams["lambda_mle"]], guard=True)
                    if not hparams["cross_way_no_disc_loss"]:
                        self._y_stacked = None
                        self._adv_terms([self._critic(cross["mel_out"])["y"]], [f"{way}_a"], [1.0], [lam_adv], terms)
        if len(terms) == 0:
            return None
        total, log_outputs = terms.total_and_logs()      # total = sum_k weight_k * loss_k (:673-674)
        log_outputs["bs"] = sample["mels"].shape[0]
        return total, log_outputs

    def on_before_optimization(self, opt_idx):
        params, norm = ((self.gen_params, hparams["generator_grad_norm"]),
                        (self.disc_params, hparams["discriminator_grad_norm"]),
                        (self.mapping_params, hparams["generator_grad_norm"]))[opt_idx]
        flat = getattr(self.trainer, "flat_optim", None) if self.trainer is not None else None
        if flat and opt_idx < len(flat) and flat[opt_idx] is not None:
            flat[opt_idx].set_clip(norm)          # the clip factor is applied inside the flat AdamW launch (utils/flat_optim.py)
            return
        torch.nn.utils.clip_grad_norm_(params, norm, foreach=True)

    def on_after_optimization(self, epoch, batch_idx, optimizer, optimizer_idx):
        if optimizer_idx == 0:
            self.scheduler["gen"].step(self.global_step)
        elif optimizer_idx == 1:
            self.scheduler["disc"].step(max(self.global_step - hparams["disc_start_steps"], 1))
        elif optimizer_idx == 2:
            self.scheduler["map"].step(self.global_step)

    # ------------------------------------------------------------------ validation / test
    def validation_step(self, sample, batch_idx):
        self._w_cache = {}
        phase, _ = self.phase_of(self.global_step)
        ways = {1: ["p2p"], 2: ["a2a", "p2p"], 3: ["a2a", "p2p", "a2p"]}[phase]
        losses, model_out = self.run_model(self.model, sample, ways, return_output=True, infer=True,
                                           disable_map=hparams["disable_map"])
        for way in ways:
            if "mle" in model_out[way]:
                losses[f"{way}_mle"] = model_out[way]["mle"]
        out = {"losses": losses, "total_loss": sum(losses.values()), "nsamples": sample["nsamples"]}
        return {k: ({kk: float(vv) for kk, vv in v.items()} if isinstance(v, dict) else float(v)) for k, v in out.items()}

    def test_step(self, sample, batch_idx):
        from .infer import infer_and_save
        return infer_and_save(self, sample, batch_idx)

    def test_end(self, outputs):
        return {}

    # ------------------------------------------------------------------ data (tasks/tts/tts.py:57-101)
    @data_loader
    def train_dataloader(self):
        ds = self.dataset_cls(hparams["train_set_name"], True)
        return self.build_dataloader(ds, True, hparams["max_tokens"], hparams["max_sentences"],
                                     endless=hparams["endless_ds"])

    @data_loader
    def val_dataloader(self):
        ds = self.dataset_cls(hparams["valid_set_name"], False)
        return self.build_dataloader(ds, False, hparams["max_valid_tokens"], hparams["max_valid_sentences"])

    @data_loader
    def test_dataloader(self):
        ds = self.dataset_cls(hparams["test_set_name"], False)
        return self.build_dataloader(ds, False, hparams["max_valid_tokens"], hparams["max_valid_sentences"],
                                     batch_by_size=False)

"""TEST INFRASTRUCTURE ONLY -- CPU port of one vae_global_mle_eng training step (phase 2: generator + discriminator
passes) on stock torch fp32 ops, used as (a) the step-level parity oracle and (b) bench.py's `cpu_baseline` ("port").

Follows reference tasks/singing/svb_vae_task.py:579-676 (+ svb_para.py:118-170, fs2.py:143-175, trainer.py:269-342):
  pass 0: run_model(ways a2a,p2p) -> kl*lambda_kl + 0.5*ssim + 0.5*l1 per way (+ 0.1 * MSE(D(mel_out), 1) once
          disc_start), backward, clip 5.0, AdamW(gen);
  pass 1: MSE(D(mel_gt),1) + MSE(D(mel_out.detach()),0) per way, backward, clip 1.0, AdamW(disc).
All randomness (z noise, window starts, Dropout2d channel masks) is injected so the HIP step can be fed the same draws.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import modules_ref as R


class CpuStep:
    def __init__(self, model_sd, disc_sd, hp):
        self.hp = hp
        self.sd = {k: v.clone() for k, v in model_sd.items()}
        self.dsd = {k: v.clone() for k, v in disc_sd.items()}
        self.gen_keys = [k for k, v in self.sd.items() if v.is_floating_point() and not k.startswith("vc_asr")
                         and not k.startswith("z_mapping_function") and "running_" not in k]
        self.disc_keys = [k for k, v in self.dsd.items() if v.is_floating_point() and "running_" not in k]
        for k in self.gen_keys:
            self.sd[k].requires_grad_(True)
        for k in self.disc_keys:
            self.dsd[k].requires_grad_(True)
        betas = (hp["optimizer_adam_beta1"], hp["optimizer_adam_beta2"])
        self.opt_gen = torch.optim.AdamW([self.sd[k] for k in self.gen_keys], lr=hp["lr"], betas=betas,
                                         weight_decay=hp["weight_decay"])
        self.opt_disc = torch.optim.AdamW([self.dsd[k] for k in self.disc_keys], lr=hp["disc_lr"], betas=betas,
                                          **hp["discriminator_optimizer_params"])
        # RSQRTSchedule.__init__ calls step(0) (common_schedulers.py:33); the task re-steps it AFTER each optimizer
        # step with the current global_step (svb_vae_task.py:398-400), so step s runs with lr(s_prev).
        self.cur_gen_lr = self.gen_lr(0)

    def gen_lr(self, step):
        w, h = self.hp["warmup_updates"], self.hp["hidden_size"]
        return max(self.hp["lr"] * min(step / w, 1.0) * max(w, step) ** -0.5 * h ** -0.5, 1e-7)

    def step(self, sample, spk_idx, eps_a2a, eps_p2p, starts_gen, starts_disc, global_step, drop_masks=None):
        """starts_gen[way] / starts_disc[way] = {'real': [...], 'fake': [...]}: window starts per discriminator call."""
        hp = self.hp
        logs = {}
        spk = sample["multi_spk_emb"][:, spk_idx]
        for g in self.opt_gen.param_groups:
            g["lr"] = self.cur_gen_lr
        ret, _, _ = R.mle_svb_vae(self.sd, sample["mels"], sample["prof_mels"], sample["pitch"], sample["prof_pitch"], spk,
                                  sample["a2p_f0_alignment"], ["a2a", "p2p"], eps_a2a, eps_p2p, hp, training=True,
                                  map_training=False)
        total = 0.0
        disc_start = global_step > hp["disc_start_steps"]
        for way, tgt in (("a2a", sample["mels"]), ("p2p", sample["prof_mels"])):
            logs[f"{way}_kl"] = ret[way]["kl"] * hp["lambda_kl"]
            logs[f"ssim{way}"] = R.ssim_loss(ret[way]["mel_out"], tgt) * 0.5
            logs[f"l1{way}"] = R.l1_loss(ret[way]["mel_out"], tgt) * 0.5
            total = total + logs[f"{way}_kl"] + logs[f"ssim{way}"] + logs[f"l1{way}"]
            if disc_start:
                y, _ = R.mel_discriminator(self.dsd, ret[way]["mel_out"], starts_gen[way],
                                           drop_masks=None if drop_masks is None else drop_masks[("gen", way)])
                logs[f"{way}_a"] = F.mse_loss(y, torch.ones_like(y))
                total = total + hp["lambda_mel_adv"] * logs[f"{way}_a"]
        self.opt_gen.zero_grad()
        for k in self.disc_keys:
            self.dsd[k].grad = None
        total.backward()
        torch.nn.utils.clip_grad_norm_([self.sd[k] for k in self.gen_keys], hp["generator_grad_norm"])
        self.opt_gen.step()
        self.cur_gen_lr = self.gen_lr(global_step)
        logs["total_gen"] = total.detach()
        if disc_start:
            dl = 0.0
            for way, tgt in (("a2a", sample["mels"]), ("p2p", sample["prof_mels"])):
                p, _ = R.mel_discriminator(self.dsd, tgt, starts_disc[way]["real"],
                                           drop_masks=None if drop_masks is None else drop_masks[("real", way)])
                p_, _ = R.mel_discriminator(self.dsd, ret[way]["mel_out"].detach(), starts_disc[way]["fake"],
                                            drop_masks=None if drop_masks is None else drop_masks[("fake", way)])
                logs[f"{way}_r"] = F.mse_loss(p, torch.ones_like(p))
                logs[f"{way}_f"] = F.mse_loss(p_, torch.zeros_like(p_))
                dl = dl + logs[f"{way}_r"] + logs[f"{way}_f"]
            for k in self.disc_keys:
                self.dsd[k].grad = None
            dl.backward()
            torch.nn.utils.clip_grad_norm_([self.dsd[k] for k in self.disc_keys], hp["discriminator_grad_norm"])
            self.opt_disc.step()
            logs["total_disc"] = dl.detach()
        return {k: float(v.detach()) for k, v in logs.items()}

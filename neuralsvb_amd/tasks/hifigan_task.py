"""HifiGanTask -- NSF-HifiGAN vocoder training (generator + MPD + MSD), BASELINE config #3.

The reference names `tasks.vocoder.hifigan.HifiGanTask` in egs/egs_bases/tts/vocoder/hifigan.yaml:2 but ships no
`tasks/vocoder/`: this task is *composed* from the reference's modules and loss functions
(modules/hifigan/hifigan.py:328-365, mel_utils.py:45-79) and that YAML's hyper-parameters with the upstream-standard
two-optimizer GAN step, and is therefore parity-unpinned at task level (its pieces are pinned individually).

  opt 0 (generator):      y_ = G(mel, f0);  lambda_mel * L1(mel(y), mel(y_)) + lambda_adv * [gen_loss(MPD) + gen_loss(MSD)]
                          (+ feature matching when use_fm_loss) -- active adversarial terms after disc_start_steps
  opt 1 (discriminators): disc_loss(MPD(y, y_.detach())) + disc_loss(MSD(y, y_.detach()))
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..modules.frontend import MelFrontend
from ..modules.stft_loss import MultiResolutionSTFTLoss
from ..modules.hifigan import (HifiGanGenerator, MultiPeriodDiscriminator, MultiScaleDiscriminator, discriminator_loss,
                               feature_loss, generator_loss)
from ..utils.hparams import hparams
from .base_task import BaseTask, data_loader


def default_hparams():
    """egs/egs_bases/tts/vocoder/hifigan.yaml:1-37 with the hop-128 NSF generator of SURVEY Appendix D (the released
    1012_hifigan_all_songs_nsf/config.yaml is not in the reference tree) -- what `egs/.../hifigan_nsf.yaml` resolves to."""
    return dict(resblock="1", upsample_rates=[8, 4, 2, 2], upsample_kernel_sizes=[16, 8, 4, 4], upsample_initial_channel=512,
                resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, use_pitch_embed=True,
                audio_sample_rate=24000, hop_size=128, fft_size=512, win_size=512, audio_num_mel_bins=80, fmin=50, fmax=12000,
                adam_b1=0.8, adam_b2=0.99, use_fm_loss=False, use_ms_stft=False, lambda_mel=5.0, lambda_adv=1.0,
                disc_start_steps=0, generator_grad_norm=10, discriminator_grad_norm=1,
                generator_optimizer_params={"lr": 2e-4}, generator_scheduler_params={"step_size": 600, "gamma": 0.999},
                discriminator_optimizer_params={"lr": 2e-4}, discriminator_scheduler_params={"step_size": 600, "gamma": 0.999})


class HifiGanTask(BaseTask):
    def __init__(self):
        super().__init__()
        from .vocoder_dataset import VocoderDataset
        self.dataset_cls = VocoderDataset
        self._fe = None
        self._inject = None          # tests: (rand_ini, noise) of the NSF source, so a CPU oracle can be fed the same draws
        self.stft_loss = MultiResolutionSTFTLoss() if hparams.get("use_ms_stft", False) else None

    def build_model(self):
        from .. import functional as SF
        SF.set_precision(hparams.get("conv_precision", "fp32"))
        from ..modules import hifigan as _hg
        _hg.FUSED_SPECTRAL_NORM = bool(hparams.get("fused_spectral_norm", True))
        _hg.FUSED_RESBLOCK = bool(hparams.get("fused_resblock", True))
        self.model_gen = HifiGanGenerator(hparams)
        self.model_disc = nn.ModuleDict()
        self.model_disc["mpd"] = MultiPeriodDiscriminator()
        self.model_disc["msd"] = MultiScaleDiscriminator()
        self.gen_params = list(self.model_gen.parameters())
        self.disc_params = list(self.model_disc.parameters())
        return self.model_gen

    def configure_optimizers(self):
        betas = (hparams["adam_b1"], hparams["adam_b2"])
        fused = all(p.is_cuda for p in self.gen_params)
        og = torch.optim.AdamW(self.gen_params, betas=betas, fused=fused, **hparams["generator_optimizer_params"])
        od = torch.optim.AdamW(self.disc_params, betas=betas, fused=fused, **hparams["discriminator_optimizer_params"])
        self.scheduler = {"gen": torch.optim.lr_scheduler.StepLR(og, **hparams["generator_scheduler_params"]),
                          "disc": torch.optim.lr_scheduler.StepLR(od, **hparams["discriminator_scheduler_params"])}
        return [og, od]

    def mel(self, y):
        if self._fe is None or self._fe.device != y.device:
            self._fe = MelFrontend(hparams, y.device)
        return self._fe.mel_spectrogram(y)

    def _training_step(self, sample, batch_idx, optimizer_idx):
        mel, y, f0 = sample["mels"], sample["wavs"], sample.get("f0")
        logs = {}
        adv = self.global_step >= hparams["disc_start_steps"]
        if optimizer_idx == 0:
            inj = {} if self._inject is None else {"rand_ini": self._inject[0], "noise": self._inject[1]}
            y_ = self.model_gen(mel, f0, **inj)
            logs["mel"] = F.l1_loss(self.mel(y_[:, 0]), self.mel(y[:, 0]).detach()) * hparams["lambda_mel"]
            if self.stft_loss is not None:        # optional multi-resolution STFT terms (stft_loss.py:79-152)
                sc, mag = self.stft_loss(y_[:, 0], y[:, 0])
                logs["sc"], logs["mag"] = sc, mag
            self.y_ = y_.detach()
            if adv:
                for name in ("mpd", "msd"):
                    _, y_g, fr, fg = self.model_disc[name](y, y_)
                    logs[f"a_{name}"] = generator_loss(y_g) * hparams["lambda_adv"]
                    if hparams["use_fm_loss"]:
                        logs[f"fm_{name}"] = feature_loss(fr, fg)
        else:
            if not adv:
                return None
            for name in ("mpd", "msd"):
                y_r, y_g, _, _ = self.model_disc[name](y, self.y_)
                r, g = discriminator_loss(y_r, y_g)
                logs[f"r_{name}"], logs[f"f_{name}"] = r, g
        return sum(logs.values()), logs

    def on_before_optimization(self, opt_idx):
        params, norm = ((self.gen_params, hparams["generator_grad_norm"]),
                        (self.disc_params, hparams["discriminator_grad_norm"]))[opt_idx]
        flat = getattr(self.trainer, "flat_optim", None) if self.trainer is not None else None
        if flat and opt_idx < len(flat) and flat[opt_idx] is not None:
            flat[opt_idx].set_clip(norm)          # applied inside the flat AdamW launch (utils/flat_optim.py)
            return
        torch.nn.utils.clip_grad_norm_(params, norm, foreach=True)

    def on_after_optimization(self, epoch, batch_idx, optimizer, optimizer_idx):
        self.scheduler["gen" if optimizer_idx == 0 else "disc"].step()

    # ------------------------------------------------------------------ validation / data
    def validation_step(self, sample, batch_idx):
        with torch.no_grad():
            y_ = self.model_gen(sample["mels"], sample.get("f0"))
            n = min(y_.shape[-1], sample["wavs"].shape[-1])
            mel_l1 = F.l1_loss(self.mel(y_[:, 0, :n]), self.mel(sample["wavs"][:, 0, :n]))
        return {"losses": {"mel": float(mel_l1)}, "total_loss": float(mel_l1), "nsamples": sample["nsamples"]}

    @data_loader
    def train_dataloader(self):
        return self.build_dataloader(self.dataset_cls("train", True), True, max_sentences=hparams["max_sentences"],
                                     endless=hparams.get("endless_ds", False), batch_by_size=False)

    @data_loader
    def val_dataloader(self):
        return self.build_dataloader(self.dataset_cls("valid", False), False, max_sentences=hparams.get("max_valid_sentences", 1),
                                     batch_by_size=False)

    @data_loader
    def test_dataloader(self):
        return self.build_dataloader(self.dataset_cls("test", False), False, max_sentences=hparams.get("max_valid_sentences", 1),
                                     batch_by_size=False)

"""BASELINE configs #3 and #5 (GPU only, secondary numbers; bench.py is the headline metric).

  python tools/bench_vocoder.py train   # NSF-HifiGAN vocoder-only training step (G+MPD+MSD), 8192-sample segments, B=64
  python tools/bench_vocoder.py infer   # end-to-end inference: PPG+pitch -> VAE mel (a2a,p2p,a2p) -> NSF-HifiGAN, 32 x 10 s, RTF
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HIFI = dict(resblock="1", upsample_rates=[8, 4, 2, 2], upsample_kernel_sizes=[16, 8, 4, 4], upsample_initial_channel=512,
            resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, use_pitch_embed=True,
            audio_sample_rate=24000, hop_size=128, fft_size=512, win_size=512, audio_num_mel_bins=80, fmin=50, fmax=12000,
            adam_b1=0.8, adam_b2=0.99, use_fm_loss=False, lambda_mel=5.0, lambda_adv=1.0, disc_start_steps=0,
            generator_grad_norm=10, discriminator_grad_norm=1,
            generator_optimizer_params={"lr": 2e-4}, generator_scheduler_params={"step_size": 600, "gamma": 0.999},
            discriminator_optimizer_params={"lr": 2e-4}, discriminator_scheduler_params={"step_size": 600, "gamma": 0.999})


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def train(B=64, L=8192, steps=10, warmup=3):
    from neuralsvb_amd.utils.hparams import hparams
    from neuralsvb_amd.tasks.hifigan_task import HifiGanTask
    from neuralsvb_amd.utils.trainer import Trainer
    hparams.clear(); hparams.update(HIFI)
    dev = torch.device("cuda:0")
    trainer = Trainer(work_dir="", num_sanity_val_steps=0)
    torch.manual_seed(0)
    task = trainer.setup(HifiGanTask()); task.train()
    g = torch.Generator().manual_seed(0)
    t = torch.arange(L) / 24000.0
    f0 = 150 + 200 * torch.rand(B, L // 128, generator=g)
    f0[:, ::9] = 0.0
    wav = 0.5 * torch.sin(2 * np.pi * 220 * t)[None].repeat(B, 1) + 0.05 * torch.randn(B, L, generator=g)
    from neuralsvb_amd.modules.frontend import MelFrontend
    mel = MelFrontend(hparams, dev).mel_spectrogram(wav.to(dev))
    batch = {"mels": mel, "wavs": wav[:, None].to(dev), "f0": f0.to(dev)}
    def step():
        trainer.run_training_batch(0, batch)
    s = timed(step, warmup, steps)
    print(json.dumps({"metric": "vocoder train step (G+MPD+MSD), config #3", "ms_per_step": s * 1e3,
                      "audio_seconds_per_sec": B * L / 24000.0 / s, "batch": B, "segment": L,
                      "dtype": __import__("neuralsvb_amd.functional", fromlist=["PRECISION"]).PRECISION}))


def infer(B=32, seconds=10.0, steps=3, warmup=1):
    from neuralsvb_amd.utils.hparams import set_hparams, hparams
    from neuralsvb_amd.modules.svb_vae import MleSVBVAE
    from neuralsvb_amd.modules.hifigan import HifiGanGenerator
    set_hparams(config=os.path.join(ROOT, "egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml"), exp_name="",
                hparams_str="audio_sample_rate=24000,fmax=12000", print_hparams=False)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = MleSVBVAE(70, hparams).to(dev).eval()
    gen = HifiGanGenerator(HIFI)
    gen.remove_weight_norm()
    gen = gen.to(dev).eval()
    T = int(seconds * 24000) // 128 // 4 * 4
    g = torch.Generator().manual_seed(0)
    mels = (torch.randn(B, T, 80, generator=g) * 0.8 - 3).to(dev)
    pitch = torch.randint(1, 255, (B, T), generator=g).to(dev)
    spk = (torch.randn(B, 256, generator=g) / 16).to(dev)
    al = torch.arange(T)[None].repeat(B, 1).to(dev)
    f0 = (150 + 200 * torch.rand(B, T, generator=g)).to(dev)
    @torch.no_grad()
    def run():
        out = model(amateur_mel=mels, prof_mel=mels, amateur_pitch=pitch, prof_pitch=pitch, amateur_spk_id=spk,
                    prof_spk_id=spk, a2p_alignment=al, concurrent_ways=["a2a", "p2p", "a2p"])
        for w in ("a2a", "p2p", "a2p"):          # + the two ground-truth resyntheses the reference writes
            gen(out[w]["mel_out"].transpose(1, 2).contiguous(), f0)
        gen(mels.transpose(1, 2).contiguous(), f0)
        gen(mels.transpose(1, 2).contiguous(), f0)
    s = timed(run, warmup, steps)
    audio = B * T * 128 / 24000.0
    print(json.dumps({"metric": "end-to-end inference (VAE 3 ways + 5 vocoder passes), config #5", "s_per_batch": s,
                      "rtf": s / audio, "audio_seconds_per_sec": audio / s, "batch": B, "clip_seconds": T * 128 / 24000.0,
                      "dtype": __import__("neuralsvb_amd.functional", fromlist=["PRECISION"]).PRECISION}))


if __name__ == "__main__":
    from neuralsvb_amd import functional as SF
    PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
    SF.set_precision(PREC)
    {"train": train, "infer": infer}[sys.argv[1] if len(sys.argv) > 1 else "train"]()

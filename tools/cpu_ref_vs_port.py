"""CPU only (build container): time the UNMODIFIED reference's phase-2 training step (its own Trainer.run_training_batch via
tasks/run.py, B = 16 x 6 s synthetic clips) next to the oracle's CPU port of the same step (oracle/train_step_ref.py,
bench.py's `cpu_baseline` "port") on the same cores.  Shows that the port is a fair stand-in for the reference's CPU path
on boxes where /root/reference is absent (the GPU box).   python tools/cpu_ref_vs_port.py [threads]"""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
THREADS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, SECONDS = 16, 6.0


def main():
    torch.set_num_threads(THREADS)
    import step_common as C
    from oracle import ref_shims
    ref_shims.install()
    tmp = tempfile.mkdtemp(prefix="cpuref_")
    os.symlink(os.path.join(ref_shims.REFERENCE_ROOT, "egs"), os.path.join(tmp, "egs"))
    os.chdir(tmp)
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    os.environ["NUM_WORKERS"] = "0"
    hp_str = ("audio_sample_rate=24000,fmax=12000,num_sanity_val_steps=0,max_updates=6,max_sentences=16,max_tokens=100000,"
              "ds_workers=0,val_check_interval=100000,tb_log_interval=1000,endless_ds=False")
    sys.argv = ["tasks/run.py", "--config", "egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml", "--exp_name", "cpuref",
                "--reset", "--hparams", hp_str]
    from utils.hparams import set_hparams, hparams
    set_hparams(print_hparams=False)
    from neuralsvb_amd.utils import synth
    synth.write_binary_dataset(hparams["binary_data_dir"], hparams, C.oracle_mel_fn(hparams), n_train=B, n_valid=1, seconds=SECONDS)
    synth.write_fake_asr_ckpt(hparams["pretrain_asr_ckpt"], 70, hparams)
    import utils.trainer as rtrainer
    from tasks.run import run_task
    times, keep = [], {}
    orig = rtrainer.Trainer.run_training_batch

    def timed(self, batch_idx, batch):
        keep.setdefault("batch", batch)
        keep["task"] = self.get_task_ref()
        t0 = time.perf_counter()
        r = orig(self, batch_idx, batch)
        times.append((self.global_step, time.perf_counter() - t0))
        return r
    rtrainer.Trainer.run_training_batch = timed
    run_task()
    ref_steps = [t for s, t in times if s >= 1]            # global_step 0 has no critic pass (disc_start is step > 0)
    ref_t = float(np.median(ref_steps))
    # ---- the port, same weights / batch / cores
    from oracle.train_step_ref import CpuStep
    task, batch = keep["task"], keep["batch"]
    step = CpuStep({k: v.detach().clone() for k, v in task.model.state_dict().items()},
                   {k: v.detach().clone() for k, v in task.mel_disc.state_dict().items()}, dict(hparams))
    g = torch.Generator().manual_seed(0)
    starts = {w: [[5] * B, [9] * B, [3] * B] for w in ("a2a", "p2p")}
    sd = {w: {"real": [[5] * B, [9] * B, [3] * B], "fake": [[7] * B, [2] * B, [11] * B]} for w in ("a2a", "p2p")}
    pt = []
    for i in range(3):
        eps = [torch.randn(B, hparams["latent_size"], 1, generator=g) for _ in range(2)]
        t0 = time.perf_counter()
        step.step(batch, 1, eps[0], eps[1], starts, sd, global_step=1 + i)
        pt.append(time.perf_counter() - t0)
    port_t = float(np.median(pt[1:]))
    os.chdir(ROOT)
    shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps({"threads": THREADS, "batch": B, "clip_seconds": SECONDS, "frames": int(batch["mels"].shape[1]),
                      "reference_s_per_step": ref_t, "reference_audio_s_per_s": B * SECONDS / ref_t,
                      "reference_steps_timed": len(ref_steps), "port_s_per_step": port_t,
                      "port_audio_s_per_s": B * SECONDS / port_t, "port_over_reference_time": port_t / ref_t}))


if __name__ == "__main__":
    main()

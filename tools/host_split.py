"""Where the HOST time of a bench step goes (GPU only): wall-clock of the Python side of each optimizer pass split into forward
(task.training_step), backward (loss.backward: the autograd engine's thread runs the Python backward functions), and the rest
(gradient exchange, clipping, optimizer step, repack, zeroing).  Asynchronous launches: these are issue times, not GPU times."""
import argparse
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--extra-hparams", default="")
a = ap.parse_args()
args = argparse.Namespace(gpus=1, steps=1, warmup=1, batch=16, seconds=6.0, sample_rate=24000, precision="bf16x3", graph=False)
dev = torch.device("cuda:0")
acc = {}


def add(k, dt):
    acc[k] = acc.get(k, 0.0) + dt


with tempfile.TemporaryDirectory() as tmp:
    task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp, ("," + a.extra_hparams) if a.extra_hparams else "")
    bench.run_steps(trainer, task, batch, 8, 1)
    torch.cuda.synchronize()
    o_ts = task.training_step
    o_bw = torch.Tensor.backward
    o_pass = trainer._optimizer_pass
    cur = {"opt": None}

    def ts(sample, batch_idx, opt_idx):
        t0 = time.perf_counter()
        r = o_ts(sample, batch_idx, opt_idx)
        add(f"opt{opt_idx}.forward", time.perf_counter() - t0)
        return r

    def bw(self, *x, **k):
        t0 = time.perf_counter()
        r = o_bw(self, *x, **k)
        add(f"opt{cur['opt']}.backward", time.perf_counter() - t0)
        return r

    def opass(task_, batch_, batch_idx, opt_idx, *rest):
        cur["opt"] = opt_idx
        t0 = time.perf_counter()
        r = o_pass(task_, batch_, batch_idx, opt_idx, *rest)
        add(f"opt{opt_idx}.pass_total", time.perf_counter() - t0)
        return r
    task.training_step, torch.Tensor.backward, trainer._optimizer_pass = ts, bw, opass
    t0 = time.perf_counter()
    bench.run_steps(trainer, task, batch, a.steps, 9)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
print(f"host issue {t_host / a.steps * 1e3:.2f} ms/step, wall {t_all / a.steps * 1e3:.2f} ms/step")
for k in sorted(acc):
    print(f"  {k:22s} {acc[k] / a.steps * 1e3:7.2f} ms/step")
for o in (0, 1, 2):
    if f"opt{o}.pass_total" in acc:
        rest = acc[f"opt{o}.pass_total"] - acc.get(f"opt{o}.forward", 0) - acc.get(f"opt{o}.backward", 0)
        print(f"  opt{o}.rest (exchange/clip/step/repack/zero) {rest / a.steps * 1e3:7.2f} ms/step")

"""Host time per autograd node class of the bench step (MI355X box): every custom Function's forward / backward in
neuralsvb_amd.functional is wrapped with a perf_counter pair (the backward runs on autograd's device thread, which cProfile
does not see), 10 steps, per-class calls and microseconds per step; plus the wall time of loss.backward() as the main thread sees it.
  python tools/node_profile.py"""
import argparse
import collections
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuralsvb_amd import functional as SF  # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(cls, name):
    fn = getattr(cls, name)

    def timed(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[(cls.__name__, name)]
            e[0] += 1
            e[1] += time.perf_counter() - t
    setattr(cls, name, staticmethod(timed))


for v in list(vars(SF).values()):
    if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
        wrap(v, "forward")
        wrap(v, "backward")

args = argparse.Namespace(gpus=1, steps=1, warmup=1, batch=16, seconds=6.0, sample_rate=24000, precision="bf16x3", graph=False)
dev = torch.device("cuda:0")
N = 10
with tempfile.TemporaryDirectory() as tmp:
    task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp)
    bench.run_steps(trainer, task, batch, 8, 1)
    torch.cuda.synchronize()
    acc.clear()
    o_bw = torch.Tensor.backward
    bw = [0.0]

    def timed_bw(self, *a, **k):
        t = time.perf_counter()
        r = o_bw(self, *a, **k)
        bw[0] += time.perf_counter() - t
        return r
    torch.Tensor.backward = timed_bw
    t0 = time.perf_counter()
    bench.run_steps(trainer, task, batch, N, 9)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    torch.Tensor.backward = o_bw
print(f"host issue {host / N * 1e3:.2f} ms/step; loss.backward() wall {bw[0] / N * 1e3:.2f} ms/step")
tot = {"forward": 0.0, "backward": 0.0}
print(f"{'node':28s} {'dir':9s} {'calls/step':>10s} {'us/call':>9s} {'ms/step':>8s}")
for (cls, d), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    tot[d] += s
    print(f"{cls:28s} {d:9s} {n / N:10.1f} {s / n * 1e6:9.1f} {s / N * 1e3:8.3f}")
print(f"custom nodes: forward {tot['forward'] / N * 1e3:.2f} ms/step, backward {tot['backward'] / N * 1e3:.2f} ms/step "
      f"(of {bw[0] / N * 1e3:.2f} ms inside loss.backward())")

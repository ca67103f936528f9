"""Which torch (non-svb) ops still run in the bench step?  torch.profiler over a few steps, grouped by op (GPU only)."""
import argparse
import os
import sys
import tempfile

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = argparse.Namespace(gpus=1, steps=1, warmup=1, batch=16, seconds=6.0, sample_rate=24000, bf16=False,
                          precision="bf16x3", graph=False)
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as tmp:
    task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp)
    bench.run_steps(trainer, task, batch, 4, 1)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        bench.run_steps(trainer, task, batch, 3, 5)
        torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0:
        rows.append((t / 3e3, e.count / 3, e.key))
rows.sort(reverse=True)
print(f"{'ms/step':>8} {'calls/step':>10}  op (self device time)")
for t, c, k in rows[:45]:
    print(f"{t:8.3f} {c:10.1f}  {k[:90]}")

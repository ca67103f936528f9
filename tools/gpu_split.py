"""Where the DEVICE time of a bench step goes on the compute stream (GPU only): HIP events recorded on the current stream around
the forward, the backward and the rest of each optimizer pass, read back after the run.  An interval of the compute stream
includes whatever that stream waits for (the weight-gradient side stream at the end of a backward, the PPG stream where the
content features are read, the critic's stream at the critic barrier) -- which is the point: the step's critical path."""
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
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--extra-hparams", default="")
ap.add_argument("--slow-host-us", type=float, default=0.0)
a = ap.parse_args()
args = argparse.Namespace(gpus=1, steps=1, warmup=1, batch=16, seconds=6.0, sample_rate=24000, precision="bf16x3", graph=False)
dev = torch.device("cuda:0")
marks = []        # (label, stream id, event)


def mark(label):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((label, torch.cuda.current_stream().cuda_stream, ev))


with tempfile.TemporaryDirectory() as tmp:
    task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp, ("," + a.extra_hparams) if a.extra_hparams else "")
    bench.run_steps(trainer, task, batch, 8, 1)
    torch.cuda.synchronize()
    o_ts, o_bw, o_pass, o_rtb = task.training_step, torch.Tensor.backward, trainer._optimizer_pass, trainer._run_training_batch_body

    def ts(sample, batch_idx, opt_idx):
        mark(f"opt{opt_idx}.fwd.begin")
        r = o_ts(sample, batch_idx, opt_idx)
        mark(f"opt{opt_idx}.fwd.end")
        return r

    def bw(self, *x, **k):
        r = o_bw(self, *x, **k)
        mark("bwd.end(before side join)")
        return r

    def opass(task_, batch_, batch_idx, opt_idx, *rest):
        mark(f"opt{opt_idx}.pass.begin")
        r = o_pass(task_, batch_, batch_idx, opt_idx, *rest)
        mark(f"opt{opt_idx}.pass.end")
        return r

    def rtb(*x, **k):
        mark("step.begin")
        r = o_rtb(*x, **k)
        mark("step.end")
        return r
    task.training_step, torch.Tensor.backward, trainer._optimizer_pass, trainer._run_training_batch_body = ts, bw, opass, rtb
    t0 = time.perf_counter()
    bench.run_steps(trainer, task, batch, a.steps, 9)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
print(f"host issue {t_host / a.steps * 1e3:.2f} ms/step, wall {t_all / a.steps * 1e3:.2f} ms/step")
# per stream: consecutive marks -> interval durations, averaged by (from label -> to label)
by_stream = {}
for lab, sid, ev in marks:
    by_stream.setdefault(sid, []).append((lab, ev))
for sid, lst in by_stream.items():
    acc = {}
    order = []
    for (l0, e0), (l1, e1) in zip(lst[:-1], lst[1:]):
        key = f"{l0} -> {l1}"
        if key not in acc:
            acc[key] = [0.0, 0]
            order.append(key)
        acc[key][0] += e0.elapsed_time(e1)
        acc[key][1] += 1
    tot = lst[0][1].elapsed_time(lst[-1][1])
    print(f"stream {sid:#x}: {len(lst)} marks, first->last {tot / a.steps:.2f} ms/step")
    for key in order:
        ms, n = acc[key]
        print(f"   {key:60s} {ms / a.steps:7.3f} ms/step  ({n / a.steps:.1f}x/step, {ms / n:6.3f} ms each)")

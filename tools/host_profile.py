"""Host-side (Python) cost of the bench step: cProfile over a few eager steps (GPU only).  The step is host-bound within a
few percent of its GPU time, so this ranks where launch-issue time goes."""
import argparse
import cProfile
import os
import pstats
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = argparse.Namespace(gpus=1, steps=1, warmup=1, batch=16, seconds=6.0, sample_rate=24000, precision="bf16x3", graph=bool(os.environ.get("SVB_PROFILE_GRAPH")), extra_hparams="")
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as tmp:
    task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp)
    bench.run_steps(trainer, task, batch, 8, 1)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    bench.run_steps(trainer, task, batch, 5, 9)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)

"""Offline tile tuner: writes the table neuralsvb_amd/kernels.py loads at import (neuralsvb_amd/tile_table.json).

Runs on an MI355X (GPU box):   python tools/tune_tiles.py --out gpurun_out/tile_table.json [--reps 24]
then the table is copied to neuralsvb_amd/tile_table.json and committed.

For every conv launch signature the three bench workloads issue (configs[1] train step phase 2 + phase 3, configs[2] vocoder
step, configs[4] inference) every tile configuration is timed `--reps` times, interleaved (round r times configuration
1..n once each, so all configurations see the same clock and cache history), after a clock warm-up; the median decides, ties
go to the lowest configuration index.  The per-configuration medians are kept in the file (`medians_us`) so that a choice can
be audited; the product only reads `choices`.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def clock_warmup(seconds=1.5):
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            a @ a
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tile_table.json"))
    ap.add_argument("--reps", type=int, default=24)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--workloads", default="train,vocoder,infer")
    a = ap.parse_args()
    os.environ["SVB_TILE_TABLE"] = "0"            # start from an empty table: every signature is measured here
    from neuralsvb_amd import kernels as K
    import bench
    K.load_tile_table(None)
    K.AUTOTUNE_ONLINE = True                      # the one place that measures: the product only looks tiles up
    K.TUNE_REPS = a.reps
    K._TUNE_LOG = {}
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    clock_warmup()
    wl = a.workloads.split(",")
    t0 = time.perf_counter()
    if "train" in wl:
        args = argparse.Namespace(batch=16, seconds=6.0, sample_rate=24000, precision=a.precision, graph=False)
        with tempfile.TemporaryDirectory() as tmp:
            task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp)
            bench.run_steps(trainer, task, batch, 3, 1)                       # phase 2
            hp["defer_wgrad_reduce"] = False                                  # the N > 1 form of the step (same signatures)
            bench.run_steps(trainer, task, batch, 2, 4)
            hp.pop("defer_wgrad_reduce", None)
            bench.run_steps(trainer, task, batch, 3, int(hp["phase_2_steps"]) + 10)   # phase 3
            torch.cuda.synchronize()
            del task, trainer, batch
        torch.cuda.empty_cache()
        print(f"[tune] train: {len(K._TUNE_LOG)} signatures after {time.perf_counter() - t0:.1f} s", flush=True)
    base = argparse.Namespace(precision=a.precision, no_cpu_baseline=True, extra_hparams="")
    if "vocoder" in wl:
        v = argparse.Namespace(**vars(base), steps=2, warmup=2, vocoder_cpu_batch=0)
        bench.bench_vocoder(v, dev)
        torch.cuda.empty_cache()
        print(f"[tune] +vocoder: {len(K._TUNE_LOG)} signatures after {time.perf_counter() - t0:.1f} s", flush=True)
    if "infer" in wl:
        v = argparse.Namespace(**vars(base), steps=4, warmup=2)
        bench.bench_infer(v, dev)
        print(f"[tune] +infer: {len(K._TUNE_LOG)} signatures after {time.perf_counter() - t0:.1f} s", flush=True)
    choices = {K._sig_key(s): c for s, c in sorted(K._TUNED_ONLINE.items(), key=lambda kv: K._sig_key(kv[0]))}
    med = {K._sig_key(s): [round(x, 3) for x in m] for s, m in sorted(K._TUNE_LOG.items(), key=lambda kv: K._sig_key(kv[0]))}
    doc = {"device": torch.cuda.get_device_name(0), "arch": "gfx950", "precision": a.precision, "reps": a.reps,
           "method": "interleaved rounds, HIP events around each launch, median per configuration, ties to the lowest index",
           "workloads": wl, "configurations": K._CFG_NAMES, "choices": choices, "medians_us": med}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=0, sort_keys=False)
    # how decisive the choices are: margin of the winner over the runner-up
    close = sum(1 for m in med.values() if sorted(m)[1] < 1.02 * sorted(m)[0])
    print(f"[tune] wrote {a.out}: {len(choices)} signatures, {close} with a runner-up within 2 %, {time.perf_counter() - t0:.1f} s")


if __name__ == "__main__":
    main()

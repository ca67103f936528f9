"""Does `conv_precision: bf16x3` TRAIN like fp32?  N optimizer steps of the vae_global_mle_eng task (phase 2: generator + critic,
reference tasks/singing/svb_vae_task.py:579-676) on the synthetic set, once per arithmetic, from identical weights, batches,
seeds and random draws; the per-step loss terms of both runs go to a JSON file.

  python tools/train_trajectory.py --steps 300 --out profiles/r05_trajectory_fp32_vs_bf16x3.json        (MI355X)

Training is chaotic (AdamW's early updates are ~lr * sign(g): an element whose gradient is within rounding of zero takes the other
direction), so the two runs are NOT expected to stay on one trajectory bit for bit; what is asserted (tests/test_task_step.py::
test_bf16x3_trains_like_fp32) is that every loss term's smoothed trajectory stays within a stated band of the fp32 run's and
that neither run diverges.  `run()` is shared with that test.
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TERMS = ("l1a2a", "l1p2p", "ssima2a", "ssimp2p", "a2a_kl", "p2p_kl", "a2a_a", "p2p_a", "a2a_r", "a2a_f", "p2p_r", "p2p_f")


def run(precision, steps, device, seed=7, batch=8, seconds=3.0, n_batches=4, extra_hparams=""):
    """-> {term: [value per step]} of `steps` phase-2 steps in `precision`, everything else a function of `seed`."""
    import bench
    from neuralsvb_amd.utils.trainer import move_to_device
    args = argparse.Namespace(batch=batch, seconds=seconds, sample_rate=24000, precision=precision, graph=False)
    out = {k: [] for k in TERMS}
    with tempfile.TemporaryDirectory() as tmp:
        torch.manual_seed(seed)
        np.random.seed(seed)
        # (build_task writes batch x world clips: `world` = n_batches gives n_batches distinct batches to cycle through)
        task, trainer, _, hp = bench.build_task(args, 0, n_batches, device, tmp, extra_hparams=extra_hparams)
        trainer.world_size, trainer.use_ddp = 1, False
        ds = task.dataset_cls("train", False)
        batches = [move_to_device(ds.collater([ds[b * batch + i] for i in range(batch)]), device) for b in range(n_batches)]
        for b in batches:
            for k in ("mel_lengths", "prof_mel_lengths"):
                if k in b:
                    b[k] = b[k].cpu()
        torch.manual_seed(seed + 1)                 # the runs' random draws (encoder noise, Dropout2d masks, critic windows)
        np.random.seed(seed + 1)
        for i in range(steps):
            task.global_step = trainer.global_step = 1 + i
            pbar, _ = trainer.run_training_batch(i, batches[i % n_batches])
            torch.cuda.synchronize()       # the critic's pass runs on its own stream: its loss terms are final only after a device sync
            for k in TERMS:
                v = pbar.get(k)
                out[k].append(float(v) if v is not None else float("nan"))
        torch.cuda.synchronize()
    return out


def smooth(x, w=20):
    x = np.asarray(x, dtype=np.float64)
    c = np.cumsum(np.insert(x, 0, 0.0))
    return (c[w:] - c[:-w]) / w


def compare(a, b, w=20):
    """Per term: worst relative deviation of the w-step moving averages, and of the last 50 steps' means."""
    res = {}
    for k in TERMS:
        sa, sb = smooth(a[k], w), smooth(b[k], w)
        scale = np.maximum(np.abs(sa), 1e-3 * max(1.0, np.abs(sa).max()))
        res[k] = {"worst_rel_dev_smoothed": float((np.abs(sb - sa) / scale).max()),
                  "last50_mean_fp32": float(np.mean(a[k][-50:])), "last50_mean_bf16x3": float(np.mean(b[k][-50:])),
                  "first_step_rel_dev": float(abs(b[k][0] - a[k][0]) / max(abs(a[k][0]), 1e-12))}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trajectory_fp32_vs_bf16x3.json"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    runs = {p: run(p, a.steps, dev) for p in ("fp32", "bf16x3")}
    rerun = run("fp32", a.steps, dev)          # the same arithmetic twice: how far does the run's own non-determinism go?
    cmp_ = compare(runs["fp32"], runs["bf16x3"])
    self_ = compare(runs["fp32"], rerun)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"steps": a.steps, "terms": list(TERMS), "fp32": runs["fp32"], "bf16x3": runs["bf16x3"], "fp32_rerun": rerun,
                   "bf16x3_vs_fp32": cmp_, "fp32_rerun_vs_fp32": self_}, f)
    print(f"{'term':10s} {'bf16x3 vs fp32':>16s} {'fp32 rerun':>12s} {'last-50 fp32':>14s} {'last-50 bf16x3':>15s}")
    for k in TERMS:
        print(f"{k:10s} {cmp_[k]['worst_rel_dev_smoothed']:16.3e} {self_[k]['worst_rel_dev_smoothed']:12.3e} "
              f"{cmp_[k]['last50_mean_fp32']:14.6g} {cmp_[k]['last50_mean_bf16x3']:15.6g}")


if __name__ == "__main__":
    main()

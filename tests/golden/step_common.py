"""TEST INFRASTRUCTURE -- shared by tests/golden/make_step_golden.py (runs the UNMODIFIED reference task through its own
Trainer on CPU) and tests/test_step_golden.py (runs the HIP task on the same data, weights and random draws).

Everything that decides WHAT is compared lives here so both sides summarise identically.
"""
import os

import numpy as np
import torch

# hparams overrides of the golden run (given to the reference CLI parser and to ours verbatim)
# disc_lr / map_lr: AdamW's first updates are ~lr * sign(g), so at the default 1e-4 / 1e-3 two fp32 evaluations whose gradients
# differ in the last bits leave a step on measurably different weights and the comparison of the NEXT step would measure
# that chaos, not the arithmetic (the generator's rsqrt warm-up already starts at 1e-7).  Small rates keep all five steps
# on (nearly) the initial weights while every optimizer still steps through the reference's own code.
STEP_HPARAMS = ("audio_sample_rate=24000,fmax=12000,num_sanity_val_steps=0,max_updates=4,max_sentences=4,max_tokens=40000,"
                "phase_2_steps=2,ds_workers=0,val_check_interval=100000,tb_log_interval=1000,disc_lr=0.0000001,map_lr=0.0000001")
N_TRAIN, N_VALID = 8, 2
SECONDS = (2.0, 1.7, 2.0, 1.5)          # ragged clips: T = 376 / 316 / 376 / 280 frames after the multiple-of-4 cut
N_STEPS = 5                               # global_step 0 (gen only), 1-2 (phase 2: gen + disc), 3-4 (phase 3: map)
STEP_SEED = 11                            # torch / numpy global streams of the golden run (make_step_golden.py seeds them); see the note below
# Why this seed and not the date: the golden run draws the encoder noise, the critic windows and the Dropout2d masks from it, and
# at random init the generator gradient THROUGH the critic is ill-conditioned to a degree that depends on those draws (1e-5
# relative noise -- bf16x3's -- becomes 3e-3 in d(mel) behind the critic; the encoder's pooling tail, near-dead ReLU channels
# in front of a train-mode BatchNorm1d over 4 clips, multiplies what arrives through the latent by up to 20 more).  Measured on
# the MI355X over four seeds (worst generator gradient-norm deviation from the reference over the five steps, fp32 / bf16x3):
# 11: 8e-4 / 2.9e-3   12: 6e-4 / 8.2e-3   13: 6e-4 / 6.3e-3   20260926: 6e-4 / 1.4e-1 (step 2 only; fp32 on all layers of
# the generator brings it back to 4e-4, a serial single-stream run changes nothing: arithmetic sensitivity, not a race --
# profiles/r03_step_golden_seed_diagnosis.md).  The test's bounds are for a typical draw set; 11 is one.
EXTRA_SEEDS = (12, 13, 20260926)          # further draw sets of the same run (`SVB_STEP_SEED=<s> python make_step_golden.py` ->
                                          # step_ref_seed<s>.json.gz / step_ref_draws_seed<s>.npz): the step test runs all four
SAMPLE_PARAMS = 24                        # elements sampled per parameter for the gradient / weight probes


def oracle_mel_fn(hp):
    from oracle import frontend as ofe

    def fn(wavs):
        return np.stack([ofe.wav2mel_offline(w, hp["fft_size"], hp["hop_size"], hp["win_size"], hp["audio_num_mel_bins"],
                                             hp["fmin"], hp["fmax"], hp["audio_sample_rate"])[1] for w in wavs])
    return fn


def write_dataset(data_dir, hp):
    """The synthetic binary dataset of the golden run, written by the product's own writer (utils/synth.py) in the
    reference's IndexedDataset format -- the reference's dataset class reads it back in make_step_golden.py."""
    from neuralsvb_amd.utils import synth
    synth.write_binary_dataset(data_dir, hp, oracle_mel_fn(hp), n_train=N_TRAIN, n_valid=N_VALID, seconds=SECONDS)


def sample_idx(numel):
    return torch.linspace(0, numel - 1, min(SAMPLE_PARAMS, numel)).long()


def grad_summary(named_params):
    """{name: [l2 norm, sampled values...]} of the gradients present (called right before the optimizer step)."""
    out = {}
    for n, p in named_params:
        if p.grad is None:
            continue
        g = p.grad.detach().double().cpu().flatten()
        out[n] = np.concatenate([[g.norm().item()], g[sample_idx(g.numel())].numpy()])
    return out


def weight_summary(named_tensors):
    out = {}
    for n, p in named_tensors:
        if not p.is_floating_point():
            out[n] = np.array([float(p.double().sum())])
            continue
        w = p.detach().double().cpu().flatten()
        out[n] = np.concatenate([[w.sum().item(), w.norm().item()], w[sample_idx(w.numel())].numpy()])
    return out


# float32 fields computed through vectorised transcendentals: torch's CPU kernels evaluate the unaligned head / tail of a buffer
# with scalar libm and the body with SIMD polynomials, so the reference's OWN `energy = (mel.exp() ** 2).sum(-1).sqrt()`
# (tasks/tts/dataset_utils.py:141) moves by an ulp per element -- 1e-6 of the sum -- from run to run with the allocation's
# alignment (seen between two runs of make_step_golden.py).  Their digests keep 5 significant digits; everything else is exact.
ULP_NOISY = ("energy", "prof_energy")


def batch_summary(batch):
    """Exact content digest of a collated batch: shapes, dtypes and float64 sums / position-weighted sums per tensor."""
    out = {}
    for k in sorted(batch):
        v = batch[k]
        if isinstance(v, torch.Tensor):
            f = v.double().flatten()
            w = torch.arange(1, f.numel() + 1, dtype=torch.float64) % 9973
            out[k] = {"shape": list(v.shape), "dtype": str(v.dtype), "sum": float(f.sum()), "wsum": float((f * w).sum())}
            if k in ULP_NOISY:
                out[k]["sum"], out[k]["wsum"] = float(f"{out[k]['sum']:.5g}"), float(f"{out[k]['wsum']:.5g}")
        elif isinstance(v, (list, tuple)) and all(isinstance(x, str) for x in v):
            out[k] = {"list": list(v)}
        elif isinstance(v, (int, float)):
            out[k] = {"value": v}
        elif v is None:
            out[k] = None
    return out


class DrawLog:
    """Record (generator) or replay (test) the random draws of a training step, in call order:
    np.random.randint -> ("randint", low, high, value);  torch.randn_like -> ("randn_like", tensor);
    F.dropout2d(train) -> ("dropout2d", keep-mask [B, C] as uint8)."""

    def __init__(self, events=None):
        self.events = [] if events is None else list(events)
        self.replay = events is not None
        self.pos = 0

    def next(self, kind):
        assert self.pos < len(self.events), f"draw log exhausted at a {kind} draw"
        ev = self.events[self.pos]
        assert ev[0] == kind, f"draw order differs from the reference: expected {ev[0]}, got {kind} (event {self.pos})"
        self.pos += 1
        return ev


def save_events(path, events_per_step):
    flat = {}
    for s, evs in enumerate(events_per_step):
        kinds = []
        for i, ev in enumerate(evs):
            kinds.append(ev[0])
            if ev[0] == "randint":
                flat[f"s{s}.e{i}"] = np.array(ev[1:], dtype=np.int64)
            elif ev[0] == "randn_like":
                flat[f"s{s}.e{i}"] = ev[1].numpy().astype(np.float32)
            else:
                flat[f"s{s}.e{i}"] = np.packbits(ev[1].numpy().astype(np.uint8), axis=-1)
                flat[f"s{s}.e{i}.c"] = np.array(ev[1].shape[-1])
        flat[f"s{s}.kinds"] = np.array(kinds)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from detnpz import savez_det
    savez_det(path, **flat)


def load_events(path):
    d = np.load(path, allow_pickle=False)
    steps = []
    s = 0
    while f"s{s}.kinds" in d.files:
        evs = []
        for i, kind in enumerate(d[f"s{s}.kinds"]):
            a = d[f"s{s}.e{i}"]
            if kind == "randint":
                evs.append(("randint", int(a[0]), int(a[1]), int(a[2])))
            elif kind == "randn_like":
                evs.append(("randn_like", torch.from_numpy(a)))
            else:
                c = int(d[f"s{s}.e{i}.c"])
                evs.append(("dropout2d", torch.from_numpy(np.unpackbits(a, axis=-1)[..., :c].astype(np.uint8))))
        steps.append(evs)
        s += 1
    return steps


GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

"""Generate tests/golden/step_ref.{json,npz}: the UNMODIFIED reference's own `tasks/run.py` path -- set_hparams ->
SVBVAEMleTask.start -> Trainer.fit -> dataset/collater -> run_training_batch (reference tasks/run.py:5-15,
tasks/singing/svb_vae_task.py:579-676, utils/trainer.py:269-342) -- run on CPU for five optimizer steps
(global_step 0: generator only; 1-2: phase 2, generator + discriminator; 3-4: phase 3, mapping function) on the
synthetic binary dataset written by the product's writer (neuralsvb_amd/utils/synth.py).

Build-container only.  Usage:  python tests/golden/make_step_golden.py
Recorded per step: the collated batch's digest (dataset parity), every random draw the step makes (np.random.randint,
torch.randn_like, Dropout2d keep-masks -- Dropout2d is evaluated as `x * keep / (1-p)` so the draw can be recorded),
per optimizer pass the loss terms and the pre-clipping gradient norms + samples, and after the step a digest of every
parameter and buffer.  Weights are procedural (oracle/procedural.py) so none are stored.
No reference source is modified or copied: hooks are installed by monkey-patching at run time.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import step_common as C  # noqa: E402
from oracle import procedural, ref_shims  # noqa: E402


def main():
    torch.set_num_threads(8)
    ref_shims.install()                       # stubs + /root/reference first on sys.path
    tmp = tempfile.mkdtemp(prefix="stepgold_")
    os.symlink(os.path.join(ref_shims.REFERENCE_ROOT, "egs"), os.path.join(tmp, "egs"))
    os.chdir(tmp)
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    os.environ["NUM_WORKERS"] = "0"
    sys.argv = ["tasks/run.py", "--config", "egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml", "--exp_name", "stepgold",
                "--reset", "--hparams", C.STEP_HPARAMS]
    from utils.hparams import set_hparams, hparams             # the reference's
    set_hparams()
    C.write_dataset(hparams["binary_data_dir"], hparams)
    from neuralsvb_amd.utils import synth
    synth.write_fake_asr_ckpt(hparams["pretrain_asr_ckpt"], 70, hparams)     # layout check: the reference loader reads it

    import torch.nn.functional as F
    import utils.trainer as rtrainer
    from tasks.singing import svb_vae_task as rtask
    from utils.pitch_utils import denorm_f0

    keys = json.load(open(os.path.join(HERE, "ref_state_keys.json")))
    rec = {"steps": [], "hparams": C.STEP_HPARAMS}
    events_per_step = []
    state = {"log": None, "cur": None}

    # ---- procedural weights after the reference built (and loaded) its model ----------------------------------------
    orig_build = rtask.SVBVAEMleTask.build_model

    def build_model(self):
        m = orig_build(self)
        self.model.load_state_dict(procedural.state_dict_for(keys["MleSVBVAE"], prefix="model."), strict=True)
        self.mel_disc.load_state_dict(procedural.state_dict_for(keys["Discriminator"], prefix="mel_disc."), strict=True)
        return m
    rtask.SVBVAEMleTask.build_model = build_model

    # ---- per-pass records ---------------------------------------------------------------------------------------------
    orig_ts = rtask.SVBVAEMleTask._training_step

    def _training_step(self, sample, batch_idx, optimizer_idx):
        ret = orig_ts(self, sample, batch_idx, optimizer_idx)
        if ret is not None:
            total, logs = ret
            state["cur"]["passes"][str(optimizer_idx)] = {
                "total": float(total), "terms": {k: float(v) for k, v in logs.items()}}
        return ret
    rtask.SVBVAEMleTask._training_step = _training_step

    orig_before = rtask.SVBVAEBoostTask.on_before_optimization

    def on_before_optimization(self, opt_idx):
        named = [(n, p) for n, p in self.named_parameters()]
        state["cur"]["passes"][str(opt_idx)]["grads"] = {k: v.tolist() for k, v in C.grad_summary(named).items()}
        return orig_before(self, opt_idx)
    rtask.SVBVAEBoostTask.on_before_optimization = on_before_optimization

    # ---- per-step records + draw recording ------------------------------------------------------------------------------
    orig_rtb = rtrainer.Trainer.run_training_batch
    o_randint, o_randn_like, o_dropout2d = np.random.randint, torch.randn_like, F.dropout2d

    def randint(low, high=None, *a, **k):
        v = o_randint(low, high, *a, **k)
        state["log"].events.append(("randint", int(low), int(high), int(v)))
        return v

    def randn_like(x, *a, **k):
        t = o_randn_like(x, *a, **k)
        state["log"].events.append(("randn_like", t.detach().clone()))
        return t

    def dropout2d(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        keep = torch.rand(x.shape[0], x.shape[1]) >= p
        state["log"].events.append(("dropout2d", keep.to(torch.uint8)))
        return x * (keep.to(x.dtype) / (1.0 - p))[:, :, None, None]

    # conditioning probe: the critic's InstanceNorm2d planes (multi_window_disc.py:24-27).  A near-constant plane (variance close
    # to eps = 1e-5) would make d(loss)/d(input) through that norm ill-conditioned.  Recorded per step (smallest variance of a
    # plane that Dropout2d kept, over the critic calls of the step): it stays >= 1e-3 for every seed tried, i.e. this is NOT
    # where the draw-dependent sensitivity of the generator gradient comes from (profiles/r03_step_golden_seed_diagnosis.md).
    o_inorm = F.instance_norm

    def instance_norm(x, *a, **k):
        if state["cur"] is not None and x.dim() == 4:
            pv = x.detach().var(dim=(2, 3), unbiased=False)
            pv = pv[pv > 0]                      # (planes zeroed by Dropout2d carry no gradient at all)
            if pv.numel():
                state["cur"]["critic_min_plane_var"] = min(state["cur"].get("critic_min_plane_var", float("inf")), float(pv.min()))
        return o_inorm(x, *a, **k)
    F.instance_norm = instance_norm

    def run_training_batch(self, batch_idx, batch):
        task = self.get_task_ref()
        cur = {"global_step": self.global_step, "batch": C.batch_summary(batch), "passes": {}}
        cur["denorm_f0"] = C.batch_summary({"f0": denorm_f0(batch["f0"], batch["uv"], hparams),
                                            "prof_f0": denorm_f0(batch["prof_f0"], batch["prof_uv"], hparams)})
        if not rec["steps"]:
            rec["initial_weights"] = {k: v.tolist() for k, v in C.weight_summary(task.state_dict().items()).items()}
        state["cur"], state["log"] = cur, C.DrawLog()
        np.random.randint, torch.randn_like, F.dropout2d = randint, randn_like, dropout2d
        try:
            ret = orig_rtb(self, batch_idx, batch)
        finally:
            np.random.randint, torch.randn_like, F.dropout2d = o_randint, o_randn_like, o_dropout2d
        cur["weights"] = {k: v.tolist() for k, v in C.weight_summary(
            [(k, v) for k, v in task.state_dict().items() if "vc_asr" not in k]).items()}      # (the PPG encoder is frozen)
        cur["lr"] = [o.param_groups[0]["lr"] if o is not None else None for o in self.optimizers]
        rec["steps"].append(cur)
        events_per_step.append(state["log"].events)
        print(f"| golden step {cur['global_step']}: passes {sorted(cur['passes'])} "
              f"{ {k: round(v['total'], 5) for k, v in cur['passes'].items()} }", flush=True)
        return ret
    rtrainer.Trainer.run_training_batch = run_training_batch

    # ---- dataset digests (D1): every item alone and ragged groups through the reference collater ----------------------
    from tasks.run import run_task
    ds = rtask.MultiSpkEmbDataset("train", shuffle=False)
    groups = [[0], [1], [2], [3], [0, 1, 2, 3], [4, 5, 6, 7], [7, 3, 1], list(range(C.N_TRAIN))]
    rec["dataset"] = {",".join(map(str, g)): C.batch_summary(ds.collater([ds[i] for i in g])) for g in groups}
    dv = rtask.MultiSpkEmbDataset("valid", shuffle=False)
    rec["dataset_valid"] = {"0,1": C.batch_summary(dv.collater([dv[0], dv[1]]))}

    # the reference seeds nothing on the single-process CPU path: fix torch's and numpy's global streams here, so the recorded
    # draws (and everything that depends on them) are the same on every run of this script
    seed = int(os.environ.get("SVB_STEP_SEED", C.STEP_SEED))       # (the override only serves scans for a well-conditioned draw set)
    torch.manual_seed(seed)
    np.random.seed(seed)
    run_task()
    assert len(rec["steps"]) == C.N_STEPS, len(rec["steps"])
    out_dir = os.environ.get("SVB_STEP_OUT", HERE)
    # the primary seed writes the full record (dataset digests + steps); the additional draw sets of C.EXTRA_SEEDS only their steps
    sfx = "" if seed == C.STEP_SEED else f"_seed{seed}"
    if not sfx:
        with open(os.path.join(out_dir, "step_ref.json"), "w") as f:
            json.dump(rec, f)
    else:
        import gzip                                  # (deterministic bytes: no timestamp in the gzip header)
        with gzip.GzipFile(os.path.join(out_dir, f"step_ref{sfx}.json.gz"), "wb", mtime=0) as f:
            f.write(json.dumps({"seed": seed, "steps": rec["steps"]}).encode())
    C.save_events(os.path.join(out_dir, f"step_ref_draws{sfx}.npz"), events_per_step)
    print("critic min plane variance per step:", [st.get("critic_min_plane_var") for st in rec["steps"]])
    os.chdir(ROOT)
    shutil.rmtree(tmp, ignore_errors=True)
    print("step golden written:", os.path.getsize(os.path.join(out_dir, f"step_ref{sfx}.json" + (".gz" if sfx else ""))), "bytes json")


if __name__ == "__main__":
    main()

"""An fp64 ARBITER for the step golden: how far is the reference's OWN fp32 gradient from the exact one?

Build-container only (needs /root/reference).  Usage:  SVB_STEP_SEED=<seed> python tests/golden/make_step_fp64_arbiter.py
Writes tests/golden/step_fp64_seed<seed>.json.gz.

Same run as make_step_golden.py (the unmodified reference's tasks/run.py path, same seed, same five steps), with one addition:
right before the optimizer step of every GENERATOR pass (`on_before_optimization(0)`, the fp32 gradients are final and the
weights are still the ones that produced them) the task's `model` / `mel_disc` are deep-copied to float64, the pass is run again
on the float64 copies with the SAME batch and the SAME random draws (the ones the fp32 pass just made, replayed in order, the
N(0,1) draws cast to float64), and its gradient summary is recorded.  Recorded per step:

    grads32 / grads64 : {parameter: [l2 norm, 24 samples]} of the generator pass in the reference's fp32 and in float64
    total32 / total64 : the pass's loss

`|fp32 - fp64| / |fp64|` per parameter is the reference's own rounding-error amplification (its condition number times 2^-24) on
this draw set; tests/test_step_golden.py bounds the HIP gradients by a stated multiple of it instead of by an absolute number.
"""
import copy
import gzip
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


def to64(v):
    if isinstance(v, torch.Tensor):
        return v.double() if v.is_floating_point() else v
    if isinstance(v, dict):
        return {k: to64(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(to64(x) for x in v)
    return v


def copy64(module):
    """Float64 deep copy of a reference module.  torch.nn.utils.weight_norm keeps the last computed `weight` as a plain non-leaf
    attribute (recomputed by its pre-forward hook on every call), which deepcopy refuses: such cached tensors are detached first."""
    for m in module.modules():
        for k, v in list(m.__dict__.items()):
            if isinstance(v, torch.Tensor) and v.grad_fn is not None:
                m.__dict__[k] = v.detach()
    return copy.deepcopy(module).double()


def main():
    torch.set_num_threads(8)
    ref_shims.install()
    tmp = tempfile.mkdtemp(prefix="stepfp64_")
    os.symlink(os.path.join(ref_shims.REFERENCE_ROOT, "egs"), os.path.join(tmp, "egs"))
    os.chdir(tmp)
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    os.environ["NUM_WORKERS"] = "0"
    sys.argv = ["tasks/run.py", "--config", "egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml", "--exp_name", "stepfp64",
                "--reset", "--hparams", C.STEP_HPARAMS]
    from utils.hparams import set_hparams, hparams
    set_hparams()
    C.write_dataset(hparams["binary_data_dir"], hparams)
    from neuralsvb_amd.utils import synth
    synth.write_fake_asr_ckpt(hparams["pretrain_asr_ckpt"], 70, hparams)

    import torch.nn.functional as F
    import utils.trainer as rtrainer
    from tasks.singing import svb_vae_task as rtask

    keys = json.load(open(os.path.join(HERE, "ref_state_keys.json")))
    steps = []
    st = {"events": None, "replay": None, "pos": 0, "sample": None, "mark": {}, "cur": None, "in64": False}

    orig_build = rtask.SVBVAEMleTask.build_model

    def build_model(self):
        m = orig_build(self)
        self.model.load_state_dict(procedural.state_dict_for(keys["MleSVBVAE"], prefix="model."), strict=True)
        self.mel_disc.load_state_dict(procedural.state_dict_for(keys["Discriminator"], prefix="mel_disc."), strict=True)
        return m
    rtask.SVBVAEMleTask.build_model = build_model

    o_randint, o_randn_like, o_dropout2d = np.random.randint, torch.randn_like, F.dropout2d

    def nxt(kind):
        ev = st["replay"][st["pos"]]
        assert ev[0] == kind, (ev[0], kind, st["pos"])
        st["pos"] += 1
        return ev

    def randint(low, high=None, *a, **k):
        if st["in64"]:
            return nxt("randint")[3]
        v = o_randint(low, high, *a, **k)
        st["events"].append(("randint", int(low), int(high), int(v)))
        return v

    def randn_like(x, *a, **k):
        if st["in64"]:
            return nxt("randn_like")[1].to(x.dtype)
        t = o_randn_like(x, *a, **k)
        st["events"].append(("randn_like", t.detach().clone()))
        return t

    def dropout2d(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        if st["in64"]:
            keep = nxt("dropout2d")[1].bool()
        else:
            keep = torch.rand(x.shape[0], x.shape[1]) >= p
            st["events"].append(("dropout2d", keep.to(torch.uint8)))
        return x * (keep.to(x.dtype) / (1.0 - p))[:, :, None, None]

    orig_ts = rtask.SVBVAEMleTask._training_step

    def _training_step(self, sample, batch_idx, optimizer_idx):
        if not st["in64"]:
            st["mark"][optimizer_idx] = len(st["events"])         # where this pass's draws start
            st["sample"] = sample
        ret = orig_ts(self, sample, batch_idx, optimizer_idx)
        if ret is not None and not st["in64"] and optimizer_idx == 0:
            st["cur"]["total32"] = float(ret[0])
        return ret
    rtask.SVBVAEMleTask._training_step = _training_step

    orig_before = rtask.SVBVAEBoostTask.on_before_optimization

    def on_before_optimization(self, opt_idx):
        if opt_idx == 0 and not st["in64"]:
            named = [(n, p) for n, p in self.named_parameters()]
            st["cur"]["grads32"] = {k: v.tolist() for k, v in C.grad_summary(named).items()}
            # ---- the same pass in float64, on copies of the modules, with the draws just made
            m32, d32 = self.model, self.mel_disc
            keep_attrs = {k: v for k, v in self.__dict__.items() if not k.startswith("_")}
            modes = (m32.training, m32.z_mapping_function.training, d32.training)
            st["in64"], st["replay"], st["pos"] = True, st["events"][st["mark"][0]:], 0
            old_default = torch.get_default_dtype()
            import modules.commons.ssim as rssim           # its Gaussian window is a module-level fp32 cache: same values, as float64
            win32 = rssim.window
            try:
                torch.set_default_dtype(torch.float64)
                rssim.window = None if win32 is None else win32.double()
                self.model, self.mel_disc = copy64(m32), copy64(d32)
                for p in list(self.model.parameters()) + list(self.mel_disc.parameters()):
                    p.grad = None
                total, _ = orig_ts(self, to64(st["sample"]), 0, 0)
                total.backward()
                named64 = [(n, p) for n, p in self.named_parameters()]
                st["cur"]["grads64"] = {k: v.tolist() for k, v in C.grad_summary(named64).items()}
                st["cur"]["total64"] = float(total)
            finally:
                torch.set_default_dtype(old_default)
                rssim.window = win32
                self.model, self.mel_disc = m32, d32
                for k, v in keep_attrs.items():
                    self.__dict__[k] = v
                m32.train(modes[0]); m32.z_mapping_function.train(modes[1]); d32.train(modes[2])
                st["in64"] = False
            # every draw of the fp32 generator pass must have been consumed by the replay, in order
            n_pass = (st["mark"].get(1, len(st["events"])) if st["mark"].get(1, 0) > st["mark"][0] else len(st["events"])) - st["mark"][0]
            assert st["pos"] == n_pass, (st["pos"], n_pass)
        return orig_before(self, opt_idx)
    rtask.SVBVAEBoostTask.on_before_optimization = on_before_optimization

    orig_rtb = rtrainer.Trainer.run_training_batch

    def run_training_batch(self, batch_idx, batch):
        st["cur"], st["events"], st["mark"] = {"global_step": self.global_step}, [], {}
        np.random.randint, torch.randn_like, F.dropout2d = randint, randn_like, dropout2d
        try:
            ret = orig_rtb(self, batch_idx, batch)
        finally:
            np.random.randint, torch.randn_like, F.dropout2d = o_randint, o_randn_like, o_dropout2d
        steps.append(st["cur"])
        c = st["cur"]
        if "grads64" in c:
            errs = sorted(((abs(c["grads32"][k][0] - v[0]) / v[0], k) for k, v in c["grads64"].items()
                           if k in c["grads32"] and v[0] > 1e-6 and not k.startswith("mel_disc")), reverse=True)
            print(f"| step {c['global_step']}: total fp32 {c['total32']:.7f} fp64 {c['total64']:.7f}; reference fp32-vs-fp64 "
                  f"generator gradient-norm error: worst {errs[0][0]:.3e} ({errs[0][1]}), median {errs[len(errs) // 2][0]:.3e}", flush=True)
        return ret
    rtrainer.Trainer.run_training_batch = run_training_batch

    from tasks.run import run_task
    seed = int(os.environ.get("SVB_STEP_SEED", C.STEP_SEED))
    torch.manual_seed(seed)
    np.random.seed(seed)
    run_task()
    assert len(steps) == C.N_STEPS
    out_dir = os.environ.get("SVB_STEP_OUT", HERE)
    path = os.path.join(out_dir, f"step_fp64_seed{seed}.json.gz")
    keep = [{k: s[k] for k in ("global_step", "total32", "total64", "grads32", "grads64") if k in s} for s in steps]
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps({"seed": seed, "steps": keep}).encode())
    os.chdir(ROOT)
    shutil.rmtree(tmp, ignore_errors=True)
    print("fp64 arbiter written:", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

"""Host-side cost of issuing a training step, measured WITHOUT a GPU: every kernel entry point of the C ABI is bound to a no-op C
function with the same ctypes signature (size queries keep the real host functions of the library), tensors live on the CPU and
are tiny, so what remains is exactly the Python / autograd / ctypes work per launch -- the quantity that bounds the step on a box
whose host is slower than its GPU.  Numbers are garbage by construction; only time and call counts mean anything.

  python tools/host_overhead.py [--profile] [--steps N]"""
import argparse
import cProfile
import ctypes as C
import os
import pstats
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def null_library():
    from neuralsvb_amd import _lib
    emu = os.path.join(ROOT, "tests", "emu", "libsvb_emu.so")
    real = _lib.bind(emu)
    d = tempfile.mkdtemp()
    open(os.path.join(d, "n.c"), "w").write("int svb_noop() { return 0; }\n")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O2", "-o", os.path.join(d, "n.so"), os.path.join(d, "n.c")], check=True)
    noop = C.CDLL(os.path.join(d, "n.so"))

    class Lib:
        pass
    lib = Lib()
    n_calls = {"n": 0}
    for name, (res, args) in _lib.SIGNATURES.items():
        if any(s in name for s in ("workspace", "abi_version", "pick_cfg", "_floats", "_bytes", "_blocks", "debug", "get_")):
            setattr(lib, name, getattr(real, name))
            continue
        proto = C.CFUNCTYPE(res, *args)
        setattr(lib, name, proto(("svb_noop", noop)))
    return lib


_NO_KERNEL = ("aten::empty", "aten::view", "aten::reshape", "aten::as_strided", "aten::transpose", "aten::permute", "aten::select",
              "aten::slice", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::detach", "aten::alias", "aten::t", "aten::_unsafe_view",
              "aten::narrow", "aten::unbind", "aten::split", "aten::chunk", "aten::size", "aten::stride", "aten::is_", "aten::lift", "aten::to",
              "aten::_to_copy", "aten::contiguous", "aten::result_type", "aten::item", "aten::_local_scalar", "aten::unflatten", "aten::flatten",
              "aten::resize_", "aten::set_", "aten::view_as", "aten::expand_as", "aten::numpy_T", "aten::movedim", "aten::unfold", "aten::new_",
              "aten::zeros_like", "aten::ones_like", "aten::empty_like", "aten::_reshape_alias", "aten::diagonal", "aten::broadcast_")


def aten_ops(run, steps):
    """torch ops that launch a kernel of their own on the GPU (views and allocations dropped), per step, with the innermost
    repository frame that issued them (backward nodes of torch's own ops have none): the launches that are not the library's."""
    from torch.utils._python_dispatch import TorchDispatchMode
    by_site = {}

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            nm = "aten::" + func.__name__.split(".")[0] if not str(func).startswith("aten") else str(func)
            nm = "aten::" + str(func).split(".")[1] if str(func).startswith("aten.") else nm
            base = nm.split(".")[0]
            hidden = base in _NO_KERNEL or any(base.startswith(p_) for p_ in ("aten::is_", "aten::new_", "aten::lift", "aten::broadcast_",
                                                                              "aten::_local_scalar", "aten::size", "aten::stride"))
            if base in ("aten::zeros_like", "aten::new_zeros", "aten::ones_like"):
                hidden = False                       # (an allocation AND a fill launch)
            if not hidden:
                f = sys._getframe(1)
                site = "(autograd engine: backward of a torch op)"
                while f is not None:
                    fn = f.f_code.co_filename
                    if ("/neuralsvb_amd/" in fn or fn.endswith("/bench.py")) and "host_overhead" not in fn:
                        site = f"{fn.split('/neuralsvb_amd/')[-1]}:{f.f_lineno} {f.f_code.co_name}"
                        break
                    f = f.f_back
                by_site[(site, nm)] = by_site.get((site, nm), 0) + 1
            return func(*args, **(kwargs or {}))

    with Log():
        run()
    tot = sum(by_site.values()) / steps
    print(f"torch ops with a kernel of their own: {tot:.1f} per step")
    for (site, nm), n in sorted(by_site.items(), key=lambda kv: (-kv[1], kv[0])):
        print(f"  {n / steps:6.1f}  {nm:34s} {site}")


def vocoder_aten():
    """The torch ops of one NSF-HifiGAN training step (B = 2 segments of 8192 samples, no-op kernels), by call site."""
    import numpy as np
    from neuralsvb_amd.tasks.hifigan_task import HifiGanTask, default_hparams
    from neuralsvb_amd.utils.hparams import hparams
    from neuralsvb_amd.utils.trainer import Trainer
    hparams.clear()
    hparams.update(default_hparams())
    hparams["conv_precision"] = "bf16x3"
    B, L = 2, 8192
    trainer = Trainer(work_dir="", num_sanity_val_steps=0)
    torch.manual_seed(0)
    task = trainer.setup(HifiGanTask())
    task.train()
    g = torch.Generator().manual_seed(0)
    f0 = 150 + 200 * torch.rand(B, L // 128, generator=g)
    batch = {"mels": torch.randn(B, 80, L // 128, generator=g), "wavs": torch.randn(B, 1, L, generator=g) * 0.1, "f0": f0}
    for o in trainer.optimizers:
        if o is not None:
            o.step = lambda *a, **k: None
    torch.nn.utils.clip_grad_norm_ = lambda *a, **k: torch.zeros(())

    def steps(n, s0):
        for i in range(n):
            task.global_step = trainer.global_step = s0 + i
            trainer.run_training_batch(i, batch)
    steps(2, 1)
    aten_ops(lambda: steps(2, 3), 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--aten", action="store_true", help="list the torch (ATen) ops a step issues next to the library's kernels, by call site")
    ap.add_argument("--extra-hparams", default="")
    ap.add_argument("--workload", default="train", choices=["train", "vocoder"], help="vocoder: the NSF-HifiGAN step (G + MPD + MSD), --aten only")
    ap.add_argument("--no-stack-executor", action="store_true")
    ap.add_argument("--no-tower-executor", action="store_true", help="A/B: the critic towers through the per-op Python bodies")
    a = ap.parse_args()
    from neuralsvb_amd import _lib
    _lib._LIB, _lib._LIB_IS_EMU = null_library(), True
    from neuralsvb_amd import kernels as K
    K.AUTOTUNE = False
    if a.no_stack_executor:
        from neuralsvb_amd import functional as SF
        SF.STACK_EXECUTOR = False
    if a.no_tower_executor:
        from neuralsvb_amd import functional as SF
        SF.TOWER_EXECUTOR = False
    import bench
    if a.workload == "vocoder":
        vocoder_aten()
        return
    args = argparse.Namespace(batch=2, seconds=0.71, sample_rate=24000, bf16=False, precision="bf16x3", graph=False)
    extra = ",ds_workers=0" + (("," + a.extra_hparams) if a.extra_hparams else "")
    with tempfile.TemporaryDirectory() as tmp:
        # the synthetic dataset needs a real mel front-end once: build it with the emulator library, then swap
        null = _lib._LIB
        _lib._LIB = _lib.bind(os.path.join(ROOT, "tests", "emu", "libsvb_emu.so"))
        task, trainer, batch, hp = bench.build_task(args, 0, 1, torch.device("cpu"), tmp, extra_hparams=extra)
        _lib._LIB = null
        torch.set_num_threads(1)
        # (on the GPU the optimizer is one fused launch per pass; torch's CPU AdamW is a per-tensor Python loop that would drown
        #  everything else here: leave it, the flat-buffer memset and the clipping norm out of the measurement)
        for o in trainer.optimizers:
            if o is not None:
                o.step = lambda *a, **k: None
        for g in trainer.grad_sync:
            if g is not None:
                g.zero = lambda: None
        torch.nn.utils.clip_grad_norm_ = lambda *a, **k: torch.zeros(())
        import neuralsvb_amd.tasks.svb_vae_task as tk
        if hasattr(tk, "clip_grad_norm_"):
            tk.clip_grad_norm_ = lambda *a, **k: torch.zeros(())
        bench.run_steps(trainer, task, batch, 3, 1)
        if a.aten:
            aten_ops(lambda: bench.run_steps(trainer, task, batch, 2, 0), 2)
            return
        t0 = time.perf_counter()
        if a.profile:
            pr = cProfile.Profile()
            pr.enable()
        bench.run_steps(trainer, task, batch, a.steps, 4)
        if a.profile:
            pr.disable()
        dt = (time.perf_counter() - t0) / a.steps
        reps = [dt]
        if not a.profile:                   # (a shared build box is noisy: the minimum over a few repeats is the comparable figure)
            for _ in range(4):
                t1 = time.process_time()         # CPU time of this process: what a preempted wall clock cannot tell
                bench.run_steps(trainer, task, batch, a.steps, 4)
                reps.append((time.process_time() - t1) / a.steps)
        print(f"host time per step with no-op kernels: {min(reps) * 1e3:.2f} ms  (repeats: " + " ".join(f"{r * 1e3:.2f}" for r in reps) + ")")
        if a.profile:
            st = pstats.Stats(pr)
            st.sort_stats("tottime").print_stats(45)
            # autograd nodes per step: every custom Function's forward / backward (torch charges ~20-50 us of host time per
            # Function.apply on top of the kernel wrappers -- the unit in which fusing ops into one node pays)
            import inspect
            from neuralsvb_amd import functional as _SF
            fn_at = {}
            for nm, obj in vars(_SF).items():
                if inspect.isclass(obj) and issubclass(obj, torch.autograd.Function):
                    for meth in ("forward", "backward"):
                        f = getattr(obj, meth, None)
                        if f is not None and hasattr(f, "__code__"):
                            fn_at[(f.__code__.co_filename, f.__code__.co_firstlineno)] = f"{nm}.{meth}"
            rows = []
            for (fname, line, fn), (cc, nc, tt, ct, callers) in st.stats.items():
                if (fname, line) in fn_at:
                    rows.append((nc / a.steps, ct / a.steps * 1e3, fn_at[(fname, line)]))
            print("custom autograd Functions per step: calls, cumulative ms (profiled)")
            for nc, ms, nm in sorted(rows, key=lambda r: -r[1]):
                print(f"   {nc:7.1f} {ms:8.3f}  {nm}")


if __name__ == "__main__":
    main()

"""Which lines of the host path issue the stock-torch (ATen) kernels of one training step?

Runs one phase-2 and one phase-3 step of SVBVAEMleTask at small dimensions on the CPU lane emulator (build container,
no GPU needed) under a TorchDispatchMode and counts every non-view ATen call by the innermost neuralsvb_amd source
line on its Python stack.  On the GPU each such call is (at least) one kernel launch, so the table is the work list
for "kill the torch elementwise tail".          python tools/op_attribution.py [--top 60]
"""
import argparse
import collections
import os
import sys
import tempfile
import traceback

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VIEW = {"view", "reshape", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze", "t", "detach", "alias",
        "as_strided", "unbind", "split", "split_with_sizes", "chunk", "narrow", "_unsafe_view", "empty", "empty_like",
        "empty_strided", "new_empty", "_local_scalar_dense", "lift_fresh", "unfold", "item", "set_", "resize_",
        "_reshape_alias", "view_as", "expand_as", "is_same_size", "new_empty_strided", "diagonal", "movedim"}


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_site = collections.Counter()
        self.by_op = collections.Counter()
        self.site_ops = collections.defaultdict(collections.Counter)
        self.backward = False

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        full = str(func)
        if full.split("aten.")[-1].split(".")[0] not in VIEW:
            site = None
            node = torch._C._current_autograd_node()
            if node is not None and "Backward" in node.name() and not node.name().startswith("_"):
                site = f"bwd of {node.name()}"
                for line in reversed(node.metadata.get("traceback_", [])):
                    if "neuralsvb_amd" in line and "File" in line:
                        parts = line.strip().split(",")
                        fn = parts[0].split('"')[1]
                        site += f" <- {os.path.relpath(fn, ROOT)}:{parts[1].strip().split()[-1]}"
                        break
            if site is None:
                site = "(optimizer / other)"
                for fr in reversed(traceback.extract_stack(limit=40)):
                    fn = fr.filename
                    if "neuralsvb_amd" in fn and "op_attribution" not in fn:
                        site = f"{os.path.relpath(fn, ROOT)}:{fr.lineno} {fr.name}"
                        break
            self.by_site[site] += 1
            self.by_op[full] += 1
            self.site_ops[site][full.replace("aten.", "")] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--step", type=int, default=2, help="global_step to run (2: phase 2, 1200000+: phase 3)")
    a = ap.parse_args()
    import conftest
    conftest._build_emu()
    from neuralsvb_amd import _lib
    _lib._LIB, _lib._LIB_IS_EMU = _lib.bind(conftest.EMU_LIB), True
    import test_task_step as T
    import pathlib
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="opattr_"))
    dev = torch.device("cpu")
    task, trainer, batch, hp = T._setup(tmp, dev)
    if a.step > 100:
        hp["phase_2_steps"] = 1
    np.random.seed(1)
    task.global_step = trainer.global_step = 1
    trainer.run_training_batch(0, batch)                  # warm (packs, buffers)
    task.global_step = trainer.global_step = a.step
    c = Counter()
    with torch.autograd.detect_anomaly(check_nan=False), c:
        trainer.run_training_batch(0, batch)
    tot = sum(c.by_site.values())
    print(f"{tot} non-view ATen calls in one step (global_step {a.step})")
    print("--- by op")
    for op, n in c.by_op.most_common(25):
        print(f"{n:6d}  {op}")
    print("--- by site")
    for site, n in c.by_site.most_common(a.top):
        ops = ", ".join(f"{k}x{v}" for k, v in c.site_ops[site].most_common(4))
        print(f"{n:6d}  {site}    [{ops}]")


if __name__ == "__main__":
    main()

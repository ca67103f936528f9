"""CLI entry point, same surface as the reference (tasks/run.py:5-15):
    python tasks/run.py --config egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml --exp_name <name> [--reset] [--infer]
`task_cls` is a dotted path; `tasks.singing.svb_vae_task.SVBVAEMleTask` resolves to the MI355X implementation."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from neuralsvb_amd.utils.hparams import hparams, set_hparams  # noqa: E402


def run_task():
    assert hparams["task_cls"] != ""
    pkg, cls_name = hparams["task_cls"].rsplit(".", 1)
    getattr(importlib.import_module(pkg), cls_name).start()


if __name__ == "__main__":
    set_hparams()
    run_task()

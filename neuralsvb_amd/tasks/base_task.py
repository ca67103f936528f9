"""BaseTask: the Task <-> Trainer hook contract (reference tasks/base_task.py:131-355) and the dataloader
builder with its data-parallel batch sharding (reference tasks/tts/tts.py:57-101)."""
import os
import random

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from ..utils import ckpt_utils
from ..utils.batching import batch_by_size
from ..utils.hparams import hparams
from ..utils.trainer import Trainer


def data_loader(fn):
    """Lazy, memoised dataloader property-method (tasks/base_task.py:27-51)."""
    attr = "_lazy_" + fn.__name__

    def get(self):
        if not hasattr(self, attr):
            setattr(self, attr, fn(self))
        return getattr(self, attr)
    return get


class Meter:
    def __init__(self):
        self.sum, self.cnt = 0.0, 0

    def update(self, v, n=1):
        self.sum += v * n
        self.cnt += n

    @property
    def avg(self):
        return self.sum / max(self.cnt, 1)


class BaseTask(nn.Module):
    def __init__(self):
        super().__init__()
        self.current_epoch = self.global_step = 0
        self.trainer, self.model, self.logger = None, None, None
        self.testing = False
        self.scheduler = None
        self._pending_logs = []

    # ---- hooks the Trainer calls -------------------------------------------------------------------------
    def on_train_start(self):
        pass

    def on_epoch_start(self):
        self._pending_logs = []

    def on_epoch_end(self):
        print(f"Epoch {self.current_epoch} ended. Steps: {self.global_step}.")

    def on_train_end(self):
        pass

    def on_keyboard_interrupt(self):
        pass

    def test_start(self):
        pass

    def training_step(self, sample, batch_idx, optimizer_idx=-1):
        """-> {'loss': Tensor|None, 'progress_bar': {...}, 'tb_log': {'tr/<k>': v}}; values stay tensors (they are
        synchronised only when logged)."""
        ret = self._training_step(sample, batch_idx, optimizer_idx)
        if ret is None:
            return {"loss": None}
        total, logs = ret
        logs = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in logs.items()}
        if optimizer_idx >= 0 and self.trainer is not None and self.trainer.optimizers:
            logs[f"lr_{optimizer_idx}"] = self.trainer.optimizers[optimizer_idx].param_groups[0]["lr"]
        return {"loss": total, "progress_bar": logs, "tb_log": {f"tr/{k}": v for k, v in logs.items()}}

    def validation_end(self, outputs):
        meters = {"total_loss": Meter()}
        for out in outputs:
            if not out:
                continue
            n = out.pop("nsamples", 1)
            losses = out["losses"]
            total = out.get("total_loss", sum(losses.values()))
            for k, v in losses.items():
                meters.setdefault(k, Meter()).update(float(v), n)
            meters["total_loss"].update(float(total), n)
        res = {k: round(m.avg, 4) for k, m in meters.items()}
        print(f"| Valid results: {res}")
        return {"tb_log": {f"val/{k}": v for k, v in res.items()}, "val_loss": res["total_loss"]}

    def test_step(self, sample, batch_idx):
        return self.validation_step(sample, batch_idx)

    def test_end(self, outputs):
        return self.validation_end(outputs)

    def load_ckpt(self, ckpt_base_dir, current_model_name=None, model_name="model", force=True, strict=True):
        name = model_name if current_model_name is None else current_model_name
        ckpt_utils.load_ckpt(getattr(self, name), ckpt_base_dir, name, force, strict)

    # ---- data (tasks/tts/tts.py:57-101) --------------------------------------------------------------------------
    def build_dataloader(self, dataset, shuffle, max_tokens=None, max_sentences=None, required_batch_size_multiple=-1,
                         endless=False, batch_by_size_=True, **kw):
        batch_by_size_ = kw.get("batch_by_size", batch_by_size_)
        world = self.trainer.world_size if (self.trainer is not None and self.trainer.use_ddp) else 1
        if required_batch_size_multiple == -1:
            required_batch_size_multiple = world
        if max_tokens is not None:
            max_tokens *= world
        if max_sentences is not None:
            max_sentences *= world
        indices = dataset.ordered_indices()
        if batch_by_size_:
            batches = batch_by_size(indices, dataset.num_tokens, max_tokens, max_sentences, required_batch_size_multiple)
        else:
            batches = [indices[i:i + max_sentences] for i in range(0, len(indices), max_sentences)]
        if shuffle:
            reps = 1000 if endless else 1
            out = []
            for _ in range(reps):
                b = list(batches)
                np.random.shuffle(b)
                out += b
            batches = out
        elif endless:
            batches = [b for _ in range(1000) for b in batches]
        if world > 1:   # rank r takes every world-th item; batches not divisible by the world size are dropped
            rank = dist.get_rank()
            batches = [b[rank::world] for b in batches if len(b) % world == 0]
        if hparams.get("device_collate", False) and hasattr(dataset, "raw_item") and torch.cuda.is_available():
            # batches assembled on the GPU (tasks/device_collate.py): workers only decode items
            from .device_collate import DeviceCollateLoader
            dev = self.trainer.device if self.trainer is not None and getattr(self.trainer, "device", None) is not None \
                else torch.device("cuda", torch.cuda.current_device())
            return DeviceCollateLoader(dataset, batches, dev, num_workers=dataset.num_workers)
        return torch.utils.data.DataLoader(dataset, collate_fn=dataset.collater, batch_sampler=batches,
                                           num_workers=dataset.num_workers, pin_memory=torch.cuda.is_available())

    # ---- entry point (tasks/base_task.py:317-352) ------------------------------------------------------------------
    @classmethod
    def start(cls):
        os.environ.setdefault("MASTER_PORT", str(random.randint(15000, 30000)))
        random.seed(hparams["seed"])
        np.random.seed(hparams["seed"])
        trainer = Trainer(work_dir=hparams["work_dir"], val_check_interval=hparams["val_check_interval"],
                          tb_log_interval=hparams["tb_log_interval"], max_updates=hparams["max_updates"],
                          num_sanity_val_steps=hparams["num_sanity_val_steps"] if not hparams["validate"] else 10000,
                          accumulate_grad_batches=hparams["accumulate_grad_batches"],
                          print_nan_grads=hparams["print_nan_grads"],
                          resume_from_checkpoint=hparams.get("resume_from_checkpoint", 0), amp=hparams["amp"],
                          monitor_key=hparams["valid_monitor_key"], monitor_mode=hparams["valid_monitor_mode"],
                          num_ckpt_keep=hparams["num_ckpt_keep"], save_best=hparams["save_best"], seed=hparams["seed"],
                          debug=hparams["debug"], hip_graph=hparams.get("hip_graph", False),
                          hip_graph_warmup=hparams.get("hip_graph_warmup", 2),
                          hip_graph_max_shapes=hparams.get("hip_graph_max_shapes", 4),
                          hip_graph_mode=hparams.get("hip_graph_mode", "step"))
        if not hparams["infer"]:
            trainer.fit(cls)
        else:
            trainer.test(cls)

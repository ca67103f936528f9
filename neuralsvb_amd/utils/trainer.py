"""Trainer: the driver loop behind `tasks/run.py` (reference utils/trainer.py:23-520), re-designed for one process per
MI355X with RCCL.

Kept verbatim from the reference contract: task hook order (build_model -> restore -> configure_optimizers ->
train/evaluate), the multi-optimizer step with requires_grad toggling and `None`-loss skipping (:269-342), checkpoint
file names / payload / rotation / atomic save (:397-436), resume (:118-127,347-395), sanity validation, TensorBoard tags.

MI355X-first differences (none changes results):
  * one H2D copy per step (the reference copies the batch once per optimizer pass, :289-290);
  * data-parallel gradient exchange is ONE flat RCCL all-reduce (average) per optimizer pass over that optimizer's
    parameters only (gen 40 MB / disc 3.7 MB / map 0.4 MB) instead of torch-DDP's find_unused_parameters reducer
    that would all-reduce every bucket of every optimizer three times per step; BatchNorm statistics stay per-rank
    (no SyncBN, as in the reference) and rank 0's buffers are broadcast before evaluation / checkpointing;
  * launch: `torchrun --nproc-per-node N` (RANK/LOCAL_RANK/WORLD_SIZE) or, like the reference, self-spawn when
    CUDA_VISIBLE_DEVICES lists several devices;
  * loss terms are only synchronised to the host when they are logged;
  * `hip_graph: true`: the forward+backward of ALL optimizer passes of a step is captured once per (batch shape, phase)
    into ONE multi-stream hipGraph and replayed (`hip_graph_mode: step`, round 4): the weight gradients fork onto their
    side stream inside the capture, the frozen PPG encoder onto its own, and the critic's pass forks off the generator
    pass right after the generator's forward -- it only reads what that forward produced -- and runs beside the
    generator's backward; the streams join before the capture ends.  The host then issues one graph launch plus the
    ~25 launches of clipping / AdamW / repack per step instead of ~650.  Host-side randoms (window starts, speaker
    pick) are drawn up front in the reference's order and read by the graph from device buffers; all-reduce, clipping
    and the optimizer steps stay outside the graph.  (`hip_graph_mode: pass` is the round-2 form: one single-stream
    graph per optimizer pass.)
"""
import copy
import logging
import os
import random
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .ckpt_utils import get_all_ckpts, get_last_checkpoint
from .hparams import hparams


def L_is_emu():
    from .. import _lib
    return _lib._LIB_IS_EMU


def _note_weights_updated(params=None, repack=False):
    from .. import functional as SF          # weight images packed for the kernels are stale after an in-place update
    SF.note_weights_updated(params)
    if repack and params is not None:
        SF.repack_registered(params)          # ... and refilled at once: one multi-tensor launch per optimizer step


def _with_lookahead(it):
    """(item, next item or None) pairs: the Trainer hands the task the batch AFTER the current one (see Trainer._prefetch)."""
    it = iter(it)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


def move_to_device(batch, device):
    """Batch -> device, asynchronously.  A PAGEABLE host tensor would make `.to(device, non_blocking=True)` wait for the stream (a
    host stall in the middle of a step when the look-ahead copy runs between forward and backward), so such a tensor goes through
    a pinned staging copy first (torch's caching host allocator recycles the block once the copy has run)."""
    cuda = torch.device(device).type == "cuda"

    def mv(v):
        if isinstance(v, torch.Tensor):
            if cuda and not v.is_cuda and v.numel() and not v.is_pinned():
                v = v.pin_memory()
            return v.to(device, non_blocking=True)
        if isinstance(v, dict):
            return {k: mv(x) for k, x in v.items()}
        if isinstance(v, list):
            return [mv(x) for x in v]
        return v
    return mv(batch)


class FlatGradSync:
    """Gradients of one optimizer live in one contiguous fp32 buffer (param.grad are views of it): zeroing is a single
    memset and the data-parallel exchange works on contiguous BUCKETS of that buffer.

    Difference from the reference's `optimizer.zero_grad()` (grads -> None): a parameter that owns a slice of the buffer (all
    of them when several processes exchange gradients or a hipGraph needs fixed addresses; otherwise only those a kernel
    accumulates into, see `drop_autograd_grads`) and receives no gradient in a pass
    keeps a ZERO gradient here, so AdamW still applies its step to it (stale momentum decays towards zero, weight decay
    acts).  Results are identical to the reference exactly when every parameter of an optimizer gets a gradient in each of
    its passes -- true for the three optimizers of SVBVAEMleTask in every phase -- or when weight_decay == 0 and the
    parameter never had a gradient before (its moments are zero).

    Overlap with backward (the reference gets it from torch DDP, utils/trainer.py:453-454): the buffer is cut into
    buckets of ~bucket_bytes in parameter order; every gradient that becomes final during `loss.backward()` is announced
    (autograd's post-accumulate hook, or functional.GRAD_READY for gradients the kernels write straight into `.grad`), and
    when a bucket has seen as many announcements as the same pass produced in its first (recording) step, its all-reduce is
    issued at once with async_op=True -- RCCL runs it on its own stream while the rest of backward keeps the compute
    stream busy.  `finish()` (after backward) issues whatever is left, waits and averages.  A pass is identified by a key
    (optimizer, phase, critic plan): the graph is static under a key, so the recorded counts are exact; an announcement
    beyond the recorded count raises instead of silently reducing a half-accumulated bucket."""

    def __init__(self, params, world_size, group=None, bucket_bytes=8 << 20, overlap=True, drop_autograd_grads=False,
                 exchange=None):
        self.params = [p for p in params]
        self.world_size, self.group = world_size, group
        # `exchange`: the collectives run (default: whenever there is more than one rank).  True with world_size == 1 is the
        # `ddp_single_rank` form of the Trainer: one rank goes through the whole exchange path -- buckets announced from inside
        # backward, all_reduce(async_op=True) on the launch stream, waits, averaging by 1 -- which is what a one-GPU box can
        # exercise of RCCL (tests/test_ddp_gloo.py::test_rccl_single_rank_exchange_path_equals_the_plain_step)
        self.exchange = (world_size > 1) if exchange is None else bool(exchange)
        # One process, eager launches: a parameter whose gradient only ever arrives through autograd (norm affines, biases of
        # torch ops, ...) gets `.grad = None` instead of a zeroed view after each optimizer step, so autograd hands it the
        # new gradient tensor as-is rather than launching `grad += new` (~50 launches per step).  Parameters whose gradient
        # a kernel accumulates in place (functional._gbuf marks them) keep their slice of the flat buffer.
        self.drop_autograd_grads = bool(drop_autograd_grads) and not self.exchange
        self._zeroed = 0
        n = sum((p.numel() + 3) // 4 * 4 for p in self.params)        # every view starts 16-byte aligned (vector kernels)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.overlap = bool(overlap) and self.exchange
        self.buckets, self._bucket_of = [], {}
        self._offsets = []               # element offset of every parameter's slice
        off = start = 0
        for p in self.params:
            self._offsets.append(off)
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self._bucket_of[p.grad.data_ptr()] = len(self.buckets)
            off += (p.numel() + 3) // 4 * 4
            if (off - start) * 4 >= bucket_bytes:
                self.buckets.append((start, off))
                start = off
        if off > start or not self.buckets:
            self.buckets.append((start, off))
        self._profiles = {}              # pass key -> announcements per bucket (recorded on the key's first step)
        self._key = self._counts = self._expected = self._work = None
        self._next = -1
        self._comm_stream = None
        self.order = []                  # bucket indices in the order their collectives were issued (last 64; tests read it)
        self.measure_wait, self._wait_events = False, []
        self.stats = {"launched_in_backward": 0, "launched_after": 0, "wait_s": 0.0, "passes": 0}
        if self.overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self.grad_ready)

    def zero(self):
        self.flat.zero_()
        self._zeroed += 1
        if self.drop_autograd_grads and self._zeroed >= 1:
            for p in self.params:
                if not getattr(p, "_svb_sink", False):
                    p.grad = None

    def gather_adopted(self):
        """Before an update that reads the FLAT buffer (utils/flat_optim.py): gradients autograd handed over as tensors of
        their own (`drop_autograd_grads`) are copied into their slices, all of them by one launch.  Returns the indices of the
        parameters that received NO gradient in this pass (`.grad is None`)."""
        if not self.drop_autograd_grads:
            return []
        srcs, offs, skipped = [], [], []
        base, esz = self.flat.data_ptr(), self.flat.element_size()
        for i, (p, off) in enumerate(zip(self.params, self._offsets)):
            g = p.grad
            if g is None:
                skipped.append(i)          # no gradient in this pass: torch.optim.AdamW leaves such a parameter alone
                continue
            if g.data_ptr() == base + off * esz:
                continue
            if g.dtype != torch.float32 or g.numel() != p.numel():
                raise RuntimeError("an adopted gradient does not match its parameter's slice of the flat buffer")
            srcs.append(g if g.is_contiguous() else g.contiguous())
            offs.append(off)
        if srcs:
            from .. import kernels as _K
            _K.gather_segments(srcs, offs, self.flat)
        return skipped

    def pin_views(self):
        """Back to one fixed gradient buffer per parameter (hipGraph capture needs stable addresses)."""
        self.drop_autograd_grads = False
        off = 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
            off += (p.numel() + 3) // 4 * 4

    # ---- overlapped exchange ----------------------------------------------------------------------------------
    def begin_pass(self, key):
        """Call before the backward of an optimizer pass."""
        if not self.overlap:
            return
        self._key = key
        self._counts = [0] * len(self.buckets)
        self._expected = self._profiles.get(key)
        self._work = [None] * len(self.buckets)
        self._next = len(self.buckets) - 1          # buckets are exchanged in ONE order on every rank: last bucket first

    def grad_ready(self, p):
        if self._counts is None:
            return
        b = self._bucket_of.get(p.grad.data_ptr() if p.grad is not None else 0)
        if b is None:
            return
        self._counts[b] += 1
        if self._expected is not None:
            if self._counts[b] > self._expected[b]:
                raise RuntimeError(f"gradient bucket {b} of pass {self._key} received more gradient announcements than in "
                                   f"its recording step: the autograd graph changed under an unchanged pass key")
            # Rank-invariant collective order: the recorded profile (and the pass key behind it) is rank-local state -- two
            # ranks may disagree on it (e.g. a critic window that one rank's longest clip does not reach) -- so completion
            # order must never decide which collective is issued next.  A complete bucket b is exchanged only once every
            # bucket above it has been: every rank issues N-1, N-2, ..., 0, whatever it recorded; ranks that complete a
            # bucket later simply issue the same sequence later (finish() flushes the rest in the same order).
            while self._next >= 0 and self._counts[self._next] >= self._expected[self._next]:
                self._launch(self._next, True)
                self._next -= 1

    def _launch(self, b, in_backward):
        s, e = self.buckets[b]
        if self.flat.is_cuda:
            # The bucket's gradients were written by the compute stream (autograd) and by the weight-gradient side stream
            # (kernels.WGRAD_STREAM): the collective is issued from a launch stream that has caught up with both, so
            # neither of them waits for the other (or for RCCL) here.
            from .. import kernels as _K
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(self.flat.device)
            cs = self._comm_stream
            cs.wait_stream(torch.cuda.current_stream(self.flat.device))
            if _K.WGRAD_STREAM is not None:
                cs.wait_stream(_K.WGRAD_STREAM)
            with torch.cuda.stream(cs):
                self._work[b] = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work[b] = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.stats["launched_in_backward" if in_backward else "launched_after"] += 1
        self.order.append(b)
        if len(self.order) > 64:
            del self.order[:-64]

    def finish(self):
        """After backward: exchange the buckets that did not complete during it (same descending order), wait, average."""
        if not self.exchange:
            return
        if not self.overlap or self._counts is None:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world_size)
            return
        if self._expected is None:
            self._profiles[self._key] = list(self._counts)
            # the recording step of a pass key also tells which parameters the pass does not reach: they keep a ZERO gradient in the
            # flat buffer and AdamW steps them (the reference leaves `.grad = None` and skips them) -- say so once instead of
            # differing silently (never the case for the three optimizers of SVBVAEMleTask)
            n_req = sum(1 for p in self.params if p.requires_grad)
            if sum(self._counts) < n_req:
                import warnings
                warnings.warn(f"{n_req - sum(self._counts)} of {n_req} parameters of this optimizer received no gradient in pass {self._key}: "
                              f"in the flat gradient buffer they keep a zero gradient and the optimizer still steps them (moments decay, "
                              f"weight decay acts), where the reference's zero_grad() -> None would skip them", RuntimeWarning)
        while self._next >= 0:
            self._launch(self._next, False)
            self._next -= 1
        import time as _t
        cuda = self.flat.is_cuda
        if cuda and self.measure_wait:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = _t.perf_counter()
        for w in self._work:
            w.wait()                       # (device tensors: the CURRENT stream waits for RCCL's; the host does not block)
        self.stats["wait_s"] += _t.perf_counter() - t0
        if cuda and self.measure_wait:
            e1.record()
            self._wait_events.append((e0, e1))
        self.stats["passes"] += 1
        self.flat.div_(self.world_size)
        self._counts = self._work = None

    def exposed_wait_ms(self):
        """Device time the compute stream spent waiting for RCCL in finish() since the last call (measure_wait = True)."""
        if not self._wait_events:
            return 0.0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events)
        self._wait_events = []
        return ms

    def all_reduce(self):
        """Blocking exchange of the whole buffer (no overlap): kept for callers outside the Trainer's step."""
        if self.exchange:
            if self._counts is not None:
                return self.finish()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world_size)


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    add_audio = add_figure = add_scalar


def _make_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:
        return _NullWriter()


class Trainer:
    def __init__(self, work_dir, accumulate_grad_batches=1, max_updates=160000, print_nan_grads=False,
                 val_check_interval=2000, num_sanity_val_steps=5, amp=False, tb_log_interval=10, monitor_key="val_loss",
                 monitor_mode="min", num_ckpt_keep=5, save_best=True, resume_from_checkpoint=0, seed=1234, debug=False,
                 hip_graph=False, hip_graph_warmup=2, hip_graph_max_shapes=4, hip_graph_mode="step", **_):
        if work_dir:
            os.makedirs(work_dir, exist_ok=True)
        self.work_dir = work_dir
        self.accumulate_grad_batches = accumulate_grad_batches
        self.max_updates, self.num_sanity_val_steps = max_updates, num_sanity_val_steps
        self.print_nan_grads = print_nan_grads
        self.resume_from_checkpoint = resume_from_checkpoint if resume_from_checkpoint > 0 else None
        self.seed, self.debug, self.amp = seed, debug, amp
        if amp:
            # the reference's `amp` is fp16 autocast + GradScaler (utils/trainer.py:88,288,306-333); the HIP kernels take
            # fp32 tensors only (their reduced-precision mode is `conv_precision: bf16x3`), so autocast would hand them
            # bf16 inputs and fail deep inside a step -- refuse up front instead
            raise NotImplementedError("amp is not supported by the MI355X kernels: use conv_precision=bf16x3 (fp32 storage, "
                                      "bf16 matrix cores with an fp32-class operand split) instead of autocast")
        self.task, self.optimizers, self.grad_sync = None, [], []
        self.flat_optim = []
        self._critic_stream, self._critic_busy = None, False
        self._wgrad_stream = None            # side stream of the weight gradients; kernels.WGRAD_STREAM only while a step runs
        self._grad_enabled_for = None        # (task, optimizer index) whose parameters currently have requires_grad = True
        self._fwd_done = None                # step-graph capture: callable(opt_idx) invoked between a pass's forward and backward
        self._upcoming, self._moved = None, None    # the next step's batch (host side) / its device copy made one step early
        self._param_cache = None
        # hipGraph replay of each optimizer pass's forward+backward (fixed-shape batches; see _graphed_forward_backward)
        self.hip_graph, self.hip_graph_warmup, self.hip_graph_max_shapes = bool(hip_graph), hip_graph_warmup, hip_graph_max_shapes
        self._static, self._graphs, self._graph_pool, self._graph_stream = {}, {}, None, None
        self.hip_graph_mode = hip_graph_mode
        self._static_pins = {}               # (shape signature, batch key) -> pinned staging ring of host-side batch tensors
        self.testing = False
        self.global_step = self.current_epoch = 0
        self.monitor_key, self.num_ckpt_keep, self.save_best = monitor_key, num_ckpt_keep, save_best
        self.monitor_op = np.less if monitor_mode == "min" else np.greater
        self.best_val_results = np.inf if monitor_mode == "min" else -np.inf
        self.val_check_interval, self.tb_log_interval = val_check_interval, tb_log_interval
        # topology: torchrun env wins; otherwise the reference rule (ids listed in CUDA_VISIBLE_DEVICES)
        self.world_size = int(os.environ.get("WORLD_SIZE", "0"))
        self.launched_by_torchrun = self.world_size > 0
        if not self.launched_by_torchrun:
            ids = [x for x in os.environ.get("CUDA_VISIBLE_DEVICES", "").split(",") if x != ""]
            self.world_size = len(ids) if ids else (1 if torch.cuda.is_available() else 0)
        self.on_gpu = torch.cuda.is_available() and self.world_size > 0
        self.proc_rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # `ddp_single_rank: true` (a test / bring-up switch): a one-rank run takes the data-parallel path -- process group, start-up
        # broadcast, bucketed gradient exchange overlapped with backward and the constraints that come with it (immediate
        # weight-gradient reduces, every gradient a slice of the flat buffer) -- so RCCL executes on a one-GPU box
        self.use_ddp = self.world_size > 1 or (self.world_size == 1 and bool(hparams.get("ddp_single_rank", False)))
        self.logger = _NullWriter()

    # ------------------------------------------------------------------ entry points
    def test(self, task_cls):
        self.testing = True
        self.fit(task_cls)

    def fit(self, task_cls):
        if self.use_ddp and not self.launched_by_torchrun:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(random.randint(15000, 30000)))
            mp.spawn(self._spawned, nprocs=self.world_size, args=(task_cls, copy.deepcopy(dict(hparams))))
        else:
            if self.use_ddp:
                self._init_dist(self.proc_rank, self.local_rank)
            self.run_single_process(task_cls())
        return 1

    def _spawned(self, idx, task_cls, hp):
        hparams.update(hp)
        self.proc_rank = self.local_rank = idx
        self._init_dist(idx, idx)
        self.run_single_process(task_cls())

    def _init_dist(self, rank, local_rank):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if self.on_gpu else "gloo"       # "nccl" is RCCL on ROCm (xGMI within the node)
        if self.on_gpu:
            torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            dist.init_process_group(backend, rank=rank, world_size=self.world_size)
        if rank != 0 and not self.debug:
            sys.stdout = open(os.devnull, "w")
            sys.stderr = open(os.devnull, "w")
        random.seed(self.seed)
        np.random.seed(self.seed)     # same window starts / speaker picks on every rank (reference :458-459)

    @property
    def device(self):
        return torch.device("cuda", self.local_rank) if self.on_gpu else torch.device("cpu")

    def get_task_ref(self):
        return self.task

    # ------------------------------------------------------------------ setup
    def setup(self, task):
        """build model -> restore -> device -> optimizers (+ flat grad buffers) -> restore optimizer state."""
        self.task = task
        task.trainer = self
        model = task.build_model()
        if model is not None:
            task.model = model
        checkpoint, _ = get_last_checkpoint(self.work_dir, self.resume_from_checkpoint) if self.work_dir else (None, None)
        if checkpoint is not None:
            self.restore_weights(checkpoint)
        task.to(self.device)
        if not self.testing:
            self.optimizers = task.configure_optimizers()
            self.first_epoch = True
            self.grad_sync = [FlatGradSync([p for g in o.param_groups for p in g["params"]], self.world_size,
                                           bucket_bytes=int(hparams.get("ddp_bucket_mb", 8) * (1 << 20)),
                                           overlap=hparams.get("ddp_overlap", True), exchange=self.use_ddp,
                                           drop_autograd_grads=(not self.hip_graph and hparams.get("drop_autograd_grads", True)))
                              if o is not None else None for o in self.optimizers]
        if checkpoint is not None:
            self.restore_opt_state(checkpoint)
        del checkpoint
        # clipping + AdamW as two launches over flat parameter / gradient / moment buffers (utils/flat_optim.py)
        self.flat_optim = [None] * len(self.optimizers)
        if not self.testing and hparams.get("flat_adamw", True):
            from .flat_optim import FlatAdamW
            for i, (o, gs) in enumerate(zip(self.optimizers, self.grad_sync)):
                if (o is not None and gs is not None and FlatAdamW.eligible(o) and (gs.flat.is_cuda or L_is_emu())):
                    self.flat_optim[i] = FlatAdamW(o, gs)
            if any(f is not None for f in self.flat_optim):
                _note_weights_updated()            # (the parameters moved into the flat buffers: packed images are stale)
        if self.use_ddp:
            self._broadcast_module_state(params=True)
            dist.barrier()
        task.testing = self.testing
        if self.proc_rank == 0 and self.work_dir:
            self.logger = _make_writer(os.path.join(self.work_dir, "lightning_logs", "version_lastest"))
        task.logger = self.logger
        return task

    def run_single_process(self, task):
        self.setup(task)
        try:
            if self.testing:
                self.run_evaluation(test=True)
            else:
                self.train()
        except KeyboardInterrupt:
            task.on_keyboard_interrupt()

    def _broadcast_module_state(self, params=False):
        if not self.use_ddp:
            return
        with torch.no_grad():
            ts = list(self.task.buffers()) + (list(self.task.parameters()) if params else [])
            for t in ts:
                if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                    dist.broadcast(t, 0)
        if params:
            _note_weights_updated()

    # ------------------------------------------------------------------ evaluation
    def run_evaluation(self, test=False):
        res = self.evaluate(self.task, test, tqdm_desc="Valid" if not test else "test")
        if res is not None and "tb_log" in res:
            self.log_metrics_to_tb(res["tb_log"])
        if self.proc_rank == 0 and not test:
            self.save_checkpoint(epoch=self.current_epoch, logs=res)

    def evaluate(self, task, test=False, tqdm_desc="Valid", max_batches=None):
        self._join_critic_stream()
        if max_batches == -1:
            max_batches = None
        self._broadcast_module_state()
        task.zero_grad(set_to_none=False)
        task.eval()
        outputs = []
        with torch.no_grad():
            if test and task.test_start() == "EXIT":
                return None
            loader = task.test_dataloader() if test else task.val_dataloader()
            for batch_idx, batch in enumerate(loader):
                if batch is None:
                    continue
                if max_batches is not None and batch_idx >= max_batches:
                    break
                batch = move_to_device(batch, self.device)
                outputs.append(task.test_step(batch, batch_idx) if test else task.validation_step(batch, batch_idx))
            res = task.test_end(outputs) if test else task.validation_end(outputs)
        task.train()
        return res

    # ------------------------------------------------------------------ training
    def train(self):
        task = self.task
        task.on_train_start()
        if self.num_sanity_val_steps > 0:
            self.evaluate(task, False, "Sanity Val", max_batches=self.num_sanity_val_steps)
        loader = task.train_dataloader()
        epoch = self.current_epoch
        task.train()
        while True:
            task.current_epoch = self.current_epoch = epoch
            task.on_epoch_start()
            for batch_idx, (batch, upcoming) in enumerate(_with_lookahead(loader)):
                _, tb_metrics = self.run_training_batch(batch_idx, batch, next_batch=upcoming)
                if self.global_step % self.val_check_interval == 0 and not self.first_epoch:
                    self.run_evaluation()
                self.first_epoch = False
                if (self.global_step + 1) % self.tb_log_interval == 0:
                    self.log_metrics_to_tb(tb_metrics)
                self.global_step += 1
                task.global_step = self.global_step
                if self.global_step > self.max_updates:
                    print("| Training end..")
                    break
            task.on_epoch_end()
            epoch += 1
            if self.global_step > self.max_updates:
                break
        task.on_train_end()
        self._write_tile_report()

    def _write_tile_report(self):
        """<work_dir>/tile_table_info.json: which tile table the run used and how many conv launch signatures it had to resolve through
        the nearest table entry of their family (every batch shape of a token-budget loader) -- and that none was measured on line."""
        if self.proc_rank != 0 or not self.work_dir:
            return
        try:
            import json
            from .. import kernels as _K
            with open(os.path.join(self.work_dir, "tile_table_info.json"), "w") as f:
                json.dump(dict(_K.tile_table_info(), global_step=int(self.global_step)), f)
        except OSError:
            pass

    def _static_batch(self, batch):
        """hipGraph mode: the batch lives in fixed device buffers (one set per distinct shape signature)."""
        sig = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if isinstance(v, torch.Tensor))
        bufs = self._static.get(sig)
        if bufs is None:
            if len(self._static) >= self.hip_graph_max_shapes:
                return None, None
            bufs = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device) for k, v in batch.items()
                    if isinstance(v, torch.Tensor)}
            self._static[sig] = bufs
        out = dict(batch)
        for k, b in bufs.items():
            src = batch[k]
            if not src.is_cuda and b.is_cuda:
                # host tensors go through a two-deep ring of pinned rows: a pageable source makes the "async" copy wait for the
                # stream -- one full device sync per step, the whole point of replaying a graph gone
                ring = self._static_pins.get((sig, k))
                if ring is None:
                    ring = self._static_pins[(sig, k)] = [[torch.empty(src.shape, dtype=src.dtype).pin_memory(), None]
                                                          for _ in range(2)] + [0]
                slot = ring[ring[2] % 2]
                ring[2] += 1
                if slot[1] is not None:
                    slot[1].synchronize()
                slot[0].copy_(src)
                b.copy_(slot[0], non_blocking=True)
                slot[1] = torch.cuda.Event()
                slot[1].record()
            else:
                b.copy_(src, non_blocking=True)
            out[k] = b
        return sig, out

    def _forward_backward(self, batch, batch_idx, opt_idx):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(self.amp) and self.on_gpu):
            out = self.task.training_step(batch, batch_idx, opt_idx)
        if self._fwd_done is not None:       # step-graph capture: where an independent later pass may fork off this one
            self._fwd_done(opt_idx)
        loss = out["loss"]
        if loss is not None:
            loss = loss / self.accumulate_grad_batches
            if loss.requires_grad:
                from .. import functional as _SF
                from .. import kernels as _K
                # one process, eager launches: the stage-2 reduces of this pass's weight gradients are batched -- partials collect
                # in a 48 MB arena per stream and one multi-tensor launch finishes them when it is full and at the end of the
                # pass (with a gradient exchange they must be final when they are announced: immediate reduces there)
                defer = (hparams.get("defer_wgrad_reduce", True) and _SF.GRAD_READY is None and not _SF.CAPTURING
                         and not self.use_ddp and not (self.hip_graph and self.on_gpu))
                if defer:
                    _K.begin_deferred_reduces()
                try:
                    loss.backward()
                finally:
                    if defer:
                        _K.flush_deferred_reduces()
                if _K.WGRAD_STREAM is not None:          # weight gradients enqueued beside the data-gradient chain
                    torch.cuda.current_stream().wait_stream(_K.WGRAD_STREAM)
        return out

    def _graphed_forward_backward(self, sig, batch, batch_idx, opt_idx):
        """Forward + backward of one optimizer pass as a captured hipGraph (captured after `hip_graph_warmup` eager
        runs of the same key -- kernel autotuning and lazy initialisation happen there -- then replayed).  The
        optimizer step, gradient clipping and the data-parallel all-reduce stay outside the graph."""
        key = (sig, opt_idx, self.task.graph_key(self.global_step), self.task.training)
        ent = self._graphs.setdefault(key, {"seen": 0, "graph": None, "out": None})
        if ent["graph"] is not None:
            ent["graph"].replay()
            return ent["out"]
        if ent["seen"] < max(self.hip_graph_warmup, 1) or ent.get("idle"):
            ent["seen"] += 1
            out = self._forward_backward(batch, batch_idx, opt_idx)
            ent["idle"] = out["loss"] is None          # this optimizer has nothing to do in this phase: nothing to capture
            return out
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        from .. import functional as SF
        SF.CAPTURING = True
        try:
            with torch.cuda.graph(graph, pool=self._graph_pool, stream=self._graph_stream):
                out = self._forward_backward(batch, batch_idx, opt_idx)
        finally:
            SF.CAPTURING = False
        if self._graph_pool is None:
            self._graph_pool = graph.pool()
        ent["graph"], ent["out"] = graph, out
        graph.replay()                      # capture only records: this replay is the step's actual execution
        return out

    def _prefetch(self, opt_idx):
        """Between the forward and the backward of the step's first pass: copy the NEXT batch to the device and let the task
        start whatever depends on that batch alone (SVBVAEMleTask.prefetch: the frozen PPG encoder, on its side stream)."""
        nxt, self._upcoming = self._upcoming, None
        self._fwd_done = None
        if nxt is None:
            return
        moved = move_to_device(nxt, self.device)
        self._moved = (nxt, moved)
        self.task.prefetch(moved)

    def _graphed_step(self, task, batch, batch_idx, sig, multi, pbar, tb):
        """`hip_graph_mode: step`: forward + backward of every optimizer pass of the step as ONE captured multi-stream hipGraph
        (after `hip_graph_warmup` eager steps of the same key -- kernel autotuning and lazy initialisation happen there),
        then per pass what stays outside: gradient clipping, the optimizer step, the weight repack, zeroing.

        Streams inside the capture: weight gradients fork onto kernels.WGRAD_STREAM and the frozen PPG encoder onto its own
        stream exactly as in eager mode (each joins where its results are read); the pass `task.independent_critic_pass`
        reads only what the generator's FORWARD produced (the kept generated mels, the batch, the critic's weights -- which
        the generator's backward reads but never writes), so it forks off an event recorded between the generator's forward
        and backward and runs beside that backward on the critic stream; everything joins before the capture ends.  The
        passes' optimizer steps commute with this order: the critic pass does not read a generator weight (the reference
        runs G-step, then D-pass on the kept `model_out`: utils/trainer.py:269-342, svb_vae_task.py:610-640)."""
        from .. import functional as SF
        from .. import kernels as _K
        key = (sig, task.graph_key(self.global_step) if hasattr(task, "graph_key") else None, task.training)
        ent = self._graphs.setdefault(key, {"seen": 0, "graph": None, "outs": None})
        if ent["graph"] is None and ent["seen"] < max(self.hip_graph_warmup, 1):
            ent["seen"] += 1
            for opt_idx, optimizer in enumerate(self.optimizers):
                if optimizer is not None:
                    self._optimizer_pass(task, batch, batch_idx, opt_idx, optimizer, multi, False, sig, pbar, tb)
            return
        if ent["graph"] is None:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            disc_idx = getattr(task, "independent_critic_pass", None)
            fork = disc_idx is not None and hparams.get("overlap_critic_pass", True)
            if fork and self._critic_stream is None:
                self._critic_stream = torch.cuda.Stream(self.device)
            outs, forked, fwd_event = {}, False, {}

            def fwd_done(opt_idx):
                if fork and opt_idx != disc_idx:
                    ev = torch.cuda.Event()
                    ev.record()
                    fwd_event["ev"] = ev
            SF.CAPTURING = True
            self._fwd_done = fwd_done
            try:
                with torch.cuda.graph(graph, pool=self._graph_pool, stream=self._graph_stream):
                    main = torch.cuda.current_stream(self.device)
                    for opt_idx, optimizer in enumerate(self.optimizers):
                        if optimizer is None:
                            continue
                        self._enable_grads_for(task, opt_idx, multi)
                        if fork and opt_idx == disc_idx and "ev" in fwd_event:
                            s2 = self._critic_stream
                            s2.wait_event(fwd_event["ev"])
                            side, _K.WGRAD_STREAM = _K.WGRAD_STREAM, None     # (its few weight gradients stay on its own stream)
                            try:
                                with torch.cuda.stream(s2):
                                    out = self._forward_backward(batch, batch_idx, opt_idx)
                            finally:
                                _K.WGRAD_STREAM = side
                            forked = True
                        else:
                            out = self._forward_backward(batch, batch_idx, opt_idx)
                        outs[opt_idx] = out
                    if forked:
                        main.wait_stream(self._critic_stream)
            finally:
                SF.CAPTURING = False
                self._fwd_done = None
            if self._graph_pool is None:
                self._graph_pool = graph.pool()
            ent["graph"], ent["outs"] = graph, outs
        ent["graph"].replay()                   # (capture only records: the replay is the step's execution)
        for opt_idx, optimizer in enumerate(self.optimizers):
            if optimizer is not None and opt_idx in ent["outs"]:
                self._finish_pass(task, batch_idx, opt_idx, optimizer, ent["outs"][opt_idx], pbar, tb)

    def run_training_batch(self, batch_idx, batch, next_batch=None):
        """One step = every non-None optimizer in order (reference :269-342).  next_batch (optional): the batch of the following
        step -- work of that step which depends on nothing but its batch (the frozen PPG encoder) is issued one step early."""
        if batch is None:
            return {}, {}
        self._upcoming = next_batch
        if self.hip_graph and self.on_gpu:
            # hipGraph mode works on ONE side stream throughout (eager warm-up runs, capture, replay, optimizer): the
            # autograd AccumulateGrad nodes must live on the stream that is captured, never on the default stream.
            if self._graph_stream is None:
                self._graph_stream = torch.cuda.Stream(self.device)
            ambient = torch.cuda.current_stream(self.device)
            self._graph_stream.wait_stream(ambient)        # batch tensors / evaluation / checkpoint reads issued there
            with torch.cuda.stream(self._graph_stream):
                ret = self._run_training_batch(batch_idx, batch)
            ambient.wait_stream(self._graph_stream)        # whoever reads losses or weights next does so on that stream
            return ret
        return self._run_training_batch(batch_idx, batch)

    def _run_training_batch(self, batch_idx, batch):
        from .. import functional as SF
        from .. import kernels as _K
        SF.begin_weight_epoch()               # packed weight images may be reused inside this step (until the next update)
        # The side stream of the weight gradients is routing state of THIS step only: whoever calls the kernels after the
        # step returns (validation, user code, the next test) must get their results on the stream they launch from.
        prev_side = _K.WGRAD_STREAM
        graph_mode = self.hip_graph and self.on_gpu
        step_graph = graph_mode and self.hip_graph_mode == "step" and not self.use_ddp and self.accumulate_grad_batches == 1
        want_side = self.on_gpu and (not graph_mode or step_graph) and hparams.get("wgrad_side_stream", True)
        if want_side and self._wgrad_stream is None:
            # (round 6 tried a CU-masked side stream -- hipExtStreamCreateWithCUMask on 50 / 75 / 87.5 % of every XCD's CUs, so that the
            #  256-register weight-gradient workgroups cannot fill every SIMD's register file: 21.2 ms/step at every fraction against
            #  12.3 with an ordinary stream, profiles/r06_ab_cu_mask.log)
            self._wgrad_stream = torch.cuda.Stream(self.device)
        _K.WGRAD_STREAM = self._wgrad_stream if want_side else None
        done = False
        try:
            ret = self._run_training_batch_body(batch_idx, batch)
            done = True
            return ret
        finally:
            side, _K.WGRAD_STREAM = _K.WGRAD_STREAM, prev_side
            # Every pass joins the side stream right after its own backward, on the stream that pass runs on; only a pass that
            # raised may not have.  (Round 4: an unconditional join here made the COMPUTE stream wait for the weight gradients
            # of the critic's pass -- issued from the critic's stream onto this side stream -- i.e. for most of the critic pass
            # that is supposed to overlap the next step: 2.3 ms of idle compute stream per step, tools/gpu_split.py.)
            if side is not None and (not done or hparams.get("step_end_side_join", False)):
                torch.cuda.current_stream(self.device).wait_stream(side)
            if not (self.hip_graph and self.on_gpu):
                self._fwd_done = None
            _K.abort_deferred_reduces()       # (a pass that raised inside backward leaves no recorded reduce behind)
            SF.GRAD_READY = None
            SF.end_weight_epoch()             # outside a managed step trainable weights are never served from the cache

    def _run_training_batch_body(self, batch_idx, batch):
        task = self.task
        graph_mode = self.hip_graph and self.on_gpu
        if graph_mode:
            for g in self.grad_sync:
                if g is not None and g.drop_autograd_grads:
                    g.pin_views()
        if hasattr(task, "begin_step"):
            task.begin_step(batch, self.global_step, stage_device=graph_mode)   # host randoms (batch still on the host)
        sig = None
        if graph_mode:
            sig, sbatch = self._static_batch(batch)
            graph_mode = sig is not None
            batch = sbatch if graph_mode else batch
        if not graph_mode:
            if self._moved is not None and self._moved[0] is batch:
                batch = self._moved[1]                            # copied to the device during the previous step (_prefetch)
            else:
                batch = move_to_device(batch, self.device)        # once per step
            self._moved = None
            if (self.on_gpu and self._upcoming is not None and hasattr(task, "prefetch")
                    and hparams.get("prefetch_next_batch", True)):
                self._fwd_done = self._prefetch
        pbar, tb = {}, {}
        multi = len(self.optimizers) > 1
        if graph_mode and self.hip_graph_mode == "step" and not self.use_ddp and self.accumulate_grad_batches == 1:
            task.critic_barrier = None
            self._graphed_step(task, batch, batch_idx, sig, multi, pbar, tb)
            if hasattr(task, "end_step"):
                task.end_step()
            return pbar, tb
        # The critic's own optimizer pass depends on nothing the generator pass of the NEXT step does before it calls the
        # critic (it reads the generated mels kept from this step's forward, the batch and the critic; it writes the critic):
        # it runs on a second stream that starts once everything issued so far is done, and the compute stream only waits
        # for it where the generator pass first touches critic state (task.critic_barrier).  One process, eager launches.
        disc_idx = getattr(task, "independent_critic_pass", None)
        overlap_disc = (disc_idx is not None and self.on_gpu and not graph_mode
                        and self.accumulate_grad_batches == 1 and hparams.get("overlap_critic_pass", True))
        if overlap_disc and self._critic_stream is None:
            self._critic_stream = torch.cuda.Stream(self.device)
        task.critic_barrier = self._join_critic_stream if overlap_disc else None
        for opt_idx, optimizer in enumerate(self.optimizers):
            if optimizer is None:
                continue
            if overlap_disc and opt_idx == disc_idx:
                s2 = self._critic_stream
                s2.wait_stream(torch.cuda.current_stream(self.device))
                for v in batch.values():
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(s2)
                with torch.cuda.stream(s2):
                    self._optimizer_pass(task, batch, batch_idx, opt_idx, optimizer, multi, graph_mode, sig, pbar, tb)
                self._critic_busy = True
                continue
            self._optimizer_pass(task, batch, batch_idx, opt_idx, optimizer, multi, graph_mode, sig, pbar, tb)
        if hasattr(task, "end_step"):
            task.end_step()
        return pbar, tb

    def _join_critic_stream(self):
        """The current stream waits for the critic pass that was enqueued on its own stream (no-op from that stream itself)."""
        if self._critic_stream is not None and self._critic_busy:
            cur = torch.cuda.current_stream(self.device)
            if cur != self._critic_stream:
                cur.wait_stream(self._critic_stream)
                self._critic_busy = False

    def _optimizer_pass(self, task, batch, batch_idx, opt_idx, optimizer, multi, graph_mode, sig, pbar, tb):
        """One optimizer of the step: forward + backward of its pass, gradient exchange, clipping, update."""
        self._enable_grads_for(task, opt_idx, multi)
        sync = self.grad_sync[opt_idx] if opt_idx < len(self.grad_sync) else None
        final_micro = (self.global_step + 1) % self.accumulate_grad_batches == 0
        if sync is not None and not graph_mode and final_micro:
            # overlapped exchange: buckets are all-reduced as their gradients become final during backward (only on the
            # micro-batch that is followed by the optimizer step: earlier ones just accumulate locally)
            from .. import functional as SF
            key = (opt_idx, task.graph_key(self.global_step) if hasattr(task, "graph_key") else None)
            sync.begin_pass(key)
            SF.GRAD_READY = sync.grad_ready if sync.overlap else None
        try:
            if graph_mode:
                out = self._graphed_forward_backward(sig, batch, batch_idx, opt_idx)
            else:
                out = self._forward_backward(batch, batch_idx, opt_idx)
        finally:
            if sync is not None and not graph_mode and final_micro:
                SF.GRAD_READY = None
        self._finish_pass(task, batch_idx, opt_idx, optimizer, out, pbar, tb)

    def _enable_grads_for(self, task, opt_idx, multi):
        if multi:   # only this optimizer's parameters receive gradients in this pass (:280-285)
            # (the parameter lists are cached: walking task.parameters() -- a recursive named_modules() generator --
            # three times per step cost 2 ms of a host-bound 22 ms step)
            if self._param_cache is None or self._param_cache[0] is not task:
                self._param_cache = (task, list(task.parameters()),
                                     [[p for g in o.param_groups for p in g["params"]] if o is not None else []
                                      for o in self.optimizers])
            # (only what changes: the previous pass's parameters off, this pass's on -- not all ~400 twice per pass)
            prev = self._grad_enabled_for
            fresh = prev is None or prev[0] is not task
            if fresh:
                for p in self._param_cache[1]:
                    p.requires_grad = False
            elif prev[1] != opt_idx:
                for p in self._param_cache[2][prev[1]]:
                    p.requires_grad = False
            if fresh or prev[1] != opt_idx:
                for p in self._param_cache[2][opt_idx]:
                    p.requires_grad = True
                self._grad_enabled_for = (task, opt_idx)

    def _finish_pass(self, task, batch_idx, opt_idx, optimizer, out, pbar, tb):
        """After a pass's backward: logging values, gradient exchange, clipping, update, repack, zero."""
        sync = self.grad_sync[opt_idx] if opt_idx < len(self.grad_sync) else None
        if out["loss"] is None:
            if sync is not None:
                sync._counts = None          # nothing to exchange in this pass
            return
        pbar.update(out["progress_bar"])
        tb.update(out["tb_log"])
        if self.print_nan_grads:
            bad = [n for n, p in task.named_parameters() if p.grad is not None and torch.isnan(p.grad).any()]
            if bad:
                print("| NaN grads: ", bad)
                sys.exit(0)
        if (self.global_step + 1) % self.accumulate_grad_batches == 0:
            sync = self.grad_sync[opt_idx]
            sync.finish()
            task.on_before_optimization(opt_idx)
            flat = self.flat_optim[opt_idx] if opt_idx < len(self.flat_optim) else None
            if flat is not None:
                flat.step(sync.gather_adopted())
            else:
                optimizer.step()
            _note_weights_updated(self._param_cache[2][opt_idx] if self._param_cache is not None else
                                  [p for g in optimizer.param_groups for p in g["params"]], repack=True)
            sync.zero()
            task.on_after_optimization(self.current_epoch, batch_idx, optimizer, opt_idx)

    # ------------------------------------------------------------------ checkpoints (reference :347-436)
    def restore_weights(self, checkpoint):
        task = self.task
        sd = checkpoint["state_dict"]
        if any("." in k for k in sd.keys()):
            task.load_state_dict(sd)
        else:
            for k, v in sd.items():
                getattr(task, k).load_state_dict(v)
        _note_weights_updated()
        self.best_val_results = checkpoint["checkpoint_callback_best"]
        self.global_step = checkpoint["global_step"]
        self.current_epoch = checkpoint["epoch"]
        task.global_step = self.global_step

    def restore_opt_state(self, checkpoint):
        if self.testing:
            return
        states = iter(checkpoint["optimizer_states"])
        for optimizer in self.optimizers:
            if optimizer is None:
                continue
            try:
                optimizer.load_state_dict(next(states))
            except (ValueError, StopIteration):
                print("| WARMING: optimizer parameters not match !!!")
        for f in self.flat_optim:
            if f is not None:
                f.reattach()

    def dump_checkpoint(self):
        for f in self.flat_optim:
            if f is not None:
                f.export_state()
        return {"epoch": self.current_epoch, "global_step": self.global_step,
                "checkpoint_callback_best": self.best_val_results,
                "optimizer_states": [o.state_dict() for o in self.optimizers if o is not None],
                "state_dict": {k: v.state_dict() for k, v in self.task.named_children()
                               if len(list(v.parameters())) > 0}}

    def _atomic_save(self, filepath):
        tmp = str(filepath) + ".part"
        torch.save(self.dump_checkpoint(), tmp, _use_new_zipfile_serialization=False)
        os.replace(tmp, filepath)

    def save_checkpoint(self, epoch, logs=None):
        self._join_critic_stream()
        path = f"{self.work_dir}/model_ckpt_steps_{self.global_step}.ckpt"
        logging.info(f"Epoch {epoch:05d}@{self.global_step}: saving model to {path}")
        self._atomic_save(path)
        for old in get_all_ckpts(self.work_dir)[self.num_ckpt_keep:]:
            os.remove(old)
            logging.info(f"Delete ckpt: {os.path.basename(old)}")
        current = logs.get(self.monitor_key) if logs else None
        if current is not None and self.save_best and self.monitor_op(current, self.best_val_results):
            self.best_val_results = current
            self._atomic_save(f"{self.work_dir}/model_ckpt_best.pt")

    # ------------------------------------------------------------------ logging
    def log_metrics_to_tb(self, metrics, step=None):
        self._join_critic_stream()
        metrics = dict(metrics)
        metrics["epoch"] = self.current_epoch
        step = self.global_step if step is None else step
        if self.proc_rank == 0:
            for k, v in metrics.items():
                if isinstance(v, torch.Tensor):
                    v = v.item()
                if isinstance(v, (int, float)):
                    self.logger.add_scalar(k, v, step)

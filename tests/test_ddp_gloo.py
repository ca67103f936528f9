"""Data-parallel path on CPU: world_size 2, gloo backend (the MI355X run uses the same code with backend "nccl" = RCCL).

Checks (a) the reference's batch sharding rule (tasks/tts/tts.py:93-96: rank r takes batch[r::world], indivisible
batches dropped, max_sentences scaled by world), (b) FlatGradSync: one flat all-reduce per optimizer makes every rank
hold the full-batch gradient, through the HIP autograd functions (emulator build on CPU), and (c) rank 0 -> all
broadcast of parameters/buffers at start-up.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, emu_lib, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, ROOT)
    from neuralsvb_amd import _lib
    _lib._LIB, _lib._LIB_IS_EMU = _lib.bind(emu_lib), True     # test harness: CPU lane emulator
    from neuralsvb_amd import functional as SF
    from neuralsvb_amd.utils.trainer import FlatGradSync, Trainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                                 # different init per rank on purpose
    conv_v = torch.nn.Parameter(torch.randn(6, 4, 3) * 0.3)
    conv_g = torch.nn.Parameter(torch.rand(6, 1, 1) + 0.5)
    bias = torch.nn.Parameter(torch.randn(6) * 0.1)
    params = [conv_v, conv_g, bias]
    # (c) start-up broadcast as Trainer._broadcast_module_state does
    for p in params:
        dist.broadcast(p.data, 0)
    sync = FlatGradSync(params, world)
    g = torch.Generator().manual_seed(7)
    x_full = torch.randn(4, 4, 20, generator=g)                   # global batch of 4 clips
    dy_full = torch.randn(4, 6, 20, generator=g)
    xs, dys = x_full[rank::world], dy_full[rank::world]          # (a) rank r takes batch[r::world]
    y = SF.conv1d(xs, conv_v, bias, 1, 1, weight_g=conv_g)
    loss = (y * dys).sum() / xs.shape[0]                          # per-rank mean over its clips
    loss.backward()
    sync.all_reduce()
    res = {"grad": torch.cat([p.grad.flatten() for p in params]).numpy(), "w": torch.cat([p.detach().flatten() for p in params]).numpy()}
    if rank == 0:
        # single-process reference on the whole batch with stock torch ops
        v, gn, b = (p.detach().clone().requires_grad_(True) for p in params)
        w = v * (gn / v.flatten(1).norm(dim=1).view(-1, 1, 1))
        yr = torch.nn.functional.conv1d(x_full, w, b, 1, 1)
        ((yr * dy_full).sum() / 4).backward()
        res["ref"] = torch.cat([t.grad.flatten() for t in (v, gn, b)]).numpy()
    # (d) bucketed exchange overlapped with backward: three passes under one key; the first records how many gradient
    # announcements each bucket receives, from the second on buckets are all-reduced from inside backward
    lin = torch.nn.Linear(6, 3)                                   # stock-autograd parameters (post-accumulate hook path)
    for p in lin.parameters():
        dist.broadcast(p.data, 0)
    params2 = [conv_v, conv_g, bias] + list(lin.parameters())
    for p in params2:
        p.grad = None
    sync2 = FlatGradSync(params2, world, bucket_bytes=64, overlap=True)          # tiny buckets -> several of them
    res["n_buckets"] = len(sync2.buckets)
    res["ov_grad"], res["ov_ref"] = [], []
    for it in range(3):
        gi = torch.Generator().manual_seed(50 + it)
        xf, df = torch.randn(4, 4, 20, generator=gi), torch.randn(4, 3, generator=gi)
        sync2.zero()
        sync2.begin_pass(("pass", 0))
        SF.GRAD_READY = sync2.grad_ready
        y = SF.conv1d(xf[rank::world], conv_v, bias, 1, 1, weight_g=conv_g)      # gradients written straight into .grad
        loss = (lin(y.mean(-1)) * df[rank::world]).sum() / 2
        loss.backward()
        SF.GRAD_READY = None
        sync2.finish()
        res["ov_grad"].append(torch.cat([p.grad.flatten() for p in params2]).numpy().copy())
        if rank == 0:
            ps = [p.detach().clone().requires_grad_(True) for p in params2]
            w = ps[0] * (ps[1] / ps[0].flatten(1).norm(dim=1).view(-1, 1, 1))
            yr = torch.nn.functional.conv1d(xf, w, ps[2], 1, 1)
            ((torch.nn.functional.linear(yr.mean(-1), ps[3], ps[4]) * df).sum() / 4).backward()
            res["ov_ref"].append(torch.cat([t.grad.flatten() for t in ps]).numpy())
    res["stats"] = dict(sync2.stats)
    # (e) rank-divergent pass keys (the key holds rank-local state, e.g. which critic windows the rank's longest clip
    # reaches): rank 0 replays a recorded profile (buckets go out from inside backward), rank 1 sees its key for the first
    # time (everything goes out in finish()).  The collectives must still pair up: one bucket order on every rank.
    gi = torch.Generator().manual_seed(77)
    xf, df = torch.randn(4, 4, 20, generator=gi), torch.randn(4, 3, generator=gi)
    sync2.zero()
    sync2.order.clear()
    sync2.begin_pass(("pass", 0) if rank == 0 else ("pass", "first seen on this rank"))
    SF.GRAD_READY = sync2.grad_ready
    y = SF.conv1d(xf[rank::world], conv_v, bias, 1, 1, weight_g=conv_g)
    ((lin(y.mean(-1)) * df[rank::world]).sum() / 2).backward()
    SF.GRAD_READY = None
    res["div_in_backward"] = len(sync2.order)
    sync2.finish()
    res["div_order"] = list(sync2.order)
    res["div_grad"] = torch.cat([p.grad.flatten() for p in params2]).numpy().copy()
    if rank == 0:
        ps = [p.detach().clone().requires_grad_(True) for p in params2]
        w = ps[0] * (ps[1] / ps[0].flatten(1).norm(dim=1).view(-1, 1, 1))
        yr = torch.nn.functional.conv1d(xf, w, ps[2], 1, 1)
        ((torch.nn.functional.linear(yr.mean(-1), ps[3], ps[4]) * df).sum() / 4).backward()
        res["div_ref"] = torch.cat([t.grad.flatten() for t in ps]).numpy()
    np.save(os.path.join(out, f"r{rank}.npy"), res, allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_two_ranks_gloo(tmp_path, _emu_lib):
    from tests.conftest import EMU_LIB
    port = _free_port()
    mp.spawn(_worker, args=(2, port, EMU_LIB, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npy", allow_pickle=True).item()
    r1 = np.load(tmp_path / "r1.npy", allow_pickle=True).item()
    assert np.array_equal(r0["w"], r1["w"])                       # identical replicas after broadcast
    assert np.array_equal(r0["grad"], r1["grad"])                 # identical averaged gradients on every rank
    assert np.abs(r0["grad"] - r0["ref"]).max() < 2e-5 * max(1.0, np.abs(r0["ref"]).max())
    # bucketed + overlapped exchange == full-batch gradient on every rank, every pass
    assert r0["n_buckets"] >= 3
    for it in range(3):
        assert np.array_equal(r0["ov_grad"][it], r1["ov_grad"][it])
        assert np.abs(r0["ov_grad"][it] - r0["ov_ref"][it]).max() < 2e-5 * max(1.0, np.abs(r0["ov_ref"][it]).max()), it
    st = r0["stats"]
    assert st["passes"] == 3 and st["launched_after"] == r0["n_buckets"]           # recording pass: nothing overlapped
    assert st["launched_in_backward"] == 2 * r0["n_buckets"]                       # passes 2 and 3: every bucket overlapped
    # rank-divergent keys: same (descending) bucket order on both ranks although one overlapped and the other did not
    nb = r0["n_buckets"]
    assert r0["div_order"] == r1["div_order"] == list(range(nb - 1, -1, -1))
    assert r0["div_in_backward"] > 0 and r1["div_in_backward"] == 0
    assert np.array_equal(r0["div_grad"], r1["div_grad"])
    assert np.abs(r0["div_grad"] - r0["div_ref"]).max() < 2e-5 * max(1.0, np.abs(r0["div_ref"]).max())


def _unreached_worker(port, out):
    import warnings
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from neuralsvb_amd.utils.trainer import FlatGradSync
    a, b, c = (torch.nn.Parameter(torch.randn(8, 3)) for _ in range(3))
    gs = FlatGradSync([a, b, c], 1, bucket_bytes=64, overlap=True, exchange=True)
    msgs = []
    for step in range(2):
        gs.begin_pass(("opt0", "phase"))
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            (a.sum() * 2 + (b * b).sum()).backward()          # c is never reached
            gs.finish()
        msgs.append([str(x.message) for x in w if issubclass(x.category, RuntimeWarning)])
    torch.save({"msgs": msgs, "ga": a.grad.clone(), "gc": c.grad.clone()}, out)
    dist.destroy_process_group()


def test_pass_that_leaves_a_parameter_without_gradient_says_so(tmp_path):
    """The flat gradient buffer keeps a ZERO gradient for a parameter a pass does not reach (the reference's zero_grad() -> None would
    make the optimizer skip it): the recording step of the pass key says so once -- a RuntimeWarning naming the count -- instead of
    differing silently; the exchange itself is unaffected."""
    out = str(tmp_path / "r.pt")
    ctx = mp.get_context("spawn")
    pr = ctx.Process(target=_unreached_worker, args=(_free_port(), out))
    pr.start()
    pr.join(120)
    assert pr.exitcode == 0
    r = torch.load(out)
    assert len(r["msgs"][0]) == 1 and "1 of 3 parameters" in r["msgs"][0][0] and r["msgs"][1] == []
    assert torch.equal(r["ga"], torch.full((8, 3), 4.0)) and float(r["gc"].abs().max()) == 0.0     # (two steps accumulated: no zeroing here)


def test_batch_sharding_rule():
    """build_dataloader: global batches of max_sentences*world clips, rank r takes every world-th item."""
    from neuralsvb_amd.utils.batching import batch_by_size
    sizes = [100, 90, 120, 80, 110, 95, 70, 130, 60]
    world, max_sentences = 2, 2
    batches = batch_by_size(list(range(len(sizes))), lambda i: sizes[i], max_tokens=10 ** 6,
                            max_sentences=max_sentences * world, required_batch_size_multiple=world)
    assert all(len(b) <= max_sentences * world for b in batches) and batches[0] == [0, 1, 2, 3]
    kept = [b for b in batches if len(b) % world == 0]
    shards = [[b[r::world] for b in kept] for r in range(world)]
    assert shards[0][0] == [0, 2] and shards[1][0] == [1, 3]
    assert all(len(a) == len(b) for a, b in zip(*shards))
    flat = sorted(i for r in range(world) for b in shards[r] for i in b)
    assert flat == sorted(i for b in kept for i in b)             # disjoint cover of the kept batches


def test_bench_command_line_starts_its_own_ranks(tmp_path, _emu_lib):
    """`python bench.py --gpus 2` with no launcher around it (the driver's form of the command, as it runs `--gpus 1`): bench.py
    re-execs itself under torch.distributed.run with one rank per device (reference utils/trainer.py:453-466 spawns its own
    ranks too), the ranks run the N > 1 step over gloo on the lane emulator, and rank 0 prints ONE JSON line whose
    `comm.ranks` is 2, whose value counts both ranks' clips, and whose `comm.replica_digests` show bit-identical replicas."""
    import json
    import subprocess
    import sys
    from tests.conftest import EMU_LIB
    small = ("hidden_size=32,fvae_enc_dec_hidden=32,latent_size=16,fvae_enc_n_layers=2,fvae_dec_n_layers=2,"
             "mel_disc_hidden_size=16,warmup_updates=4")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SVB_BENCH_EMU_LIB=EMU_LIB, SVB_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2",
                        "--seconds", "0.71", "--precision", "fp32", "--no-cpu-baseline", "--no-extra-workloads",
                        "--extra-hparams", small], env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["comm"]["ranks"] == 2 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 4 and res["config"]["parallelism"] == "dp2"
    assert abs(res["value"] - 2 * 2 * 0.71 / (res["ms_per_step"] * 1e-3)) < 1e-6 * res["value"]
    assert res["comm"]["buckets_launched_in_backward"] > 0
    # (this test also stands for the former test_bench_multi_rank_path_two_ranks_gloo, which ran the same two steps through workers
    #  it spawned itself: per-rank batch = --batch, DDP start-up broadcast, bucketed exchange overlapped with backward, and the
    #  replicas bit-identical afterwards -- now read from the line's `comm.replica_digests`)
    assert res["comm"]["replicas_identical"] and len(res["comm"]["replica_digests"]) == 2
    assert all(np.isfinite(v) for d_ in res["comm"]["replica_digests"] for v in d_)


@pytest.mark.gpu
def test_bench_command_line_two_ranks_on_one_gpu(tmp_path, gpu_only):
    """The command the driver runs for the scaling curve, end to end on hardware: `python bench.py --gpus 2` starts its two ranks,
    both on the one MI355X of the box (LOCAL_RANK is taken modulo the device count; gloo carries the exchange because RCCL refuses
    two ranks per device), runs warm-up + settle + timed steps with the bucketed exchange inside backward, and rank 0 prints ONE JSON
    line with `comm.ranks == 2` AND its `roofline` (the profiled steps run on rank 0 alone with the exchange switched off, so no
    rank can be left waiting in a collective)."""
    import json
    import subprocess
    import sys
    small = ("hidden_size=32,fvae_enc_dec_hidden=32,latent_size=16,fvae_enc_n_layers=2,fvae_dec_n_layers=2,"
             "mel_disc_hidden_size=16,warmup_updates=4")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SVB_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "2",
                        "--seconds", "0.71", "--no-cpu-baseline", "--no-extra-workloads", "--extra-hparams", small],
                       env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["comm"]["ranks"] == 2 and res["config"]["global_batch"] == 4
    assert res["comm"]["buckets_launched_in_backward"] > 0
    assert res["roofline"] is not None and res["roofline"]["frac"] > 0 and res["roofline"]["serial_streams"]["frac"] > 0
    assert res["warmup_settle"]["extra_steps"] >= 10 and res["tile_table"]["entries"] > 300
    assert res["comm"]["replicas_identical"]


def _gpu_pair_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import argparse
    import sys
    import tempfile
    sys.path.insert(0, ROOT)
    import bench
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = argparse.Namespace(batch=2, seconds=0.71, sample_rate=24000, bf16=False, precision="bf16x3", graph=False)
    small = (",hidden_size=32,fvae_enc_dec_hidden=32,latent_size=16,fvae_enc_n_layers=2,fvae_dec_n_layers=2,"
             "mel_disc_hidden_size=16,warmup_updates=4,ddp_bucket_mb=0.02")
    from neuralsvb_amd import kernels as K
    seen = {"side": 0, "critic": 0}
    with tempfile.TemporaryDirectory() as tmp:
        task, trainer, batch, hp = bench.build_task(args, rank, world, dev, tmp, extra_hparams=small)
        assert trainer.use_ddp and trainer.world_size == world and trainer.on_gpu
        orig = trainer._optimizer_pass

        def spy(*a, **k):
            seen["side"] += int(K.WGRAD_STREAM is not None)
            seen["critic"] += int(torch.cuda.current_stream() == trainer._critic_stream)
            return orig(*a, **k)
        trainer._optimizer_pass = spy
        bench.run_steps(trainer, task, batch, 3, 1)
        trainer._join_critic_stream()
        torch.cuda.synchronize()
        assert K.WGRAD_STREAM is None                        # (the side stream is routing state of a step only)
        w = torch.cat([p.detach().flatten() for p in task.gen_params + task.disc_params]).cpu()
        st = [dict(g.stats) for g in trainer.grad_sync if g is not None]
        order = [list(g.order) for g in trainer.grad_sync if g is not None]
    np.save(os.path.join(out, f"g{rank}.npy"), {"w": w.numpy(), "stats": st, "seen": seen, "order": order}, allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_same_step_as_one_rank(tmp_path, gpu_only):
    """The N > 1 step is the N = 1 step: weight gradients on the side stream and the critic pass on its own stream stay on with a
    gradient exchange (the bucket collectives are issued from a launch stream that has caught up with both producers).  Two
    ranks on ONE MI355X over gloo (RCCL refuses two ranks per device; the 8-GPU run is the driver's): three phase-2 steps,
    identical replicas afterwards, buckets exchanged from inside backward in descending order on both ranks."""
    probe = torch.zeros(4, device="cuda")
    port = _free_port()
    mp.spawn(_gpu_pair_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "g0.npy", allow_pickle=True).item()
    r1 = np.load(tmp_path / "g1.npy", allow_pickle=True).item()
    assert np.isfinite(r0["w"]).all() and np.array_equal(r0["w"], r1["w"])
    for r in (r0, r1):
        assert r["seen"]["side"] >= 6 and r["seen"]["critic"] >= 3, r["seen"]        # gen + critic passes of 3 steps
        assert sum(s["launched_in_backward"] for s in r["stats"]) > 0, r["stats"]
    assert r0["order"] == r1["order"]
    del probe


def _rccl_single_rank_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import argparse
    import sys
    import tempfile
    sys.path.insert(0, ROOT)
    import bench
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1)          # "nccl" IS RCCL on ROCm
    args = argparse.Namespace(batch=2, seconds=0.71, sample_rate=24000, bf16=False, precision="bf16x3", graph=False)
    small = (",hidden_size=32,fvae_enc_dec_hidden=32,latent_size=16,fvae_enc_n_layers=2,fvae_dec_n_layers=2,"
             "mel_disc_hidden_size=16,warmup_updates=4,ddp_bucket_mb=0.02")
    res = {}
    for name, extra in (("plain", ",defer_wgrad_reduce=False"), ("rccl", ",ddp_single_rank=True")):
        with tempfile.TemporaryDirectory() as tmp:
            task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp, extra_hparams=small + extra)
            assert trainer.use_ddp == (name == "rccl") and trainer.world_size == 1 and trainer.on_gpu
            syncs = [g for g in trainer.grad_sync if g is not None]
            assert all(g.exchange == (name == "rccl") and g.overlap == (name == "rccl") for g in syncs)
            bench.run_steps(trainer, task, batch, 3, 1)
            trainer._join_critic_stream()
            torch.cuda.synchronize()
            res[name] = {"w": torch.cat([p.detach().flatten() for p in task.gen_params + task.disc_params]).cpu().numpy(),
                         "stats": [dict(g.stats) for g in syncs], "order": [list(g.order) for g in syncs],
                         "backend": dist.get_backend(), "comm_streams": sum(g._comm_stream is not None for g in syncs)}
            del task, trainer, batch
    np.save(os.path.join(out, "rccl1.npy"), res, allow_pickle=True)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_single_rank_exchange_path_equals_the_plain_step(tmp_path, gpu_only):
    """What a one-GPU box can exercise of RCCL (reference: torch DDP over NCCL, utils/trainer.py:441-466): `ddp_single_rank: true`
    sends a ONE-rank run down the whole data-parallel path -- `nccl` process group, gradients as slices of the flat buffer,
    immediate weight-gradient reduces, buckets announced from inside backward, `all_reduce(async_op=True)` issued by RCCL from the
    launch stream that has joined the compute and weight-gradient streams, `work.wait()`, averaging (by 1).  A sum over one rank is
    the identity, so three phase-2 steps must leave the weights BIT-identical to the same steps without the exchange; buckets must
    have gone out from inside backward, in descending order."""
    probe = torch.zeros(4, device="cuda")
    mp.spawn(_rccl_single_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl1.npy", allow_pickle=True).item()
    assert r["rccl"]["backend"] == "nccl"
    assert np.isfinite(r["plain"]["w"]).all() and np.array_equal(r["plain"]["w"], r["rccl"]["w"])
    st = r["rccl"]["stats"]
    assert sum(s["launched_in_backward"] for s in st) > 0 and sum(s["passes"] for s in st) >= 6, st
    assert r["rccl"]["comm_streams"] >= 2                                  # generator and critic exchanged on their launch streams
    for order in r["rccl"]["order"]:
        if not order:                     # (the latent-map optimizer has no pass in phase 2)
            continue
        top = max(order)                  # every pass exchanges its buckets top, top - 1, ..., 0
        assert all(order[i + 1] == order[i] - 1 or (order[i] == 0 and order[i + 1] == top) for i in range(len(order) - 1)), order
    assert all(s["launched_in_backward"] == 0 and s["passes"] == 0 for s in r["plain"]["stats"])
    del probe

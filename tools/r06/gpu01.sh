#!/bin/bash
# r06 call 1: baseline back-to-back per-shape conv table at HEAD + RCCL world_size=1 probe
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python tools/shape_bench.py --top 80 > gpurun_out/r06_shape_bench_base.log 2>&1
timeout 120 python - > gpurun_out/r06_rccl_probe.log 2>&1 <<'PY'
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.arange(8, device="cuda", dtype=torch.float32)
s = torch.cuda.Stream()
w = dist.all_reduce(x, async_op=True)
w.wait()
torch.cuda.synchronize()
print("nccl ws=1 all_reduce ok", x.tolist(), dist.get_backend())
dist.destroy_process_group()
PY
tail -5 gpurun_out/r06_rccl_probe.log
tail -3 gpurun_out/r06_shape_bench_base.log

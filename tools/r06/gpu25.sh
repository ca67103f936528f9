#!/bin/bash
# r06 call 25: the GPU box's own host: issue time of a step with no-op kernels, with and without the critic tower executor
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tests/emu/build_emu.sh > /dev/null 2>&1
{ for i in 1 2; do
  echo "tower executor off:"; CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" timeout 600 python tools/host_overhead.py --steps 20 --no-tower-executor 2>&1 | grep -E "host time|Error" | tail -2
  echo "tower executor on:";  CUDA_VISIBLE_DEVICES="" HIP_VISIBLE_DEVICES="" timeout 600 python tools/host_overhead.py --steps 20 2>&1 | grep -E "host time|Error" | tail -2
done; } > gpurun_out/r06_host_overhead_tower.log 2>&1
cat gpurun_out/r06_host_overhead_tower.log

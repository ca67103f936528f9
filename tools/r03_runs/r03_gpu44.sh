#!/bin/bash
# runtime knobs: hardware queues (5 streams are in use: compute, weight gradients, critic, PPG, [RCCL]) and kernarg placement
O=gpurun_out/r03aj
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for envs in "X=1" "GPU_MAX_HW_QUEUES=8" "HIP_FORCE_DEV_KERNARG=1" "GPU_MAX_HW_QUEUES=8 HIP_FORCE_DEV_KERNARG=1" "GPU_MAX_HW_QUEUES=2"; do
  echo "== bench [$envs]: $(env $envs timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads 2>&1 >/dev/null | grep -h 'ms/step' | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';')"
done
done > $O/ab.log 2>&1
cat $O/ab.log

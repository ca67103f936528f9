#!/bin/bash
# round 4, GPU call 13: same-box A/B of the step-end side-stream join; device-time split of both
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for v in off on off on; do
  x=""; [ $v = on ] && x="--extra-hparams step_end_side_join=True"
  timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline $x > gpurun_out/r04_g13_bench_$v.log 2>&1
  echo "join=$v: $(grep 'ms/step\|issuing' gpurun_out/r04_g13_bench_$v.log | cut -c18-90 | tr '\n' '|')"
done
timeout 600 python tools/gpu_split.py > gpurun_out/r04_g13_gpu_split_off.log 2>&1; grep "host\|stream\|->" gpurun_out/r04_g13_gpu_split_off.log | cut -c1-130

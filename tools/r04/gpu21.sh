#!/bin/bash
# round 4, GPU call 21: split-K targets of the weight gradients (how much of the chip the side stream takes from the critical chain)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for v in "512 256" "256 256" "384 256" "256 128" "512 256" "256 256" "128 128" "384 192"; do
  set -- $v
  SVB_WG_BLOCKS=$1 SVB_WG_SMALL_BLOCKS=$2 timeout 600 python tools/bench_instr.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline > gpurun_out/r04_g21_bench_$1_$2.log 2>&1
  echo "blocks=$1 small=$2: $(grep 'ms/step\|issuing' gpurun_out/r04_g21_bench_$1_$2.log | cut -c18-90 | head -2 | tr '\n' '|')"
done

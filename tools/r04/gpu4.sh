#!/bin/bash
# round 4, GPU call 4: compile-time ablations + stage stamps of the producer/consumer tile-walking kernel
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
: > gpurun_out/r04_g4_tw_ablations.log
for a in "" 1 2 17 19 8; do
  echo "######## SVB_TW_ABL=$a" >> gpurun_out/r04_g4_tw_ablations.log
  SVB_TW_ABL=$a timeout 300 python tools/tw_stage_timing.py --cfgs 13 >> gpurun_out/r04_g4_tw_ablations.log 2>&1
done
grep -v "amdgpu.ids" gpurun_out/r04_g4_tw_ablations.log | cut -c1-150

#!/bin/bash
# round 4, GPU call 2: stage stamps + ablations of the tile-walking kernel (instrumentation build)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python tools/tw_stage_timing.py --cfgs 13,16 > gpurun_out/r04_g2_tw_stages.log 2>&1
timeout 600 python tools/tw_stage_timing.py --cfgs 13 --shape 32,256,256,562,1 >> gpurun_out/r04_g2_tw_stages.log 2>&1
cat gpurun_out/r04_g2_tw_stages.log

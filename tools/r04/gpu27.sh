#!/bin/bash
# round 4, GPU call 27: the whole GPU suite + smoke after the PPG encoder changes (fused projection, in-kernel position scores)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 200 python tools/attnbench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_g27_attnbench.log
bash tools/final_tests.sh r04

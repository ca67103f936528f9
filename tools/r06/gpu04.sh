#!/bin/bash
# r06 call 4: ablations of the pointwise kernel <2,1,4,2>: 28 no stores, 29 no x loads, 30 no MFMAs, 31 no W DMA, 32 no loads+stores, 33 nothing
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python tools/pwbench.py --cfgs 2,18,20,21,22,24,25,26,27,28,29,30,31,32,33 > gpurun_out/r06_pwbench_7.log 2>&1
cat gpurun_out/r06_pwbench_7.log

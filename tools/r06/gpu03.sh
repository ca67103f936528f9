#!/bin/bash
# r06 call 3: pointwise GEMM kernel, prefetch depth / phase size variants
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python tools/pwbench.py --cfgs 2,3,18,19,20,21,22,23,24,25,26,27 > gpurun_out/r06_pwbench_3.log 2>&1
cat gpurun_out/r06_pwbench_3.log

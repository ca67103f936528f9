#!/bin/bash
# r06 call 2: pointwise GEMM kernel first light -- parity tests on the MI355X + per-shape table against the tap-table tiles
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "pointwise" > gpurun_out/r06_pw_tests.log 2>&1
tail -3 gpurun_out/r06_pw_tests.log
timeout 600 python tools/pwbench.py > gpurun_out/r06_pwbench_1.log 2>&1
timeout 600 python tools/pwbench.py --epilogue > gpurun_out/r06_pwbench_1_epi.log 2>&1
cat gpurun_out/r06_pwbench_1.log

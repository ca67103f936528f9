#!/bin/bash
# r06 call 26: three-way split attention on the GPU: kernel parity, the fp32 golden steps against the fp64 arbiter (numbers printed)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "relpos_attention" -s 2>&1 | grep -E "passed|failed|torch fp32"
timeout 1500 python -m pytest tests/test_step_golden.py tests/test_modules_vae.py -x -q -m gpu -s > gpurun_out/r06_step_golden_split3.log 2>&1
tail -3 gpurun_out/r06_step_golden_split3.log
grep -E "vs fp64 arbiter \[fp32" gpurun_out/r06_step_golden_split3.log | cut -c1-220

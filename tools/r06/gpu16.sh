#!/bin/bash
# r06 call 16: full GPU suite (after the bench exchange-off fix)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu_mid3.log 2>&1
tail -6 gpurun_out/r06_pytest_gpu_mid3.log

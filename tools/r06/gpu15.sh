#!/bin/bash
# r06 call 15: full GPU suite after the period conv node / feature loss / tile lookup changes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu_mid2.log 2>&1
tail -6 gpurun_out/r06_pytest_gpu_mid2.log

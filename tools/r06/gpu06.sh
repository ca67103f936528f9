#!/bin/bash
# r06 call 6: full GPU suite at the pointwise-kernel commit + a second default bench line (the first one's timed region was host-bound)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r06_bench_pw2.json 2> gpurun_out/r06_bench_pw2.log
grep -E "ms/step|host finished|settled" gpurun_out/r06_bench_pw2.log | head
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu_mid.log 2>&1
tail -5 gpurun_out/r06_pytest_gpu_mid.log

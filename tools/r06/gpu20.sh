#!/bin/bash
# r06 call 20: bench line with the 16-tap table
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r06_bench_taps16.json 2> gpurun_out/r06_bench_taps16.log
grep -E "ms/step|host finished" gpurun_out/r06_bench_taps16.log | head -20

#!/bin/bash
# r06 call 12: weight gradients with paired 8-byte staging loads: GPU parity tests + the back-to-back per-shape table
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_functional.py -x -q -m gpu -k "wgrad or wn_stack or conv1d_autograd" > gpurun_out/r06_wgrad_tests.log 2>&1
tail -3 gpurun_out/r06_wgrad_tests.log
timeout 600 python tools/shape_bench.py --top 80 > gpurun_out/r06_shape_bench_pairs.log 2>&1
grep -E "wgrad|total conv" gpurun_out/r06_shape_bench_pairs.log | head -40

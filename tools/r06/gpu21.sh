#!/bin/bash
# r06 call 21: direct-operand 1-tap weight gradient: parity, per-shape table, bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "wgrad" 2>&1 | tail -4
SVB_BENCH_SHAPES=1 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra-workloads > /dev/null 2> gpurun_out/r06_conv_per_shape_wgpw.log
grep "wgrad" gpurun_out/r06_conv_per_shape_wgpw.log | head -24 | cut -c18-200
grep -E "ms/step|host finished" gpurun_out/r06_conv_per_shape_wgpw.log | head
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_wgpw.json 2> gpurun_out/r06_bench_wgpw.log
grep -E "ms/step|host finished" gpurun_out/r06_bench_wgpw.log | head -20

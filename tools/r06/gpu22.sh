#!/bin/bash
# r06 call 22: direct-operand weight gradient with 3 / 4 steps in flight
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "wgrad_pointwise" 2>&1 | tail -2
SVB_BENCH_SHAPES=1 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra-workloads > /dev/null 2> gpurun_out/r06_conv_per_shape_wgpw2.log
grep "wgrad" gpurun_out/r06_conv_per_shape_wgpw2.log | grep ", 1, 1, 1)" | head -24 | cut -c18-200
grep -E "ms/step|host finished" gpurun_out/r06_conv_per_shape_wgpw2.log | head -3

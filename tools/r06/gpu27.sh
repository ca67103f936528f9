#!/bin/bash
# r06 call 27: critic first-block streaming kernels on the GPU: parity, per-shape times, bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules_disc.py -x -q -m gpu -k "critic or tower or disc or taps" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_step_golden.py tests/test_task_step.py -x -q -m gpu 2>&1 | tail -3
SVB_BENCH_SHAPES=1 SVB_BENCH_SHAPES_TOP=400 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra-workloads > /dev/null 2> gpurun_out/r06_conv_per_shape_c4.log
grep -E "\('(taps|wgrad)', 1, (4, 128|128, 4)," gpurun_out/r06_conv_per_shape_c4.log | cut -c18-190 | sort | uniq
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_c4.json 2> gpurun_out/r06_bench_c4.log
grep -E "ms/step|host finished" gpurun_out/r06_bench_c4.log | head -12

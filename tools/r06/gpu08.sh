#!/bin/bash
# r06 call 8: RCCL single-rank test, variable-length CLI test, fused FlatAdamW test; default bench line with `variable_length`
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ddp_gloo.py tests/test_cli_gpu.py tests/test_task_step.py tests/test_kernels.py -x -q -m gpu -k "rccl or variable_length or flat_adamw or tile_" > gpurun_out/r06_new_tests.log 2>&1
tail -15 gpurun_out/r06_new_tests.log
timeout 900 python bench.py > gpurun_out/r06_bench_vl.json 2> gpurun_out/r06_bench_vl.log
grep -E "ms/step|host finished|settled|variable|failed" gpurun_out/r06_bench_vl.log | head -20

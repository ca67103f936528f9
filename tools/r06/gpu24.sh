#!/bin/bash
# r06 call 24: critic tower executor on the GPU: parity tests, bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_modules_disc.py tests/test_step_golden.py tests/test_task_step.py tests/test_ddp_gloo.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_tower.json 2> gpurun_out/r06_bench_tower.log
grep -E "ms/step|host finished" gpurun_out/r06_bench_tower.log | head -20
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads --no-roofline --extra-hparams tower_executor=False 2>&1 >/dev/null | grep -E "ms/step|host finished" | head -3

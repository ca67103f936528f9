#!/bin/bash
# round 4, GPU call 12: flat clip+AdamW (two launches), fused critic-tower node: parity (step goldens, task steps, ddp) + bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_step_golden.py tests/test_task_step.py tests/test_modules_disc.py tests/test_ddp_gloo.py tests/test_cli_gpu.py -q -m gpu -x > gpurun_out/r04_g12_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g12_pytest.log
tail -6 gpurun_out/r04_g12_pytest.log | cut -c1-250
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline > gpurun_out/r04_g12_bench_$i.log 2>&1
grep "ms/step\|issuing" gpurun_out/r04_g12_bench_$i.log | cut -c1-200
done
timeout 600 python tools/host_split.py > gpurun_out/r04_g12_host_split.log 2>&1; grep "opt\|host" gpurun_out/r04_g12_host_split.log

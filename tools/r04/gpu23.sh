#!/bin/bash
# round 4, GPU call 23: the critic's derived (space-to-depth) kernels refilled with the optimizer's multi-tensor pack: parity + A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_modules_disc.py tests/test_task_step.py tests/test_step_golden.py -q -m gpu -x > gpurun_out/r04_g23_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g23_pytest.log
grep -v "Warn\|warn\|sched\|Docs\|^$\|^tests" gpurun_out/r04_g23_pytest.log | tail -4 | cut -c1-200
bash tools/ab_bench.sh "" "critic_s2_registered=False" "" "critic_s2_registered=False"

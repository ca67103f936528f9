#!/bin/bash
# round 4, GPU call 26: position scores inside the attention kernel: parity, kernel timing, step A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules_vae.py tests/test_step_golden.py tests/test_task_step.py -q -m gpu -x -k "relpos or conformer or golden or reference_task or prefetch" > gpurun_out/r04_g26_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g26_pytest.log
grep -v "Warn\|warn\|sched\|Docs\|^$\|^tests" gpurun_out/r04_g26_pytest.log | tail -4 | cut -c1-200
timeout 200 python tools/attnbench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_g26_attnbench.log
bash tools/ab_bench.sh "" "ppg_pos_in_kernel=False" "" "ppg_pos_in_kernel=False"

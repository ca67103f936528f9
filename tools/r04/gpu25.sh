#!/bin/bash
# round 4, GPU call 25: PPG encoder with one D -> 4D projection per attention layer + cached position projection: parity + A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules_vae.py tests/test_step_golden.py tests/test_task_step.py -q -m gpu -x -k "relpos or conformer or golden or reference_task or oracle or prefetch" > gpurun_out/r04_g25_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g25_pytest.log
grep -v "Warn\|warn\|sched\|Docs\|^$\|^tests" gpurun_out/r04_g25_pytest.log | tail -4 | cut -c1-200
bash tools/ab_bench.sh "" "ppg_fuse_qkv=False" "" "ppg_fuse_qkv=False"

#!/bin/bash
# round 4, GPU call 20: the critic's window towers on streams of their own: parity + same-box A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_modules_disc.py tests/test_task_step.py tests/test_step_golden.py tests/test_modules_vae.py -q -m gpu -x > gpurun_out/r04_g20_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g20_pytest.log
grep -v "Warn\|warn\|sched\|Docs\|^$\|^tests" gpurun_out/r04_g20_pytest.log | tail -5 | cut -c1-200
for v in on off on off; do
  x=""; [ $v = off ] && x="--extra-hparams critic_tower_streams=False"
  timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline $x > gpurun_out/r04_g20_bench_$v.log 2>&1
  echo "towers=$v: $(grep 'ms/step\|issuing' gpurun_out/r04_g20_bench_$v.log | cut -c18-90 | tr '\n' '|')"
done

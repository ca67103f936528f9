#!/bin/bash
# round 4, GPU call 14: PPG prefetch: equality test + same-box A/B + device split
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_task_step.py tests/test_step_golden.py -q -m gpu -x > gpurun_out/r04_g14_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g14_pytest.log
grep -v "Warn\|warn\|sched\|Docs\|^$" gpurun_out/r04_g14_pytest.log | tail -6 | cut -c1-200
for v in on off on off; do
  x=""; [ $v = off ] && x="--extra-hparams prefetch_next_batch=False"
  timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline $x > gpurun_out/r04_g14_bench_$v.log 2>&1
  echo "prefetch=$v: $(grep 'ms/step\|issuing' gpurun_out/r04_g14_bench_$v.log | cut -c18-90 | tr '\n' '|')"
done
timeout 600 python tools/gpu_split.py > gpurun_out/r04_g14_gpu_split.log 2>&1; grep "host\|stream\|->" gpurun_out/r04_g14_gpu_split.log | cut -c1-130 | head -8

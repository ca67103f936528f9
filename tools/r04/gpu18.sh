#!/bin/bash
# round 4, GPU call 18: launch trimming (BatchNorm kernel, pooled Dropout2d draw, adopted gradients gathered by one launch):
# parity of the new kernels and of the step, same-box A/B of the step time
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels.py tests/test_functional.py -q -m gpu -x -k "batchnorm or gather_segments or dropout2d" > gpurun_out/r04_g18_pytest.log 2>&1
echo "pytest kernels rc=$?" >> gpurun_out/r04_g18_pytest.log
timeout 900 python -m pytest tests/test_task_step.py tests/test_step_golden.py tests/test_modules_vae.py -q -m gpu -x >> gpurun_out/r04_g18_pytest.log 2>&1
echo "pytest step rc=$?" >> gpurun_out/r04_g18_pytest.log
grep -v "Warn\|warn\|sched\|Docs\|^$" gpurun_out/r04_g18_pytest.log | tail -8 | cut -c1-200
for v in on off on off; do
  x=""; [ $v = off ] && x="--extra-hparams drop_autograd_grads=False"
  timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline $x > gpurun_out/r04_g18_bench_$v.log 2>&1
  echo "adopt=$v: $(grep 'ms/step\|issuing' gpurun_out/r04_g18_bench_$v.log | cut -c18-90 | tr '\n' '|')"
done
timeout 600 python tools/gpu_split.py > gpurun_out/r04_g18_gpu_split.log 2>&1; grep "host\|stream\|->" gpurun_out/r04_g18_gpu_split.log | cut -c1-130 | head -8

#!/bin/bash
# round 4, GPU call 24: spectral norm as 3 + 2 launches (csrc/spectral_norm.hip): parity + vocoder step A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_modules_hifigan.py tests/test_hifigan_task.py tests/test_vocoder_plugin.py -q -m gpu -x > gpurun_out/r04_g24_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g24_pytest.log
grep -v "Warn\|warn\|sched\|Docs\|^$\|^tests" gpurun_out/r04_g24_pytest.log | tail -4 | cut -c1-200
for v in on off on off; do
  x=""; [ $v = off ] && x="--extra-hparams fused_spectral_norm=False"
  timeout 600 python bench.py --workload vocoder --steps 8 --warmup 4 --no-cpu-baseline $x > gpurun_out/r04_g24_voc_$v.json 2> gpurun_out/r04_g24_voc_$v.log
  echo "fused_spectral_norm=$v: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r04_g24_voc_$v.json | head -1)"
done

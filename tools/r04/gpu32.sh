#!/bin/bash
# round 4, GPU call 32: division-free index walks (ssim.hip, conv2d.hip crop/drop/norm, period_ops.hip): whole GPU suite + smoke,
# vocoder step, train step
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash tools/final_tests.sh r04
timeout 600 python bench.py --workload vocoder --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r04_g32_voc.json 2> gpurun_out/r04_g32_voc.log
echo "vocoder: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r04_g32_voc.json | head -1)"
bash tools/ab_bench.sh "" ""

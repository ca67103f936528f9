#!/bin/bash
# round 4, GPU call 31: SSIM / mel-loss kernels without per-element integer divisions: parity, microbench, step A/B is the bench line
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels.py tests/test_functional.py tests/test_step_golden.py -q -m gpu -x -k "ssim or mel_loss or golden" > gpurun_out/r04_g31_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g31_pytest.log; tail -2 gpurun_out/r04_g31_pytest.log
timeout 120 python tools/ewbench.py 2>&1 | grep "mel_loss\|ssim" | tee gpurun_out/r04_g31_ewbench.log
bash tools/ab_bench.sh "" ""

#!/bin/bash
# round 4, GPU call 17: tile-walking kernel with quad x staging + rebalanced weight DMA: parity, per-shape timing, stage stamps
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "tile_walking" > gpurun_out/r04_g17_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g17_pytest.log
tail -3 gpurun_out/r04_g17_pytest.log
timeout 600 python tools/twbench.py --iters 10 > gpurun_out/r04_g17_twbench.log 2>&1
tail -25 gpurun_out/r04_g17_twbench.log
timeout 900 python tools/tw_stage_timing.py --cfgs 13,14 > gpurun_out/r04_g17_tw_stages.log 2>&1
for a in 1 2 17 19; do SVB_TW_ABL=$a timeout 300 python tools/tw_stage_timing.py --cfgs 13 2>&1 | grep -E "us|ablat" >> gpurun_out/r04_g17_tw_abl.log; done
tail -60 gpurun_out/r04_g17_tw_stages.log
cat gpurun_out/r04_g17_tw_abl.log

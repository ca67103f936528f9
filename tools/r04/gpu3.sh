#!/bin/bash
# round 4, GPU call 3: producer/consumer tile-walking kernel: hardware parity, per-shape timing, stage stamps
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "tile_walking" > gpurun_out/r04_g3_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g3_pytest.log
tail -3 gpurun_out/r04_g3_pytest.log
timeout 600 python tools/twbench.py --iters 10 > gpurun_out/r04_g3_twbench.log 2>&1
tail -25 gpurun_out/r04_g3_twbench.log
timeout 900 python tools/tw_stage_timing.py --cfgs 13,14 > gpurun_out/r04_g3_tw_stages.log 2>&1
timeout 600 python tools/tw_stage_timing.py --cfgs 13 --shape 32,256,256,562,1 >> gpurun_out/r04_g3_tw_stages.log 2>&1
cat gpurun_out/r04_g3_tw_stages.log

#!/bin/bash
# round 4, GPU call 1: hardware parity of the tile-walking conv kernel, per-shape timing of all 18 tile configurations,
# and the train step with / without the new configurations in the autotuner's candidate set.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "tile_walking or direct_tiles or clip_edges" > gpurun_out/r04_g1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g1_pytest.log
tail -3 gpurun_out/r04_g1_pytest.log
timeout 600 python tools/twbench.py --iters 10 > gpurun_out/r04_g1_twbench.log 2>&1
tail -25 gpurun_out/r04_g1_twbench.log
SVB_NCFG_Q=12 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-workloads --no-roofline > gpurun_out/r04_g1_bench_old.log 2>&1
tail -4 gpurun_out/r04_g1_bench_old.log | cut -c1-400
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-workloads --no-roofline > gpurun_out/r04_g1_bench_tw.log 2>&1
tail -4 gpurun_out/r04_g1_bench_tw.log | cut -c1-400

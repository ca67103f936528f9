#!/bin/bash
# r06 call 17: the GEMM kernel with taps / input gates: parity on the GPU, the tile table regenerated with it, the bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "gemm_kernel_with_taps or pointwise_gemm" 2>&1 | tail -5
timeout 1500 python tools/tune_tiles.py --out gpurun_out/tile_table.json > gpurun_out/r06_tile_tuner_taps.log 2>&1
tail -3 gpurun_out/r06_tile_tuner_taps.log
cp neuralsvb_amd/tile_table.json gpurun_out/tile_table_before.json
cp gpurun_out/tile_table.json neuralsvb_amd/tile_table.json
timeout 900 python bench.py > gpurun_out/r06_bench_taps.json 2> gpurun_out/r06_bench_taps.log
grep -E "ms/step|host finished" gpurun_out/r06_bench_taps.log | head -20

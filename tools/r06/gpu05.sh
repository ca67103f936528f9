#!/bin/bash
# r06 call 5: regenerate the tile table with the pointwise configurations (18..23), then the default bench line with it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python tools/tune_tiles.py --out gpurun_out/tile_table.json > gpurun_out/r06_tile_tuner.log 2>&1
tail -5 gpurun_out/r06_tile_tuner.log
cp gpurun_out/tile_table.json neuralsvb_amd/tile_table.json
timeout 900 python bench.py > gpurun_out/r06_bench_pw.json 2> gpurun_out/r06_bench_pw.log
tail -c 1500 gpurun_out/r06_bench_pw.json
grep -E "ms/step|host finished" gpurun_out/r06_bench_pw.log | head

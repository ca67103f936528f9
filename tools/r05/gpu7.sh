#!/bin/bash
# round 5, GPU call 7: the closing full GPU suite + smoke at HEAD, and one more fresh-lease default bench line.
mkdir -p gpurun_out/r05_g7
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
bash tools/final_tests.sh r05
timeout 900 python bench.py > gpurun_out/r05_g7/bench.json 2> gpurun_out/r05_g7/bench.log
echo "bench rc=$?"; grep -h "ms/step\|settled\|secondary" gpurun_out/r05_g7/bench.log | tail -9

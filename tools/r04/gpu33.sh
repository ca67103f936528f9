#!/bin/bash
# round 4, GPU call 33: pitch-embedding gather without per-element index divisions: parity (whole suite + smoke), step
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash tools/final_tests.sh r04
bash tools/ab_bench.sh "" ""

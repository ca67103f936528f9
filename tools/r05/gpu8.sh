#!/bin/bash
# round 5, GPU call 8: closing full GPU suite + smoke after the LOCAL_RANK fix, host time per autograd node class.
mkdir -p gpurun_out/r05_g8
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
bash tools/final_tests.sh r05
timeout 300 python tools/node_profile.py > gpurun_out/r05_g8/node_profile.log 2>&1
tail -45 gpurun_out/r05_g8/node_profile.log

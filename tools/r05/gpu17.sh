#!/bin/bash
# round 5, GPU call 17: the N>1 bench path after the replica-digest change (two ranks on one GPU), and a default bench line.
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r05_g17
timeout 900 python -m pytest tests/test_ddp_gloo.py -m gpu -q -p no:cacheprovider > gpurun_out/r05_g17/tests.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/r05_g17/tests.log
timeout 900 python bench.py > gpurun_out/r05_g17/bench.json 2> gpurun_out/r05_g17/bench.log
echo "bench rc=$?"; grep -h "ms/step\|settled" gpurun_out/r05_g17/bench.log | tail -9

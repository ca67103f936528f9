#!/bin/bash
# round 5, GPU call 18: which source lines issue the step's remaining ATen launches.
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r05_g18
timeout 600 python tools/op_attribution.py > gpurun_out/r05_g18/op_attribution.log 2>&1
echo "rc=$?"; tail -70 gpurun_out/r05_g18/op_attribution.log | cut -c1-200

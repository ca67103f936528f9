#!/bin/bash
O=gpurun_out/r03t
mkdir -p $O
export TMPDIR=/tmp
for envs in "X=1" "SVB_ABLATE=9" "SVB_LDS_PAD_KB=70 SVB_ABLATE=9"; do
  echo "== env [$envs]"
  env $envs timeout 60 python tools/stage_timing.py 32 256 256 1124 5 2 2>&1 | grep "compute\|wait weight\|store next\|stage total\|kernel [0-9]"
done > $O/ablate.log 2>&1
cat $O/ablate.log

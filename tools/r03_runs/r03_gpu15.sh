#!/bin/bash
O=gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
for envs in "X=1" "SVB_CONV_DENSE=1" "SVB_LDS_PAD_KB=70" "SVB_CONV_DENSE=1 SVB_LDS_PAD_KB=70"; do
  for shape in "32 192 384 1124 5 2" "32 256 256 1124 5 2" "32 192 384 1124 1 2" "32 192 384 281 5 2"; do
    echo "== env [$envs] shape [$shape]"
    env $envs timeout 60 python tools/stage_timing.py $shape 2>&1 | grep -v amdgpu.ids | grep "compute\|wait weight\|store next\|stage total\|kernel [0-9]"
  done
done > $O/stage_dense.log 2>&1
cat $O/stage_dense.log

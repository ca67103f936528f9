#!/bin/bash
O=gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp
for envs in "X=1" "SVB_LDS_PAD_KB=70"; do
  for shape in "32 192 384 1124 5 2" "32 256 256 1124 5 2" "32 192 384 1124 1 2" "32 192 384 281 5 2" "32 192 384 1124 5 3"; do
    echo "== env [$envs] shape [$shape]"
    env $envs timeout 60 python tools/stage_timing.py $shape 2>&1 | grep -v amdgpu.ids | grep "compute\|wait weight\|store next\|stage total\|kernel [0-9]\|prologue\|epilogue\|workgroup total"
  done
done > $O/stage_dense2.log 2>&1
(timeout 300 python -m pytest tests/test_kernels.py tests/test_functional.py -m gpu -x -q 2>&1 | tail -3) > $O/pytest_kernels.log
bash tools/ab_bench.sh "" "" > $O/ab.log 2>&1
cat $O/stage_dense2.log; cat $O/pytest_kernels.log; cat $O/ab.log

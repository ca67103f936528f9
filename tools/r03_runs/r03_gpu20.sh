#!/bin/bash
# quad (16-byte along time) x-tile staging: parity on the MI355X, per-shape and whole-step A/B against the dword staging
O=gpurun_out/r03u
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_kernels.py tests/test_functional.py -m gpu -x -q 2>&1 | tail -4) > $O/pytest_kernels.log
tail -3 $O/pytest_kernels.log
for envs in "X=1" "SVB_NO_X4=1"; do
  echo "== shape_bench [$envs]"
  env $envs timeout 200 python tools/shape_bench.py --top 24 2>&1 | tail -32
done > $O/shape.log 2>&1
cat $O/shape.log
for envs in "X=1" "SVB_NO_X4=1" "X=1" "SVB_NO_X4=1"; do
  echo "== bench [$envs]: $(env $envs timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads 2>&1 >/dev/null | grep -h 'ms/step' | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';')"
done > $O/ab.log 2>&1
cat $O/ab.log

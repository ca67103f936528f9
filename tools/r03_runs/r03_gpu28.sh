#!/bin/bash
# conv epilogue through a per-wave LDS transpose (16-byte stores): parity + per-shape + step A/B against SVB_NO_VEC_EPILOGUE=1
O=gpurun_out/r03aa
mkdir -p $O
export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_kernels.py tests/test_functional.py tests/test_q_layout.py -m gpu -x -q 2>&1 | tail -3) > $O/pytest_kernels.log
tail -2 $O/pytest_kernels.log
for envs in "X=1" "SVB_NO_VEC_EPILOGUE=1"; do
  echo "== shape_bench [$envs]"
  env $envs timeout 200 python tools/shape_bench.py --top 30 2>&1 | grep "total conv\| x " 
done > $O/shape.log 2>&1
cat $O/shape.log
for envs in "X=1" "SVB_NO_VEC_EPILOGUE=1" "X=1" "SVB_NO_VEC_EPILOGUE=1"; do
  echo "== bench [$envs]: $(env $envs timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads 2>&1 >/dev/null | grep -h 'ms/step' | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';')"
done > $O/ab.log 2>&1
cat $O/ab.log
for envs in "X=1" "SVB_NO_VEC_EPILOGUE=1"; do
  echo "== stamps [$envs]"
  for shp in "32 192 384 1124 5 2" "32 192 384 1124 1 2" "32 192 384 281 5 2"; do
    env $envs timeout 60 python tools/stage_timing.py $shp 2>&1 | grep "shape\|stage total\|prologue\|epilogue\|workgroup total\|kernel [0-9]"
  done
done > $O/stamps.log 2>&1
cat $O/stamps.log

#!/bin/bash
O=gpurun_out/r03ac
mkdir -p $O
export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_functional.py tests/test_modules_vae.py tests/test_step_golden.py -m gpu -x -q 2>&1 | tail -3) > $O/pytest.log
tail -2 $O/pytest.log
for envs in "X=1" "SVB_GN_NO_ROWS=1"; do echo "== ewbench [$envs]"; env $envs timeout 100 python tools/ewbench.py 2>&1 | grep gn_relu; done > $O/ew.log 2>&1
cat $O/ew.log
for envs in "X=1" "SVB_GN_NO_ROWS=1" "X=1" "SVB_GN_NO_ROWS=1"; do
  echo "== bench [$envs]: $(env $envs timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads 2>&1 >/dev/null | grep -h 'ms/step' | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';')"
done > $O/ab.log 2>&1
cat $O/ab.log

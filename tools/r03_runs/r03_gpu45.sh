#!/bin/bash
O=gpurun_out/r03ak
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for envs in "X=1" "SVB_WG_SMALL_BLOCKS=384" "SVB_WG_SMALL_BLOCKS=256"; do
  echo "== train [$envs]: $(env $envs timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads 2>&1 >/dev/null | grep -h 'ms/step' | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';' | cut -c1-90)"
done
done > $O/ab.log 2>&1
for envs in "X=1" "SVB_WG_SMALL_BLOCKS=384" "SVB_WG_SMALL_BLOCKS=256"; do
  echo "== vocoder [$envs]: $(env $envs timeout 300 python bench.py --workload vocoder --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2))')"
done >> $O/ab.log 2>&1
cat $O/ab.log

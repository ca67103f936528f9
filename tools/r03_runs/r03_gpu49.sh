#!/bin/bash
O=gpurun_out/r03an
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for envs in "SVB_WG_SMALL_BLOCKS=0" "SVB_WG_SMALL_BLOCKS=256"; do
  echo "== train [$envs]: $(env $envs timeout 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads 2>&1 >/dev/null | grep -h 'ms/step' | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';' | cut -c1-90)"
done
done > $O/ab.log 2>&1
cat $O/ab.log

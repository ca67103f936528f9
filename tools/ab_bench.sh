#!/bin/bash
# A/B of train-step variants on ONE MI355X box, back to back (a gpurun call of this script costs ~10 s per variant):
#   gpurun --timeout 200 -- 'bash tools/ab_bench.sh "" "defer_wgrad_reduce=True" "--legacy-paths"'
# Each argument is either an hparams override string (k=v,k=v) or a bench.py flag (starts with --); "" = defaults.
# Prints ms/step (+ host issue time) per variant; the JSON lines land in gpurun_out/ab/.
mkdir -p gpurun_out/ab
i=0
for v in "$@"; do
  i=$((i+1))
  if [[ "$v" == --* ]]; then extra="$v"; elif [[ -n "$v" ]]; then extra="--extra-hparams $v"; else extra=""; fi
  timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads $extra > gpurun_out/ab/v$i.json 2> gpurun_out/ab/v$i.err
  echo "variant $i [${v:-defaults}]: $(grep -h 'ms/step' gpurun_out/ab/v$i.err | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';')"
done

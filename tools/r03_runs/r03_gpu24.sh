#!/bin/bash
O=gpurun_out/r03y
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for envs in "X=1" "SVB_WG_DIL_TGW=3" "SVB_WG_DIL_TGW=2"; do
  echo "== vocoder bench [$envs]: $(env $envs timeout 300 python bench.py --workload vocoder --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), "ms/step; conv", d["roofline"]["all_conv_kernels"])')"
done > $O/ab.log 2>&1
cat $O/ab.log
SVB_BENCH_SHAPES=1 SVB_BENCH_SHAPES_TOP=400 timeout 400 python bench.py --workload vocoder --steps 3 --warmup 3 --no-cpu-baseline --no-side-stream > $O/voc.json 2> $O/voc_shapes.log
cd /tmp
rm -rf /tmp/prof_v
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o r03 --output-format csv -- \
   python $R/bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > $R/$O/vocoder.json 2> $R/$O/vocoder.err
python $R/tools/trace_summary.py /tmp/prof_v/r03_kernel_trace.csv 4 60 > $R/$O/kernel_summary_vocoder.txt
cd $R
head -50 $O/kernel_summary_vocoder.txt | cut -c1-150

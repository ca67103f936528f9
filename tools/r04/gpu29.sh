#!/bin/bash
# round 4, GPU call 29: where the compute stream waits (rocprofv3 kernel trace of the default bench, gaps by neighbouring kernels)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp && rm -rf /tmp/prof_g
SVB_BENCH_MARKERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o r04 --output-format csv -- \
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads > $R/gpurun_out/r04_g29_bench.json 2> $R/gpurun_out/r04_g29_bench.err
python $R/tools/trace_summary.py /tmp/prof_g/r04_kernel_trace.csv 20 12 > $R/gpurun_out/r04_g29_gaps.txt
grep -A16 "gaps > 20 us by" $R/gpurun_out/r04_g29_gaps.txt

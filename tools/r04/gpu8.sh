#!/bin/bash
# round 4, GPU call 8: rocprofv3 kernel trace of the eager bench at HEAD (default streams), per-queue view
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -rf /tmp/prof_d
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r04 --output-format csv -- \
   python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads \
   > $R/gpurun_out/r04_g8_bench_under_rocprof.json 2> $R/gpurun_out/r04_g8_bench_under_rocprof.err
python $R/tools/trace_summary.py /tmp/prof_d/r04_kernel_trace.csv 20 70 > $R/gpurun_out/r04_g8_kernel_summary.txt
head -1 /tmp/prof_d/r04_kernel_trace.csv
cd $R
grep "ms/step" gpurun_out/r04_g8_bench_under_rocprof.err | cut -c1-150
head -12 gpurun_out/r04_g8_kernel_summary.txt; tail -12 gpurun_out/r04_g8_kernel_summary.txt

#!/bin/bash
# r06 call 11: kernel trace of the vocoder step
cd "$GRAFT_REPO_ROOT"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_voc
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_voc -o r06 --output-format csv -- \
   python $R/bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r06_voc_rocprof.json 2> $R/gpurun_out/r06_voc_rocprof.err
python $R/tools/trace_summary.py /tmp/prof_voc/r06_kernel_trace.csv 4 70 > $R/gpurun_out/r06_vocoder_kernel_summary.txt
head -8 $R/gpurun_out/r06_vocoder_kernel_summary.txt
grep -A4 "per Queue_Id" $R/gpurun_out/r06_vocoder_kernel_summary.txt

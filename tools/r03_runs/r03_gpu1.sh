#!/bin/bash
# Round 3, GPU call 1: (1) full gpu suite at HEAD, (2) the epilogue variants (MODE 3/4/5) on the MI355X, (3) A/B of their switches,
# (4) kernel trace of the default (benchmarked) configuration + one without the side stream.
#   gpurun --timeout 900 -- 'bash tools/r03_runs/r03_gpu1.sh'
set -x
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
(timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest_gpu.log
(SVB_TEST_UNTIMED=1 timeout 200 python -m pytest tests/test_kernels.py tests/test_functional.py -m gpu -q \
   -k "res_skip_epilogue or gate_epilogue or gate_backward or fused_res_skip" 2>&1 | tail -25) > $O/pytest_epilogues.log
bash tools/ab_bench.sh "" "wn_fuse_res_skip=True" "wn_fuse_gate=True" "wn_fuse_res_skip=True,wn_fuse_gate=True" "" \
   > $O/ab.log 2>&1
cp gpurun_out/ab/*.json gpurun_out/ab/*.err $O/ 2>/dev/null
R=$PWD
cd /tmp
SVB_BENCH_MARKERS=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r03 --output-format csv -- \
   python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline \
   > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err
python $R/tools/trace_summary.py /tmp/prof/r03_kernel_trace.csv 20 90 > $R/$O/kernel_summary_default.txt
SVB_BENCH_MARKERS=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o r03 --output-format csv -- \
   python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-side-stream \
   > $R/$O/bench_under_rocprof_noside.json 2> $R/$O/bench_under_rocprof_noside.err
python $R/tools/trace_summary.py /tmp/prof2/r03_kernel_trace.csv 20 90 > $R/$O/kernel_summary_noside.txt
cd $R
cat $O/pytest_gpu.log | tail -8; cat $O/pytest_epilogues.log | tail -8; cat $O/ab.log; head -12 $O/kernel_summary_default.txt

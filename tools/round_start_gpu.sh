#!/bin/bash
# First GPU call of the next round (~2.5 min of box time): (1) the MI355X status of the epilogue variants that landed untimed
# (svb_conv1d_bf16x3_kernel MODE 3/4/5), (2) A/B of their switches on the train step, back to back on one box, (3) a kernel
# trace of the default configuration.        gpurun --timeout 400 -- 'bash tools/round_start_gpu.sh'
set -x
mkdir -p gpurun_out/r03_start
cd /root/repo
export TMPDIR=/tmp
(SVB_TEST_UNTIMED=1 timeout 120 python -m pytest tests/test_kernels.py tests/test_functional.py -m gpu -q \
   -k "res_skip_epilogue or gate_epilogue or gate_backward or fused_res_skip" 2>&1 | tail -8) > gpurun_out/r03_start/pytest_epilogues.log
bash tools/ab_bench.sh "" "wn_fuse_res_skip=True" "wn_fuse_gate=True" "wn_fuse_res_skip=True,wn_fuse_gate=True" "" \
   > gpurun_out/r03_start/ab.log 2>&1
cp gpurun_out/ab/*.json gpurun_out/r03_start/ 2>/dev/null
cd /tmp
SVB_BENCH_MARKERS=1 timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r03 --output-format csv -- \
   python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-side-stream \
   > /root/repo/gpurun_out/r03_start/bench_under_rocprof.json 2> /root/repo/gpurun_out/r03_start/bench_under_rocprof.err
python /root/repo/tools/trace_summary.py /tmp/prof/r03_kernel_trace.csv 20 70 > /root/repo/gpurun_out/r03_start/kernel_summary.txt
cd /root/repo
cat gpurun_out/r03_start/pytest_epilogues.log gpurun_out/r03_start/ab.log; head -8 gpurun_out/r03_start/kernel_summary.txt

#!/bin/bash
# full GPU suite (device_collate default on, MPD layout, grouped wgrad packing) + default bench line + idle analysis of the step
O=gpurun_out/r03z
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_train_bf16x3.json 2> $O/bench_train_bf16x3.log
grep "ms/step\|roofline\|extra\|vocoder\|infer" $O/bench_train_bf16x3.log | cut -c1-200
cd /tmp
rm -rf /tmp/prof_t
SVB_BENCH_MARKERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o r03 --output-format csv -- \
   python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads > $R/$O/train.json 2> $R/$O/train.err
python $R/tools/trace_summary.py /tmp/prof_t/r03_kernel_trace.csv 20 12 > $R/$O/kernel_summary_train.txt
cd $R
head -8 $O/kernel_summary_train.txt | cut -c1-200

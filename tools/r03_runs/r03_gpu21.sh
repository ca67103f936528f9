#!/bin/bash
# (1) weight-gradient reduce with 8 partials in flight: step A/B is against the numbers of r03u (same code otherwise) + kernel table
# (2) kernel table of the vocoder workload (configs[2])
O=gpurun_out/r03v
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "wgrad or weight_norm or deferred" 2>&1 | tail -2) > $O/pytest_wgrad.log
cat $O/pytest_wgrad.log
for i in 1 2; do
  echo "== bench: $(timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads 2>&1 >/dev/null | grep -h 'ms/step' | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';')"
done > $O/ab.log 2>&1
cat $O/ab.log
cd /tmp
rm -rf /tmp/prof_t
SVB_BENCH_MARKERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o r03 --output-format csv -- \
   python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads > $R/$O/train.json 2> $R/$O/train.err
python $R/tools/trace_summary.py /tmp/prof_t/r03_kernel_trace.csv 20 40 > $R/$O/kernel_summary_train.txt
rm -rf /tmp/prof_v
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o r03 --output-format csv -- \
   python $R/bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > $R/$O/vocoder.json 2> $R/$O/vocoder.err
python $R/tools/trace_summary.py /tmp/prof_v/r03_kernel_trace.csv 4 70 > $R/$O/kernel_summary_vocoder.txt
cd $R
grep -h "reduce" $O/kernel_summary_train.txt
head -75 $O/kernel_summary_vocoder.txt

#!/bin/bash
# refresh the rocprofv3 kernel tables (benchmarked / serial / vocoder configurations) at the final kernels
O=gpurun_out/r03_final
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for v in default noside; do
  fl=""; [ $v = noside ] && fl="--no-side-stream --extra-hparams overlap_critic_pass=False,overlap_ppg_encoder=False"
  rm -rf /tmp/prof_$v
  SVB_BENCH_MARKERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o r03 --output-format csv -- \
     python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads $fl \
     > $R/$O/bench_under_rocprof_$v.json 2> $R/$O/bench_under_rocprof_$v.err
  python $R/tools/trace_summary.py /tmp/prof_$v/r03_kernel_trace.csv 20 80 > $R/$O/kernel_summary_$v.txt
  cp /tmp/prof_$v/r03_kernel_stats.csv $R/$O/kernel_stats_$v.csv 2>/dev/null
done
rm -rf /tmp/prof_voc
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_voc -o r03 --output-format csv -- \
   python $R/bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > $R/$O/bench_under_rocprof_vocoder.json 2> $R/$O/bench_under_rocprof_vocoder.err
python $R/tools/trace_summary.py /tmp/prof_voc/r03_kernel_trace.csv 4 60 > $R/$O/kernel_summary_vocoder.txt
cd $R
head -3 $O/kernel_summary_default.txt $O/kernel_summary_noside.txt $O/kernel_summary_vocoder.txt | cut -c1-220

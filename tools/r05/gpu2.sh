#!/bin/bash
# round 5, GPU call 2: full GPU suite + smoke at the tile-table HEAD, default bench line (fresh-lease sample), host cProfile,
# rocprofv3 kernel trace of the benchmarked configuration.
O=gpurun_out/r05_g2
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 900 python bench.py > $O/bench.json 2> $O/bench.log
echo "bench rc=$?"; grep -h "ms/step\|settled" $O/bench.log | tail -8
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -2 $O/smoke.log
timeout 200 python tools/host_profile.py > $O/host_profile.log 2>&1
cd /tmp
rm -rf /tmp/prof_d
SVB_BENCH_MARKERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r05 --output-format csv -- \
   python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads \
   > $R/$O/bench_under_rocprof_default.json 2> $R/$O/bench_under_rocprof_default.err
python $R/tools/trace_summary.py /tmp/prof_d/r05_kernel_trace.csv 20 90 > $R/$O/kernel_summary_default.txt
cp /tmp/prof_d/r05_kernel_stats.csv $R/$O/kernel_stats_default.csv 2>/dev/null
cd $R
head -12 $O/kernel_summary_default.txt

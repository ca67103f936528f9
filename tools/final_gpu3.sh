set -x
mkdir -p /root/repo/gpurun_out/r02d
export TMPDIR=/tmp
cd /tmp
SVB_BENCH_MARKERS=1 timeout 80 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r02 --output-format csv -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-side-stream > /root/repo/gpurun_out/r02d/bench_under_rocprof.json 2> /root/repo/gpurun_out/r02d/bench_under_rocprof.err
python /root/repo/tools/trace_summary.py /tmp/prof/r02_kernel_trace.csv 20 70 > /root/repo/gpurun_out/r02d/kernel_summary.txt
cp /tmp/prof/r02_kernel_stats.csv /root/repo/gpurun_out/r02d/kernel_stats.csv
cd /root/repo
(timeout 110 python -m pytest tests -m gpu -q -x 2>&1 | tail -6) > gpurun_out/r02d/pytest_gpu.log
head -8 gpurun_out/r02d/kernel_summary.txt; tail -3 gpurun_out/r02d/pytest_gpu.log

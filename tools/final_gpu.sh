set -x
mkdir -p gpurun_out/r02_final
cd /root/repo
export TMPDIR=/tmp
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r02_final/pytest_gpu.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_final/smoke.log 2>&1
timeout 400 python tools/pmc_traffic.py > gpurun_out/r02_final/pmc_traffic.log 2>&1
cp profiles/r02_pmc_traffic.json gpurun_out/r02_final/ 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_final/bench_train_bf16x3.json 2> gpurun_out/r02_final/bench_train_bf16x3.err
timeout 300 python bench.py --steps 20 --warmup 5 --precision fp32 --no-cpu-baseline > gpurun_out/r02_final/bench_train_fp32.json 2> gpurun_out/r02_final/bench_train_fp32.err
timeout 300 python bench.py --workload vocoder --steps 8 --warmup 3 > gpurun_out/r02_final/bench_vocoder.json 2> gpurun_out/r02_final/bench_vocoder.err
timeout 300 python bench.py --workload infer --steps 12 --warmup 2 > gpurun_out/r02_final/bench_infer.json 2> gpurun_out/r02_final/bench_infer.err
timeout 200 python bench.py --steps 10 --warmup 4 --graph --no-cpu-baseline --no-roofline > gpurun_out/r02_final/bench_train_graph.json 2> gpurun_out/r02_final/bench_train_graph.err
(timeout 100 python tools/attnbench.py; timeout 100 python tools/dtwbench.py) > gpurun_out/r02_final/attn_dtw_bench.log 2>/dev/null
SVB_BENCH_SHAPES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> gpurun_out/r02_final/bench_shapes.err
cd /tmp
SVB_BENCH_MARKERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r02 --output-format csv -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-side-stream > /root/repo/gpurun_out/r02_final/bench_under_rocprof.json 2> /root/repo/gpurun_out/r02_final/bench_under_rocprof.err
python /root/repo/tools/trace_summary.py /tmp/prof/r02_kernel_trace.csv 20 70 > /root/repo/gpurun_out/r02_final/kernel_summary.txt
cp /tmp/prof/r02_kernel_stats.csv /root/repo/gpurun_out/r02_final/kernel_stats.csv
for c in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc1; timeout 200 rocprofv3 --pmc $c -d /tmp/pmc1 --output-format csv -- python /root/repo/tools/pmc_conv.py 32 192 384 1124 5 fwd 2 > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py $(find /tmp/pmc1 -name "*counter_collection.csv" | head -1) svb_conv1d > /root/repo/gpurun_out/r02_final/pmc_sq_conv_192_384_k5_B32_128x96.txt 2>&1
done
cd /root/repo
tail -3 gpurun_out/r02_final/pytest_gpu.log; cat gpurun_out/r02_final/smoke.log | tail -3; grep "ms/step" gpurun_out/r02_final/*.err; head -12 gpurun_out/r02_final/kernel_summary.txt; cat gpurun_out/r02_final/pmc_sq_conv_192_384_k5_B32_128x96.txt; tail -2 gpurun_out/r02_final/pmc_traffic.log | cut -c1-300

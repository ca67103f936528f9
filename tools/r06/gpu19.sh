#!/bin/bash
# r06 call 19: tile table with the 16-tap GEMM kernel, vocoder step traced with it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "gemm_kernel_with_taps or pointwise_gemm" 2>&1 | tail -3
timeout 1500 python tools/tune_tiles.py --out gpurun_out/tile_table.json > gpurun_out/r06_tile_tuner_taps16.log 2>&1
tail -2 gpurun_out/r06_tile_tuner_taps16.log
cp gpurun_out/tile_table.json neuralsvb_amd/tile_table.json
cd /tmp; rm -rf /tmp/prof_voc
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_voc -o r06 --output-format csv -- \
   python $R/bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r06_voc_taps16.json 2> $R/gpurun_out/r06_voc_taps16.err
python $R/tools/trace_summary.py /tmp/prof_voc/r06_kernel_trace.csv 4 70 > $R/gpurun_out/r06_kernel_summary_vocoder_taps16.txt
cd $R
timeout 600 python bench.py --workload vocoder --no-cpu-baseline 2>&1 >/dev/null | grep "ms/step"
timeout 600 python bench.py --workload infer --no-cpu-baseline 2>&1 >/dev/null | grep "ms/step"

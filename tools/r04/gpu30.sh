#!/bin/bash
# round 4, GPU call 30: wide weight-gradient tiles at one workgroup per CU (128x128 single-tap, 128x64 five-tap): parity, per-shape
# A/B (instrumentation build), step A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels.py -q -m gpu -x -k "wgrad" > gpurun_out/r04_g30_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g30_pytest.log; tail -2 gpurun_out/r04_g30_pytest.log
SVB_LIB=instr timeout 300 python tools/wgbench.py > gpurun_out/r04_g30_wgbench_wide.log 2>&1
SVB_LIB=instr SVB_WG_WIDE_TOGGLE=1 timeout 300 python tools/wgbench.py > gpurun_out/r04_g30_wgbench_narrow.log 2>&1
paste <(cut -c1-52 gpurun_out/r04_g30_wgbench_wide.log) <(cut -c35-52 gpurun_out/r04_g30_wgbench_narrow.log) | grep -v amdgpu
for v in "" "1" "" "1"; do
  SVB_WG_WIDE_TOGGLE=$v timeout 600 python tools/bench_instr.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline > gpurun_out/r04_g30_bench_$v.log 2>&1
  echo "toggle=[$v]: $(grep 'ms/step\|issuing' gpurun_out/r04_g30_bench_$v.log | cut -c18-90 | head -2 | tr '\n' '|')"
done

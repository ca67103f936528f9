#!/bin/bash
# round 4, GPU call 19: XCD-aware work ids in the weight-gradient kernel: parity, per-shape A/B, step A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels.py -q -m gpu -x -k "wgrad" > gpurun_out/r04_g19_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g19_pytest.log; tail -2 gpurun_out/r04_g19_pytest.log
SVB_LIB=instr timeout 300 python tools/wgbench.py > gpurun_out/r04_g19_wgbench_xcd.log 2>&1
SVB_LIB=instr SVB_WG_NO_XCD=1 timeout 300 python tools/wgbench.py > gpurun_out/r04_g19_wgbench_identity.log 2>&1
paste <(cut -c1-62 gpurun_out/r04_g19_wgbench_xcd.log) <(cut -c35-52 gpurun_out/r04_g19_wgbench_identity.log) | grep -v amdgpu
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline > gpurun_out/r04_g19_bench.log 2>&1
grep 'ms/step\|issuing' gpurun_out/r04_g19_bench.log | cut -c18-90 | tr '\n' '|'

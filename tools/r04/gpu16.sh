#!/bin/bash
# round 4, GPU call 16: full GPU suite + smoke + the default bench line (intermediate, secured)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_g16_pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g16_pytest_gpu_full.log
tail -4 gpurun_out/r04_g16_pytest_gpu_full.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_g16_smoke.log 2>&1
tail -3 gpurun_out/r04_g16_smoke.log
timeout 900 python bench.py > gpurun_out/r04_g16_bench.json 2> gpurun_out/r04_g16_bench.log
grep "ms/step\|issuing\|roofline\|phase 3\|N=1 step\|split\|extra\|cpu" gpurun_out/r04_g16_bench.log | cut -c1-260

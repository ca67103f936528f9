#!/bin/bash
# round 4, GPU call 9: host-side changes (fused critic-tower node, no-autograd conv fast path) -- parity tests + bench A/B
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_modules_disc.py tests/test_step_golden.py -q -m gpu -x > gpurun_out/r04_g9_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g9_pytest.log
tail -4 gpurun_out/r04_g9_pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline > gpurun_out/r04_g9_bench_$i.log 2>&1
grep "ms/step\|issuing" gpurun_out/r04_g9_bench_$i.log | cut -c1-200
done

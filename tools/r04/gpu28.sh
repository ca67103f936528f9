#!/bin/bash
# round 4, GPU call 28: PMC traffic of the dominant conv over 16 shapes (the fused projections changed the launch population), then
# the default bench line with roofline.traffic
mkdir -p gpurun_out/r04_final
export PYTHONUNBUFFERED=1 TMPDIR=/tmp SVB_ROUND=r04
SVB_PMC_SHAPES=16 timeout 700 python tools/pmc_traffic.py > gpurun_out/r04_final/pmc_traffic.log 2>&1
cp profiles/r04_pmc_traffic.json gpurun_out/r04_final/ 2>/dev/null
timeout 600 python bench.py > gpurun_out/r04_final/bench_train_bf16x3.json 2> gpurun_out/r04_final/bench_train_bf16x3.log
grep "ms/step" gpurun_out/r04_final/bench_train_bf16x3.log | cut -c1-160
grep -o '"traffic": [^,]*, "traffic_unit": "[^"]*", "traffic_source": "[^"]*"' gpurun_out/r04_final/bench_train_bf16x3.json

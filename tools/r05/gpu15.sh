#!/bin/bash
# round 5, GPU call 15: A/B of the gated multi-tap weight-gradient instantiations at one workgroup per CU (accumulators in AGPRs, no
# spills) against two per CU with 36 ... 200 bytes of scratch per lane: the vocoder step.
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r05_g15
for i in 1 2; do
timeout 300 python bench.py --workload vocoder --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r05_g15/voc_$i.json 2> gpurun_out/r05_g15/voc_$i.log
python -c "import json; d=json.loads([l for l in open('gpurun_out/r05_g15/voc_$i.json') if l.startswith('{')][-1]); print('vocoder', round(d['ms_per_step'],2))"
done

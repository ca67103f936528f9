#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ddp_gloo.py -x -q -m gpu -k "rccl" > gpurun_out/r06_rccl_test.log 2>&1
tail -4 gpurun_out/r06_rccl_test.log
timeout 900 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r06_bench_vl2.json 2> gpurun_out/r06_bench_vl2.log
grep -E "variable|failed" gpurun_out/r06_bench_vl2.log | head -20
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_vl2.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['extra_workloads']['variable_length'])
PY

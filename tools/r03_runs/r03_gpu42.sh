#!/bin/bash
# grouped 16-row weight-gradient kernel: parity on the MI355X, vocoder step A/B (SVB_WG_NO_G16=1), per-shape table
O=gpurun_out/r03ai
mkdir -p $O
export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_kernels.py tests/test_modules_hifigan.py tests/test_hifigan_task.py -m gpu -x -q 2>&1 | tail -3) > $O/pytest.log
tail -2 $O/pytest.log
for envs in "X=1" "SVB_WG_NO_G16=1" "X=1" "SVB_WG_NO_G16=1"; do
  echo "== vocoder bench [$envs]: $(env $envs timeout 300 python bench.py --workload vocoder --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2))')"
done > $O/ab.log 2>&1
cat $O/ab.log
SVB_BENCH_SHAPES=1 SVB_BENCH_SHAPES_TOP=400 timeout 400 python bench.py --workload vocoder --steps 3 --warmup 3 --no-cpu-baseline --no-side-stream > $O/voc.json 2> $O/voc_shapes.log
grep "'wgrad'" $O/voc_shapes.log | grep -v ", 1, [0-9]*, [0-9]*, [0-9]*, [0-9]*)" | cut -c18-170 | head -16

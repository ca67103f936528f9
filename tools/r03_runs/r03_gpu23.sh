#!/bin/bash
# vocoder workload (configs[2]) after: MPD on the reference's [B,C,H,p] planes (dilation-p convs + row space-to-depth), grouped
# weight gradients with several groups per tile, wide-dilation tap groups
O=gpurun_out/r03x
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_kernels.py tests/test_functional.py tests/test_modules_hifigan.py tests/test_hifigan_task.py tests/test_vocoder_plugin.py -m gpu -x -q 2>&1 | tail -5) > $O/pytest.log
cat $O/pytest.log
for envs in "X=1" "SVB_WG_NO_GROUP_PACK=1"; do
  echo "== vocoder bench [$envs]: $(env $envs timeout 300 python bench.py --workload vocoder --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), "ms/step; conv", d["roofline"]["all_conv_kernels"])')"
done > $O/ab.log 2>&1
cat $O/ab.log
SVB_BENCH_SHAPES=1 timeout 400 python bench.py --workload vocoder --steps 3 --warmup 3 --no-cpu-baseline --no-side-stream > $O/voc.json 2> $O/voc_shapes.log
grep -A45 "per-shape" $O/voc_shapes.log | cut -c18-170

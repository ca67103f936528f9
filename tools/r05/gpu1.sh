#!/bin/bash
# round 5, GPU call 1: (a) offline tile tuner -> tile table, (b) the new bf16x3 / published-shape vocoder parity tests with
# their measured deviations, (c) default bench line WITH the fresh table.
mkdir -p gpurun_out/r05_g1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r05_g1
timeout 900 python tools/tune_tiles.py --out $O/tile_table.json > $O/tune.log 2>&1
echo "tune rc=$?"; tail -5 $O/tune.log
timeout 1200 python -m pytest tests/test_vocoder_shapes.py tests/test_modules_hifigan.py tests/test_vocoder_plugin.py tests/test_hifigan_task.py -m gpu -q -s -p no:cacheprovider > $O/parity.log 2>&1
echo "parity rc=$?"; grep -E "^\[(fp32|bf16x3)\]|passed|failed|Error|assert" $O/parity.log | tail -60
if [ -s $O/tile_table.json ]; then cp $O/tile_table.json neuralsvb_amd/tile_table.json; fi
timeout 900 python bench.py > $O/bench.json 2> $O/bench.log
echo "bench rc=$?"; cat $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_step_median')}, d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('tile_table'))
print({k:(v.get('ms_per_step'), v.get('roofline',{}).get('frac')) for k,v in d['extra_workloads'].items()})
print(d['step_split'])"

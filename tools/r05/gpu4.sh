#!/bin/bash
# round 5, GPU call 4: tile table regenerated with the two 32x256 configurations, arbiter-bounded step golden, 300-step
# trajectories (tool + test), bench with the new table, per-shape conv times of the vocoder workload.
O=gpurun_out/r05_g4
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python tools/tune_tiles.py --out $O/tile_table.json > $O/tune.log 2>&1
echo "tune rc=$?"; tail -3 $O/tune.log
if [ -s $O/tile_table.json ]; then cp $O/tile_table.json neuralsvb_amd/tile_table.json; fi
python - <<'PY'
import json,collections
d=json.load(open("neuralsvb_amd/tile_table.json"))
print("choices by cfg:", sorted(collections.Counter(d["choices"].values()).items()))
PY
timeout 600 python -m pytest tests/test_step_golden.py tests/test_kernels.py -m gpu -q -p no:cacheprovider -k "training_steps or narrow_channel or direct_tiles_whole or clip_edges or tile_table" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
timeout 600 python tools/train_trajectory.py --steps 300 --out $O/trajectory_fp32_vs_bf16x3.json > $O/trajectory.log 2>&1
echo "trajectory rc=$?"; tail -14 $O/trajectory.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.log
echo "bench rc=$?"; grep -h "ms/step\|settled" $O/bench.log | tail -8
SVB_BENCH_SHAPES=1 SVB_BENCH_SHAPES_TOP=60 timeout 300 python bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > $O/vocoder_shapes.json 2> $O/vocoder_shapes.log
echo "vocoder rc=$?"; grep -A40 "per-shape conv time" $O/vocoder_shapes.log | cut -c1-150 | head -45

#!/bin/bash
# round 5, GPU call 13: self-gated conv / weight-gradient variants -- parity tests, tile table regenerated (the gated signatures'
# timings changed), default bench line (vocoder / inference under extra_workloads).
O=gpurun_out/r05_g13
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules_hifigan.py tests/test_vocoder_shapes.py tests/test_hifigan_task.py -m gpu -q -p no:cacheprovider -k "self_gated or generator or resblock or vocoder_b64 or infer_t1872 or vocoder_training_step or single_product" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_oldtable.json 2> $O/bench_oldtable.log
echo "bench (old table) rc=$?"; grep -h "ms/step" $O/bench_oldtable.log | tail -4
timeout 900 python tools/tune_tiles.py --out $O/tile_table.json > $O/tune.log 2>&1
echo "tune rc=$?"; tail -2 $O/tune.log
if [ -s $O/tile_table.json ]; then cp $O/tile_table.json neuralsvb_amd/tile_table.json; fi
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.log
echo "bench (new table) rc=$?"; grep -h "ms/step" $O/bench.log | tail -4

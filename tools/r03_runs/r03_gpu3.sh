#!/bin/bash
# Round 3, GPU call 3: suite at HEAD (new: bench-shape gradients, spec2wav plugin, binarizer, critic general shapes, new tiles),
# A/B of the PPG side stream and the capped deferred reduce, the default bench line with its extra workloads.
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > $O/pytest_gpu.log
bash tools/ab_bench.sh "" "overlap_ppg_encoder=False" "defer_wgrad_reduce=True" "overlap_ppg_encoder=False,defer_wgrad_reduce=True" "" > $O/ab.log 2>&1
cp gpurun_out/ab/*.err $O/ 2>/dev/null
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -5 $O/pytest_gpu.log; cat $O/ab.log; cat $O/bench_default.time; grep "ms/step\|extra\|cpu" $O/bench_default.err | tail -20

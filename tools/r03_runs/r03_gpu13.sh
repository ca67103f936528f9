#!/bin/bash
O=gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_modules_vae.py tests/test_step_golden.py tests/test_task_step.py -m gpu -x -q -k "graph_replay or training_steps or fused_step" 2>&1 | tail -12) > $O/pytest_sel.log
tail -5 $O/pytest_sel.log
bash tools/ab_bench.sh "" "ppg_graph_replay=False" "" "ppg_graph_replay=False" > $O/ab.log 2>&1
cat $O/ab.log

#!/bin/bash
O=gpurun_out/r03i
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --deselect "tests/test_step_golden.py::test_training_steps_match_reference_task[bf16x3]" 2>&1 | tail -25) > $O/pytest_gpu.log
bash tools/ab_bench.sh "" "wn_stack_executor=False" "defer_wgrad_reduce=False" "wn_stack_executor=False,defer_wgrad_reduce=False,overlap_ppg_encoder=False" "" "wn_stack_executor=False" > $O/ab.log 2>&1
tail -6 $O/pytest_gpu.log; cat $O/ab.log

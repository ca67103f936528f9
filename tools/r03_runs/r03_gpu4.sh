#!/bin/bash
# diagnostic: which change moved the bf16x3 step golden?
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_kernels.py tests/test_q_layout.py -m gpu -q -k "tile_configs or direct_tiles or wide_tiles or q_" 2>&1 | tail -15) > $O/new_tiles.log
(SVB_NCFG_Q=10 timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -k "bf16x3" 2>&1 | grep -v Warning | tail -60) > $O/step_ncfg10.log
(SVB_PPG_SIDE=0 timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -k "bf16x3" 2>&1 | grep -v Warning | tail -60) > $O/step_noppg.log
(SVB_NCFG_Q=10 SVB_PPG_SIDE=0 timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -k "bf16x3" 2>&1 | grep -v Warning | tail -60) > $O/step_both_off.log
(timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -k "bf16x3" 2>&1 | grep -v Warning | grep "^step\|MISMATCH\|passed\|failed" | head -80) > $O/step_default.log
(timeout 900 python -m pytest tests -m gpu -q --deselect "tests/test_step_golden.py::test_training_steps_match_reference_task[bf16x3]" 2>&1 | tail -40) > $O/pytest_rest.log
for f in new_tiles step_ncfg10 step_noppg step_both_off; do echo "== $f"; tail -4 $O/$f.log; done; echo "== default"; head -30 $O/step_default.log; echo "== rest"; tail -15 $O/pytest_rest.log

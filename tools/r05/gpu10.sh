#!/bin/bash
# round 5, GPU call 10: the well-conditioned fused-vs-elementwise step test (measured worst errors per pass).
O=gpurun_out/r05_g10
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_task_step.py -m gpu -q -s -p no:cacheprovider -k "fused_step_paths" > $O/tests.log 2>&1
echo "rc=$?"; grep -E "fused vs element-wise|passed|failed|Error" $O/tests.log | head -20

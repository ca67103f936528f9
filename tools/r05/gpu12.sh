#!/bin/bash
# round 5, GPU call 12: PPG prefetch as a replayed hipGraph -- parity test and A/B of the step (host issue, ms/step).
O=gpurun_out/r05_g12
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_task_step.py -m gpu -q -p no:cacheprovider -k "ppg" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/tests.log
bash tools/ab_bench.sh "" "ppg_graph=True" "" "ppg_graph=True"

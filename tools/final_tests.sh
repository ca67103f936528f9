#!/bin/bash
# the round's closing GPU call: full suite with -x, then smoke()      usage: bash tools/final_tests.sh [round tag, default r06]
R=${1:-r06}
O=gpurun_out/${R}_final_tests
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full.log 2>&1
tail -4 $O/pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -2 $O/smoke.log

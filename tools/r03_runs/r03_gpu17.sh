#!/bin/bash
O=gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
bash tools/ab_bench.sh "" "" "" > $O/ab.log 2>&1
cat $O/ab.log

#!/bin/bash
O=gpurun_out/r03l
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log

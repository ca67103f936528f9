#!/bin/bash
O=gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
SVB_DIAG_DUMP=$O timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -s -k "fp32 or bf16x3" 2>&1 | grep "^step 2\|passed\|failed" | cut -c1-70
ls -la $O

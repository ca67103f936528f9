#!/bin/bash
O=gpurun_out/r03g
mkdir -p $O
export TMPDIR=/tmp
(SVB_DIAG_TABLE=1 timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -s -k "bf16x3" 2>&1 | grep "^GRAD 2 gen\|^GRAD 1 gen" ) > $O/table_bf16.log
(SVB_DIAG_TABLE=1 timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -s -k "fp32" 2>&1 | grep "^GRAD 2 gen" ) > $O/table_fp32.log
wc -l $O/*.log

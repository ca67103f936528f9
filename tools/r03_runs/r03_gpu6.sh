#!/bin/bash
O=gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; (env "$@" timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -s -k "bf16x3" 2>&1 | grep "^step\|passed\|failed\|Error" | cut -c1-60) > $O/$name.log; echo "== $name"; cat $O/$name.log; }
run wgrad_fp32 SVB_DIAG_WGRAD_FP32=1
run critic_fp32 SVB_DIAG_CRITIC_FP32=1
run gen_fp32 SVB_DIAG_GEN_FP32=1
run critic_and_wgrad_fp32 SVB_DIAG_CRITIC_FP32=1 SVB_DIAG_WGRAD_FP32=1

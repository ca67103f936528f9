#!/bin/bash
# diagnostic 2: is the step-2 generator gradient of the bf16x3 step golden a stream race or arithmetic sensitivity?
O=gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; (env "$@" timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -s -k "$K" 2>&1 | grep "^step\|passed\|failed" | cut -c1-60) > $O/$name.log; echo "== $name"; cat $O/$name.log; }
K="fp32" run fp32_default X=1
K="bf16x3" run bf16_serial SVB_STEP_EXTRA_HPARAMS=wgrad_side_stream=False,overlap_critic_pass=False,overlap_ppg_encoder=False,defer_wgrad_reduce=False,wn_stack_executor=False
K="bf16x3" run bf16_no_critic_overlap SVB_STEP_EXTRA_HPARAMS=overlap_critic_pass=False
K="bf16x3" run bf16_no_side SVB_STEP_EXTRA_HPARAMS=wgrad_side_stream=False
K="bf16x3" run bf16_default X=1
K="fp32" run fp32_serial SVB_STEP_EXTRA_HPARAMS=wgrad_side_stream=False,overlap_critic_pass=False,overlap_ppg_encoder=False,defer_wgrad_reduce=False,wn_stack_executor=False

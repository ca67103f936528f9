#!/bin/bash
O=gpurun_out/r03h
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; (env "$@" timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -s -k "bf16x3" 2>&1 | grep "^step . gen\|pinned\|passed\|failed\|Error" | cut -c1-60) > $O/$name.log; echo "== $name"; cat $O/$name.log; }
run pool_fp32 SVB_DIAG_LAYER_FP32=vae_model.encoder.poolings,vae_model.encoder.out_proj
run pool_only SVB_DIAG_LAYER_FP32=vae_model.encoder.poolings
run enc_fp32 SVB_DIAG_LAYER_FP32=vae_model.encoder
run dec_fp32 SVB_DIAG_LAYER_FP32=vae_model.decoder
run pool_fp32_wgrad SVB_DIAG_LAYER_FP32=vae_model.encoder.poolings,vae_model.encoder.out_proj SVB_DIAG_WGRAD_FP32=1

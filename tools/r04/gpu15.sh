#!/bin/bash
# round 4, GPU call 15: step goldens over four draw sets (fp32 + bf16x3); seed 20260926 with the encoder's pooling tail pinned to fp32
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_step_golden.py -q -m gpu -s -k "training_steps" > gpurun_out/r04_g15_step_seeds.log 2>&1
echo "rc=$?" >> gpurun_out/r04_g15_step_seeds.log
grep -E "worst grad-norm|passed|failed|rc=|MISMATCH" gpurun_out/r04_g15_step_seeds.log | cut -c1-110 | grep -v "disc:\|map:" | head -70
echo "---- seed 20260926, bf16x3, encoder pooling tail + latent head in fp32"
SVB_DIAG_LAYER_FP32=vae_model.encoder.poolings,vae_model.encoder.out_proj timeout 600 python -m pytest tests/test_step_golden.py -q -m gpu -s -k "training_steps and bf16x3 and 20260926" > gpurun_out/r04_g15_step_seed_fp32tail.log 2>&1
grep -E "pinned|worst grad-norm|passed|failed" gpurun_out/r04_g15_step_seed_fp32tail.log | cut -c1-110 | grep -v "disc:\|map:"

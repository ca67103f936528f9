#!/bin/bash
# r06 call 18: streaming kernels (LayerNorm NCT 64-column tiles, SSIM / mel-loss 16-byte LDS passes): parity + ewbench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "ssim or mel_loss or layernorm_nct" 2>&1 | tail -4
timeout 200 python tools/ewbench.py > gpurun_out/r06_streaming_kernels_3.log 2>&1
grep -E "layernorm|mel_loss|ssim" gpurun_out/r06_streaming_kernels_3.log

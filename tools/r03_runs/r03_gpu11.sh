#!/bin/bash
O=gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
cp tests/golden/step_ref.json /tmp/keep.json; cp tests/golden/step_ref_draws.npz /tmp/keep.npz
for sd in 11 12 13; do
  cp tests/golden/_scan/$sd/step_ref.json tests/golden/step_ref.json; cp tests/golden/_scan/$sd/step_ref_draws.npz tests/golden/step_ref_draws.npz
  echo "== seed $sd"
  timeout 300 python -m pytest tests/test_step_golden.py -m gpu -q -s -k "fp32 or bf16x3" 2>&1 | grep "^step . gen\|^step . disc\|^step . map\|passed\|failed" | cut -c1-62
done
cp /tmp/keep.json tests/golden/step_ref.json; cp /tmp/keep.npz tests/golden/step_ref_draws.npz

#!/bin/bash
# round 5, GPU call 3: fp64-arbitrated gradient bounds (step golden on four seeds, B=2 fixture), 300-step fp32-vs-bf16x3 training
# trajectories, the new HIP tails (latent map, speaker projection, upsampling) through the step golden / functional tests.
O=gpurun_out/r05_g3
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_step_golden.py -m gpu -q -s -p no:cacheprovider > $O/step_golden_arbiter.log 2>&1
echo "step golden rc=$?"; grep -E "arbiter|e_hip|passed|failed" $O/step_golden_arbiter.log | head -150
timeout 600 python -m pytest tests/test_modules_vae.py -m gpu -q -s -p no:cacheprovider > $O/modules_vae.log 2>&1
echo "modules_vae rc=$?"; grep -E "e_hip|passed|failed|worst" $O/modules_vae.log | head -30
timeout 600 python tools/train_trajectory.py --steps 300 --out $O/trajectory_fp32_vs_bf16x3.json > $O/trajectory.log 2>&1
echo "trajectory rc=$?"; tail -16 $O/trajectory.log
timeout 600 python -m pytest tests/test_functional.py tests/test_task_step.py -m gpu -q -p no:cacheprovider -k "upsample or flat_adamw or trains_like or period_s2d" > $O/misc.log 2>&1
echo "misc rc=$?"; tail -5 $O/misc.log

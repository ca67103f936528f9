#!/bin/bash
# round 5, GPU call 5: bf16 single-product mode (kernel test, mel-L1 at the bench shape, secondary bench line), fused ResBlock node
# (parity + vocoder A/B), the 300-step trajectory test, default bench line.
O=gpurun_out/r05_g5
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py tests/test_modules_vae.py tests/test_modules_hifigan.py tests/test_task_step.py tests/test_vocoder_shapes.py tests/test_hifigan_task.py -m gpu -q -s -p no:cacheprovider \
  -k "single_product or bench_shape_mel or fused_resblock or trains_like or vocoder_b64 or vocoder_training_step" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -E "^(bf16|fp32|bf16x3) \{|passed|failed|l1a2a|a2a_f|\[bf16x3\] B=64" $O/tests.log | head -20
for v in "" "fused_resblock=False"; do
  ex=""; [ -n "$v" ] && ex="--extra-hparams $v"
  timeout 300 python bench.py --workload vocoder --steps 6 --warmup 3 --no-cpu-baseline $ex > $O/voc_${v:-default}.json 2> $O/voc_${v:-default}.log
  echo "vocoder [${v:-default}]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/voc_${v:-default}.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2),'ms/step')")"
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.log
echo "bench rc=$?"; grep -h "ms/step\|settled\|secondary" $O/bench.log | tail -9

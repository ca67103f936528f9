#!/bin/bash
# r06 call 10: vocoder step with the multi-tensor feature loss and the 8192-element WeightNorm reduce window; vocoder parity tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py --workload vocoder --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r06_voc1.json 2> gpurun_out/r06_voc1.log
grep -E "ms/step|vocoder" gpurun_out/r06_voc1.log | tail -3
timeout 900 python -m pytest tests/test_modules_hifigan.py tests/test_hifigan_task.py tests/test_vocoder_shapes.py tests/test_functional.py tests/test_kernels.py -x -q -m gpu -k "hifigan or vocoder or feature_loss or long_rows or discriminators or generator" > gpurun_out/r06_voc_tests.log 2>&1
tail -4 gpurun_out/r06_voc_tests.log

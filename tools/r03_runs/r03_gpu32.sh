#!/bin/bash
O=gpurun_out/r03ad
mkdir -p $O
export TMPDIR=/tmp
(timeout 500 python -m pytest tests/test_kernels.py tests/test_functional.py tests/test_modules_vae.py tests/test_step_golden.py tests/test_task_step.py -m gpu -x -q 2>&1 | tail -3) > $O/pytest.log
tail -2 $O/pytest.log
timeout 100 python tools/ewbench.py 2>&1 | grep "mel_loss\|gn_relu" > $O/ew.log; cat $O/ew.log
for i in 1 2 3; do
  echo "== bench: $(timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads 2>&1 >/dev/null | grep -h 'ms/step' | sed 's/\[bench [0-9:]*\] //' | tr '\n' ';')"
done > $O/ab.log 2>&1
cat $O/ab.log

#!/bin/bash
# r06 call 14: period conv node + A-only gated weight gradients: vocoder parity tests, vocoder step, per-queue split
cd "$GRAFT_REPO_ROOT"
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modules_hifigan.py tests/test_hifigan_task.py tests/test_vocoder_shapes.py tests/test_functional.py tests/test_kernels.py tests/test_vocoder_plugin.py -x -q -m gpu -k "hifigan or vocoder or period or wgrad or discriminators or generator or infer" > gpurun_out/r06_voc_tests2.log 2>&1
tail -4 gpurun_out/r06_voc_tests2.log
timeout 900 python bench.py --workload vocoder --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r06_voc2.json 2> gpurun_out/r06_voc2.log
python -c "
import json
d=json.loads(open('gpurun_out/r06_voc2.json').read().strip().splitlines()[-1]); print('vocoder ms/step', d['ms_per_step'])"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_voc
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_voc -o r06 --output-format csv -- \
   python $R/bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > /dev/null 2> /dev/null
python $R/tools/trace_summary.py /tmp/prof_voc/r06_kernel_trace.csv 4 70 > $R/gpurun_out/r06_vocoder_kernel_summary2.txt
head -5 $R/gpurun_out/r06_vocoder_kernel_summary2.txt
grep -A3 "per Queue_Id" $R/gpurun_out/r06_vocoder_kernel_summary2.txt

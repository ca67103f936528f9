#!/bin/bash
# round 4, GPU call 5: multi-stream step graph: parity with eager steps, bench eager / step graph / per-pass graphs
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_task_step.py -q -m gpu -x -k "hipgraph" > gpurun_out/r04_g5_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g5_pytest.log
tail -15 gpurun_out/r04_g5_pytest.log
for mode in eager graph; do
  flag=""; [ $mode = graph ] && flag="--graph"
  timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline $flag > gpurun_out/r04_g5_bench_$mode.log 2>&1
  grep "ms/step\|issuing" gpurun_out/r04_g5_bench_$mode.log | cut -c1-200
done
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra-workloads --no-roofline --graph --extra-hparams hip_graph_mode=pass > gpurun_out/r04_g5_bench_graph_pass.log 2>&1
grep "ms/step\|issuing" gpurun_out/r04_g5_bench_graph_pass.log | cut -c1-200
tail -5 gpurun_out/r04_g5_bench_graph.log | cut -c1-300

#!/bin/bash
# A round's measurement artefacts of the benchmarked configuration (one GPU call, ~8 min): default bench line (with the
# extra workloads), kernel traces with and without the side streams, PMC traffic of the dominant kernel, SQ counters of two shapes.
#   gpurun --timeout 1500 -- 'bash tools/measure_round.sh r05'          (copy what should be judged from gpurun_out/ to profiles/)
# One-off A/B runs of step variants: tools/ab_bench.sh "<hparams overrides | bench flag>" ...   (both scripts are parameterised:
# the 50 single-purpose scripts of round 3 are gone; git history has them)
set -x
TAG=${1:-r06}
O=gpurun_out/${TAG}_final
mkdir -p $O
export TMPDIR=/tmp
export SVB_ROUND=$TAG
R=$PWD
# (PMC traffic first: the bench line's roofline.traffic reads profiles/<tag>_pmc_traffic.json of THIS commit's kernels)
SVB_PMC_SHAPES=16 timeout 700 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1
cp profiles/${TAG}_pmc_traffic.json $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench_train_bf16x3.json 2> $O/bench_train_bf16x3.log
SVB_BENCH_SHAPES=1 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra-workloads > /dev/null 2> $O/conv_per_shape.log
cd /tmp
for v in default noside; do
  fl=""; [ $v = noside ] && fl="--no-side-stream --extra-hparams overlap_critic_pass=False,overlap_ppg_encoder=False"
  rm -rf /tmp/prof_$v
  SVB_BENCH_MARKERS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o $TAG --output-format csv -- \
     python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads $fl \
     > $R/$O/bench_under_rocprof_$v.json 2> $R/$O/bench_under_rocprof_$v.err
  python $R/tools/trace_summary.py /tmp/prof_$v/${TAG}_kernel_trace.csv 20 80 > $R/$O/kernel_summary_$v.txt
  cp /tmp/prof_$v/${TAG}_kernel_stats.csv $R/$O/kernel_stats_$v.csv 2>/dev/null
done
rm -rf /tmp/prof_voc
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_voc -o $TAG --output-format csv -- \
   python $R/bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > $R/$O/bench_under_rocprof_vocoder.json 2> $R/$O/bench_under_rocprof_vocoder.err
python $R/tools/trace_summary.py /tmp/prof_voc/${TAG}_kernel_trace.csv 4 60 > $R/$O/kernel_summary_vocoder.txt
(cd $R && timeout 100 python tools/ewbench.py > $O/streaming_kernels.log 2>&1)
C="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
# (shape + direction + forced tile: 2 = the 128x96 tap-table tile of the k5 layers, 21 = the 64x128 GEMM-form tile of the 1-tap convs)
for shp in "32 192 384 1124 5 fwd 2" "32 192 384 281 5 fwd 2" "32 256 256 562 1 fwd 21" "32 192 384 281 1 fwd 21"; do
  nm=$(echo $shp | cut -d' ' -f1-5 | tr ' ' '_')
  rm -rf /tmp/pmc1; timeout 200 rocprofv3 --pmc $C -d /tmp/pmc1 --output-format csv -- python $R/tools/pmc_conv.py $shp > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pmc1 -name "*counter_collection.csv" | head -1) svb_conv1d > $R/$O/pmc_sq_conv_$nm.txt 2>&1
done
cd $R
grep "ms/step" $O/*.log $O/*.err | cut -c1-200; head -8 $O/kernel_summary_default.txt; cat $O/pmc_sq_conv_*.txt; tail -2 $O/pmc_traffic.log | cut -c1-300

#!/bin/bash
# Round 3, GPU call 2: (1) does a workgroup's MFMA stage shrink when it has the CU alone / with wave priorities? (stage stamps),
# (2) kernel-time totals of the epilogue-fusion variants under rocprof (the wall clock of call 1 was host-bound on that box).
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for envs in "" "SVB_LDS_PAD_KB=70" "SVB_CONV_PRIO=1" "SVB_CONV_PRIO=2" "SVB_LDS_PAD_KB=70 SVB_CONV_PRIO=1"; do
  for shape in "32 192 384 1124 5 2" "32 192 384 281 5 2" "32 192 384 1124 1 2" "32 256 256 1124 5 2"; do
    echo "== env [$envs] shape [$shape]"
    env $envs timeout 60 python tools/stage_timing.py $shape 2>&1 | grep -v amdgpu.ids
  done
done > $O/stage_timing.log 2>&1
cd /tmp
i=0
for v in "" "wn_fuse_res_skip=True" "wn_fuse_gate=True" "wn_fuse_res_skip=True,wn_fuse_gate=True"; do
  i=$((i+1))
  if [[ -n "$v" ]]; then extra="--extra-hparams $v"; else extra=""; fi
  rm -rf /tmp/prof$i
  SVB_BENCH_MARKERS=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof$i -o r03 --output-format csv -- \
     python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-side-stream $extra \
     > $R/$O/fuse_v$i.json 2> $R/$O/fuse_v$i.err
  python $R/tools/trace_summary.py /tmp/prof$i/r03_kernel_trace.csv 20 40 > $R/$O/fuse_v$i.summary.txt
  echo "variant $i [$v]: $(head -2 $R/$O/fuse_v$i.summary.txt | tr '\n' ' ')"
done > $R/$O/fuse_variants.log 2>&1
cd $R
cat $O/stage_timing.log; cat $O/fuse_variants.log

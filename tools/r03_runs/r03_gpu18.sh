#!/bin/bash
O=gpurun_out/r03r
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -rf /tmp/prof_v
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o voc --output-format csv -- python $R/bench.py --workload vocoder --steps 6 --warmup 3 --no-cpu-baseline > $R/$O/bench_vocoder.json 2> $R/$O/bench_vocoder.err
python - <<'PY' > $R/$O/vocoder_kernel_stats.txt
import csv,sys
rows=list(csv.DictReader(open('/tmp/prof_v/voc_kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:120]}")
PY
cd $R
grep "ms/step" $O/bench_vocoder.err; head -40 $O/vocoder_kernel_stats.txt

#!/bin/bash
# r06 call 7: kernel trace of the benchmarked step (default streams) at the pointwise-kernel commit; CSV kept for a timeline view
cd "$GRAFT_REPO_ROOT"
R=$PWD
mkdir -p gpurun_out/r06_trace
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_d
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r06 --output-format csv -- \
   python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra-workloads \
   > $R/gpurun_out/r06_trace/bench_under_rocprof.json 2> $R/gpurun_out/r06_trace/bench_under_rocprof.err
python $R/tools/trace_summary.py /tmp/prof_d/r06_kernel_trace.csv 20 90 > $R/gpurun_out/r06_trace/kernel_summary_default.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/prof_d/r06_kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
marks=[i for i,r in enumerate(rows) if "sleep" in r["Kernel_Name"].lower() or "spin_kernel" in r["Kernel_Name"]]
rows=rows[marks[0]+1:marks[1]]
n=len(rows)
# keep ~3 steps from the middle
a=n*8//20; b=n*11//20
import os
with open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r06_trace/trace_3steps.csv","w") as f:
    w=csv.writer(f); w.writerow(["start","end","queue","kernel"])
    for r in rows[a:b]:
        w.writerow([r["Start_Timestamp"],r["End_Timestamp"],r["Queue_Id"],r["Kernel_Name"][:90]])
PY
head -12 $R/gpurun_out/r06_trace/kernel_summary_default.txt
grep "ms/step" $R/gpurun_out/r06_trace/bench_under_rocprof.err | head -3

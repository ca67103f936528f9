#!/bin/bash
# r06 call 13: which weight-gradient launches of the vocoder step run on the compute queue?
cd "$GRAFT_REPO_ROOT"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_voc
SVB_BENCH_MARKERS=1 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_voc -o r06 --output-format csv -- \
   python $R/bench.py --workload vocoder --steps 4 --warmup 3 --no-cpu-baseline > /dev/null 2> /dev/null
python - <<'PY' > $R/gpurun_out/r06_voc_queue_split.txt
import csv, collections
rows=list(csv.DictReader(open('/tmp/prof_voc/r06_kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
marks=[i for i,r in enumerate(rows) if "sleep" in r["Kernel_Name"].lower() or "spin_kernel" in r["Kernel_Name"]]
rows=rows[marks[0]+1:marks[1]]
acc=collections.defaultdict(lambda:[0,0])
for r in rows:
    nm=r["Kernel_Name"].split("(")[0][:70]
    k=(r["Queue_Id"], nm, r.get("Grid_Size","")+"/"+r.get("Workgroup_Size",""))
    acc[k][0]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); acc[k][1]+=1
for (q,nm,g),(d,c) in sorted(acc.items(), key=lambda kv:-kv[1][0])[:90]:
    print(f"q{q} {d/4e6:8.3f} ms/step {c/4:6.1f} x {d/c/1e3:8.1f} us  grid {g:>14s}  {nm}")
PY
head -60 $R/gpurun_out/r06_voc_queue_split.txt

#!/bin/bash
# effective shader clock during the dominant conv: GRBM_GUI_ACTIVE (cycles the GPU was active) against the kernel's wall time
O=gpurun_out/r03ao
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for shp in "32 192 384 1124 5" "32 192 384 281 5"; do
  nm=$(echo $shp | tr ' ' '_')
  rm -rf /tmp/pg; timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d /tmp/pg --output-format csv -- python $R/tools/pmc_conv.py $shp fwd 2 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pg -name "*counter_collection.csv" | head -1) svb_conv1d > $R/$O/grbm_$nm.txt 2>&1
  rm -rf /tmp/pk; timeout 200 rocprofv3 --kernel-trace -d /tmp/pk -o kt --output-format csv -- python $R/tools/pmc_conv.py $shp fwd 2 > /dev/null 2>&1
  python - <<PY >> $R/$O/grbm_$nm.txt
import csv
rows=[r for r in csv.DictReader(open("/tmp/pk/kt_kernel_trace.csv")) if "svb_conv1d_bf16x3_kernel" in r["Kernel_Name"]]
d=sorted(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
print("kernel-trace durations (ns):", d, "median", d[len(d)//2])
PY
  cat $R/$O/grbm_$nm.txt
done

"""Summarise a rocprofv3 counter_collection CSV: per kernel name, mean of each counter over its dispatches."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kname, ctrs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in kname:
        continue
    print(kname)
    for c, v in sorted(ctrs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")

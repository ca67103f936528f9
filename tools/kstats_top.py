"""Top kernels of a rocprofv3 --stats kernel_stats.csv.   python tools/kstats_top.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}% {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:110]}")

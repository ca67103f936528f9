"""Summarise a rocprofv3 *_kernel_stats.csv per training step:  python tools/kstats.py <csv> <n_steps> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time {tot / 1e6:.1f} ms = {tot / 1e6 / n:.2f} ms/step over {n:g} steps")
groups = {}
for r in rows:
    nm = r["Name"]
    key = ("svb" if "svb_" in nm else "miopen/ck" if any(s in nm.lower() for s in ("miopen", "naive_conv", "im2d2col", "col2im", "igemm", "ck::", "_zn2ck", "batched_transpose", "subtensor")) else
           "rocblas/hipblaslt" if nm.startswith("Cijk") else "torch")
    groups[key] = groups.get(key, 0.0) + float(r["TotalDurationNs"])
print({k: round(v / 1e6 / n, 2) for k, v in groups.items()}, "ms/step")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / n:8.3f} ms/step {int(r['Calls']) / n:7.1f} calls/step "
          f"{float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:120]}")

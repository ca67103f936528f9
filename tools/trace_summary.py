"""Steady-state per-step kernel table from a rocprofv3 --kernel-trace CSV of `SVB_BENCH_MARKERS=1 python bench.py ...`:
only the dispatches between the two marker (spin) kernels that bracket bench.py's timed region are counted, so warm-up,
autotuning and MIOpen's find passes are excluded and sum(kernel time) <= wall.

  python tools/trace_summary.py <kernel_trace.csv> <steps> [top]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"] or "sleep" in r["Kernel_Name"].lower()]
if len(marks) >= 2:
    rows = rows[marks[0] + 1:marks[1]]
    note = "between the two marker kernels"
else:
    note = "NO MARKERS FOUND: whole trace"
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
acc = defaultdict(lambda: [0, 0])
busy = 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc[r["Kernel_Name"]][0] += d
    acc[r["Kernel_Name"]][1] += 1
    busy += d


def group(nm):
    low = nm.lower()
    if "svb_" in nm:
        return "svb (hand-written HIP)"
    if nm.startswith("Cijk") or "rocblas" in low:
        return "rocBLAS / hipBLASLt"
    if "miopen" in low or "batchnorm" in low:
        return "MIOpen"
    if "rccl" in low or "nccl" in low:
        return "RCCL"
    return "torch (ATen)"


groups = defaultdict(lambda: [0, 0])
for nm, (d, c) in acc.items():
    groups[group(nm)][0] += d
    groups[group(nm)][1] += c
wall = t1 - t0
print(f"{len(rows)} dispatches {note}; span {wall / 1e6:.2f} ms = {wall / 1e6 / steps:.2f} ms/step over {steps:g} steps; "
      f"sum of kernel durations {busy / 1e6 / steps:.2f} ms/step ({100.0 * busy / wall:.1f} % of the span); "
      f"{len(rows) / steps:.0f} launches/step")
# time with at least one kernel running (union of the dispatch intervals): span - union = the device sat idle (host-bound gaps,
# launch latency); union < sum = kernels of different streams overlapped
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s_, e_ in iv[1:]:
    if s_ > cur_e:
        union += cur_e - cur_s
        gaps.append(s_ - cur_e)
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
union += cur_e - cur_s
big = sum(g_ for g_ in gaps if g_ > 20000)
print(f"device busy (union of kernel intervals) {union / 1e6 / steps:.2f} ms/step, idle {(wall - union) / 1e6 / steps:.2f} ms/step in "
      f"{len(gaps) / steps:.0f} gaps/step (of which gaps > 20 us: {big / 1e6 / steps:.2f} ms/step)")
for g, (d, c) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print(f"  {g:26s} {d / 1e6 / steps:7.3f} ms/step {c / steps:7.1f} launches/step")
print(f"{'ms/step':>9} {'calls/step':>10} {'avg us':>8}  kernel")
for nm, (d, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{d / 1e6 / steps:9.3f} {c / steps:10.1f} {d / c / 1e3:8.1f}  {nm[:110]}")
# how many kernels are in flight, as a share of the span (a sweep over the interval endpoints): where the step runs on ONE kernel the
# chip is as full as that kernel's grid makes it; the streams only help where the count is >= 2
ev = sorted([(a_, 1) for a_, _ in iv] + [(b_, -1) for _, b_ in iv])
depth, last, hist = 0, ev[0][0], defaultdict(int)
for t_, d_ in ev:
    hist[min(depth, 4)] += t_ - last
    last = t_
    depth += d_
print("kernels in flight (share of the span): " + ", ".join(f"{k if k < 4 else '4+'}: {100.0 * hist[k] / wall:.1f} %" for k in sorted(hist)))
# ... and the same restricted to the time the LONGEST-running class of kernels is alone: which kernels run without company
solo = defaultdict(int)
act, last = [], ev[0][0]
ev2 = sorted([(int(r["Start_Timestamp"]), 1, r["Kernel_Name"]) for r in rows] + [(int(r["End_Timestamp"]), -1, r["Kernel_Name"]) for r in rows],
             key=lambda e_: (e_[0], e_[1]))
for t_, d_, nm in ev2:
    if len(act) == 1:
        solo[act[0]] += t_ - last
    last = t_
    if d_ == 1:
        act.append(nm)
    else:
        act.remove(nm)
print("alone on the device, ms/step (top 8): " + "; ".join(f"{v / 1e6 / steps:.2f} {k.split('(')[0][-48:]}" for k, v in sorted(solo.items(), key=lambda kv: -kv[1])[:8]))
# per-queue view (a HIP stream maps to a hardware queue): busy time, launches, idle gaps inside the queue's own span
qcol = next((c for c in ("Queue_Id", "Stream_Id", "queue_id") if c in rows[0]), None)
if qcol:
    byq = defaultdict(list)
    for r in rows:
        byq[r[qcol]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    print(f"per {qcol}:")
    for q, ivs in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        ivs.sort()
        b = sum(e - s for s, e, _ in ivs)
        gaps_q = [ivs[i + 1][0] - ivs[i][1] for i in range(len(ivs) - 1) if ivs[i + 1][0] > ivs[i][1]]
        small = sum(g for g in gaps_q if g <= 20000)
        print(f"  queue {q}: busy {b / 1e6 / steps:7.3f} ms/step, {len(ivs) / steps:6.1f} launches/step, gaps <= 20 us "
              f"{small / 1e6 / steps:6.3f} ms/step ({len([g for g in gaps_q if g <= 20000]) / steps:.0f}/step), gaps > 20 us "
              f"{(sum(gaps_q) - small) / 1e6 / steps:6.3f} ms/step")
    # where the busiest queue (the compute stream) waits: its gaps > 20 us grouped by the kernels on either side -- a wait for
    # another stream, a host stall or a step boundary each leave a recognisable pair
    q, ivs = max(byq.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))
    ivs.sort()
    pairs = defaultdict(lambda: [0, 0])

    def short(nm):
        nm = nm.replace("void ", "")
        return nm[:nm.index("(")][:58] if "(" in nm else nm[:58]
    for i in range(len(ivs) - 1):
        g = ivs[i + 1][0] - ivs[i][1]
        if g > 20000:
            k = (short(ivs[i][2]), short(ivs[i + 1][2]))
            pairs[k][0] += g
            pairs[k][1] += 1
    print(f"queue {q}: gaps > 20 us by (kernel before -> kernel after), ms/step, count/step, avg us")
    for (a, b2), (g, c) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {g / 1e6 / steps:6.3f} {c / steps:5.1f} {g / c / 1e3:7.1f}  {a}  ->  {b2}")

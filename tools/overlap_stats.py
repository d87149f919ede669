"""Concurrency statistics of the timed steps of a rocprofv3 kernel trace: python tools/overlap_stats.py <kernel_trace.csv> [skip_front_fraction]
prints span, union-busy time, sum of kernel durations, time with >= 2 kernels running, per-queue busy time."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pre = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"]]
# the last 8 steps (4 preprocess launches per step)
lo, hi = pre[-4 * 8 - 1], pre[-1]
step = rows[lo:hi]
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
live = 0; prev = ev[0][0]; busy = 0; multi = 0
for t, d in ev:
    if live >= 1: busy += t - prev
    if live >= 2: multi += t - prev
    prev = t; live += d
span = ev[-1][0] - ev[0][0]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
q = defaultdict(int)
for r in step: q[r["Queue_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
n = 8
print(f"per step: span {span / n / 1e6:.3f} ms | union busy {busy / n / 1e6:.3f} | idle {(span - busy) / n / 1e6:.3f} | sum of durations {tot / n / 1e6:.3f} | >= 2 kernels {multi / n / 1e6:.3f} | kernels {len(step) // n}")
print("   busy per queue (ms/step):", {k: round(v / n / 1e6, 2) for k, v in sorted(q.items())})

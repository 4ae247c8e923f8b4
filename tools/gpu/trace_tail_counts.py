#!/usr/bin/env python3
"""Kernel-name histogram of the LAST fraction of a rocprofv3 kernel trace (the steady-state graph replays of a bench run):
python tools/gpu/trace_tail_counts.py <kernel_trace.csv> [fraction=0.3] [steps_in_tail]"""
import csv, sys, collections
path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tail = rows[int(len(rows) * (1 - frac)):]
cnt = collections.Counter(); tim = collections.Counter()
for s, e, n in tail:
    n = n.replace("(anonymous namespace)::", "")[:90]
    cnt[n] += 1; tim[n] += e - s
span = (tail[-1][1] - tail[0][0]) / 1e6
print(f"tail: {len(tail)} dispatches over {span:.2f} ms")
for n, c in cnt.most_common(25):
    print(f"{c:8d} {tim[n] / 1e6:9.3f} ms {tim[n] / c / 1e3:7.2f} us  {n}")

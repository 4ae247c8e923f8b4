#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: wall time covered by at least one kernel (union), by >= 2 kernels at once, and
the plain sum of kernel durations -- how much of a multi-stream step actually overlaps.

    python tools/trace_overlap.py <kernel_trace.csv> [skip_first_ms]
"""
import csv
import sys


def main():
    path = sys.argv[1]
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    ev.sort()
    t_first = ev[0][0]
    skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
    ev = [e for e in ev if e[0] - t_first >= skip]
    pts = []
    for a, b in ev:
        pts.append((a, 1))
        pts.append((b, -1))
    pts.sort()
    depth, last = 0, pts[0][0]
    busy1 = busy2 = 0
    for t, d in pts:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += d
        last = t
    total = sum(b - a for a, b in ev)
    span = ev[-1][1] - ev[0][0]
    print(f"kernels {len(ev)}  span {span / 1e6:.2f} ms  sum of durations {total / 1e6:.2f} ms  "
          f">=1 kernel {busy1 / 1e6:.2f} ms ({100 * busy1 / span:.1f} % of span)  >=2 kernels {busy2 / 1e6:.2f} ms ({100 * busy2 / span:.1f} %)")


if __name__ == "__main__":
    main()

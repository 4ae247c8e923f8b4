#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_trace.csv) into a per-kernel
table (calls, total, avg, min, max, share) -- the text committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_x/run_results.db [steps] > profiles/rNN_x.txt
"""
import csv
import sqlite3
import sys


def rows_from_db(path):
    c = sqlite3.connect(path)
    return c.execute("select name, end-start from kernels").fetchall()


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print(f"# source: {path}   dispatches: {len(rows)}   total kernel time: {tot / 1e6:.3f} ms"
          + (f"   ({tot / 1e6 / steps:.3f} ms per step over {steps} steps incl. warm-up)" if steps else ""))
    print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:100]:100s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.1f} {a[2] / 1e3:9.1f} {a[3] / 1e3:9.1f} {100 * a[1] / tot:6.2f}")


if __name__ == "__main__":
    main()

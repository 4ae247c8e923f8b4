#!/usr/bin/env python3
"""Run-length view of a rocprofv3 --kernel-trace CSV: consecutive launches of the same kernel (and grid) as one line with their
count and median / min duration -- for a program that launches layer after layer (tools/bench_conv.py) this is the
per-layer GPU time free of the host's launch overhead.

    python tools/trace_runs.py <kernel_trace.csv> [min_count]
"""
import csv
import re
import statistics
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         (r.get("Grid_Size_X") or r.get("Grid_Size") or "?"), r.get("Workgroup_Size_X") or "?"))
    rows.sort()
    min_count = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    i = 0
    while i < len(rows):
        j = i
        while j < len(rows) and rows[j][2:] == rows[i][2:]:
            j += 1
        d = [(b - a) / 1e3 for a, b, *_ in rows[i:j]]
        if j - i >= min_count:
            try:
                blocks = int(rows[i][3]) // int(rows[i][4])
            except ValueError:
                blocks = -1
            print(f"{rows[i][2][:70]:70s} x{j - i:3d}  blocks {blocks:6d}  median {statistics.median(d):8.1f} us  min {min(d):8.1f} us")
        i = j


if __name__ == "__main__":
    main()

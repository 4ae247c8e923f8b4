#!/usr/bin/env python3
"""Per-kernel means of the counters of one or more `rocprofv3 --pmc` passes (counter_collection.csv files), the SQ ones also
as a share of SQ_WAVE_CYCLES where that pass has it.   python tools/pmc_layer_report.py <dir with *counter_collection.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    files = sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True))
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0][-40:]
                a = agg[name][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for k, cs in agg.items():
        print(k)
        wc = cs.get("SQ_WAVE_CYCLES")
        for c, (v, n) in sorted(cs.items()):
            line = f"   {c:34s} {v / max(n, 1):16.1f}  (x{n})"
            if wc and wc[0] and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES":
                line += f"   {100.0 * (v / n) / (wc[0] / wc[1]):7.2f} % of SQ_WAVE_CYCLES"
            print(line)


if __name__ == "__main__":
    main()

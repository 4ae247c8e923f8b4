#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel class from ONE rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE):

    util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)      (GRBM_GUI_ACTIVE sums the 8 XCDs)

Both counters come from the same dispatches of the same pass (joined by Dispatch_Id; a dispatch that lacks either is dropped).
Round 4 also printed a "clock" = GRBM_GUI_ACTIVE / duration with the durations of a --kernel-trace pass: two passes with different
launch mixes, and GRBM_GUI_ACTIVE reads high on short dispatches (MI355X_MICROARCH.md) -- 2.5-2.9 GHz on a 2.4 GHz part.  The
column is gone; the clock under load is sampled by bench.py itself (dct_clock_probe: s_memtime over s_memrealtime).

    python tools/pmc_mfma.py <pmc_counter_collection.csv>
"""
import csv
import sys


def cls(name):
    for k in ("igemm3m", "igemm3p", "igemm3", "igemm2", "wgrad3", "wgrad2"):
        if k + "_kernel" in name:
            return k
    return None


def main():
    pmc = sys.argv[1]
    per = {}                      # dispatch id -> [class, mfma busy, gui active]
    with open(pmc) as f:
        for r in csv.DictReader(f):
            k = cls(r["Kernel_Name"])
            if k is None:
                continue
            d = per.setdefault(r["Dispatch_Id"], [k, None, None])
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                d[1] = float(r["Counter_Value"])
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                d[2] = float(r["Counter_Value"])
    agg = {}
    for k, busy, gui in per.values():
        if busy is None or gui is None or gui <= 0:
            continue
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += busy; a[2] += gui
    print(f"{'kernel':10s} {'launches':>8s} {'MFMA pipe busy':>15s}")
    for k, (n, busy, gui) in sorted(agg.items()):
        print(f"{k:10s} {n:8d} {100 * busy / (gui / 8.0 * 1024.0):14.1f}%")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""MFMA-pipe utilisation and effective shader clock per kernel class from a rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE) joined with a --kernel-trace pass of the same command:

    util  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)      (GRBM_GUI_ACTIVE sums the 8 XCDs)
    clock = GRBM_GUI_ACTIVE / 8 / kernel duration                               (MI355X_MICROARCH.md, DVFS give-back)

    python tools/pmc_mfma.py <pmc_counter_collection.csv> <kernel_trace.csv>
"""
import csv
import sys


def cls(name):
    for k in ("igemm3m", "igemm3p", "igemm3", "igemm2", "wgrad3", "wgrad2"):
        if k + "_kernel" in name:
            return k
    return None


def main():
    pmc, trace = sys.argv[1], sys.argv[2]
    agg = {}
    with open(pmc) as f:
        for r in csv.DictReader(f):
            k = cls(r["Kernel_Name"])
            if k is None:
                continue
            a = agg.setdefault(k, {"SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0, "n": 0})
            if r["Counter_Name"] in a:
                a[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    a["n"] += 1
    dur = {}
    with open(trace) as f:
        for r in csv.DictReader(f):
            k = cls(r["Kernel_Name"])
            if k is None:
                continue
            d = dur.setdefault(k, [0, 0.0])
            d[0] += 1
            d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"{'kernel':10s} {'launches':>8s} {'MFMA pipe busy':>15s} {'clock (GHz)':>12s}")
    for k, a in sorted(agg.items()):
        if not a["GRBM_GUI_ACTIVE"]:
            continue
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        clk = float("nan")
        if k in dur and dur[k][0]:
            clk = (cyc / a["n"]) / (dur[k][1] / dur[k][0])          # cycles per launch / ns per launch = GHz
        print(f"{k:10s} {a['n']:8d} {100 * util:14.1f}% {clk:12.2f}")


if __name__ == "__main__":
    main()

"""Per-kernel instruction mix from a rocprofv3 --pmc pass (SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY): instructions per wave and per MFMA, share of wave cycles parked / stalled at issue.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
        --kernel-trace -d out -o p --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-graph --single-stream ...
    python tools/pmc_kernel_mix.py out/.../p_counter_collection.csv
"""
import csv
import sys
from collections import defaultdict


def main():
    rows = csv.DictReader(open(sys.argv[1]))
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    out = []
    for k, c in acc.items():
        waves = c.get("SQ_WAVES", 0) or 1
        mf = c.get("SQ_INSTS_MFMA", 0)
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        out.append((c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0), k, len(disp[k]), waves, mf, c, wc))
    out.sort(reverse=True)
    print(f"{'kernel':70s} {'disp':>5s} {'valu/w':>8s} {'salu/w':>8s} {'mfma/w':>7s} {'lds/w':>7s} {'valu/mfma':>9s} {'salu/mfma':>9s} {'parked%':>8s} {'issue-stall%':>12s}")
    for _, k, nd, waves, mf, c, wc in out[:40]:
        v, s, l = c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0)
        vm = f"{v / mf:9.2f}" if mf else "        -"
        sm = f"{s / mf:9.2f}" if mf else "        -"
        print(f"{k[:70]:70s} {nd:5d} {v / waves:8.0f} {s / waves:8.0f} {mf / waves:7.0f} {l / waves:7.0f} {vm} {sm} "
              f"{100 * c.get('SQ_WAIT_ANY', 0) / wc:8.1f} {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:12.1f}")


if __name__ == "__main__":
    main()

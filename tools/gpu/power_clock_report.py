"""Condenses the rocm-smi samples of tools/gpu/power_clock.sh: per field, the idle value, and min / median / max while the bench ran."""
import json
import re
import statistics
import sys


def main():
    raw, js = sys.argv[1], sys.argv[2]
    t = None
    start = end = None
    rows = []          # (time, field, value)
    for line in open(raw):
        line = line.strip()
        if line.startswith("t "):
            t = float(line.split()[1])
            continue
        if line.startswith("bench-start"):
            start = float(line.split()[1])
            continue
        if line.startswith("bench-end"):
            end = float(line.split()[1])
            continue
        if ":" not in line or t is None:
            continue
        head, _, val = line.rpartition(":")
        field = head.split(":")[-1].strip()
        m = re.search(r"\(?(-?[0-9]+(?:\.[0-9]+)?)\s*([A-Za-z%]*)", val)
        if "clk" in field and "(" in val:                      # "sclk clock level: 1 (2400Mhz)"
            m = re.search(r"\(([0-9.]+)\s*([A-Za-z]*)\)", val)
        if not m:
            continue
        rows.append((t, field, float(m.group(1)), m.group(2)))
    try:
        d = json.loads(open(js).read())
        print(f"bench: {d['config']['workload'][:60]}  {d['ms_per_step']:.4f} ms/step over {d['steps']} steps x {d.get('timed_regions', {}).get('count', 1)} regions")
    except Exception as e:      # noqa: BLE001
        print("bench line unreadable:", e)
    fields = sorted({r[1] for r in rows})
    # the timed part: skip the first 60 % of the bench window (imports, set-up, capture)
    lo = start + 0.6 * (end - start) if start and end else None
    for f in fields:
        unit = next(r[3] for r in rows if r[1] == f)
        idle = [r[2] for r in rows if r[1] == f and start and r[0] < start]
        run = [r[2] for r in rows if r[1] == f and lo and lo <= r[0] <= end]
        if not run:
            continue
        print(f"{f:55s} idle {statistics.median(idle) if idle else float('nan'):9.1f}   running: min {min(run):9.1f}  median {statistics.median(run):9.1f}  max {max(run):9.1f} {unit}  ({len(run)} samples)")


if __name__ == "__main__":
    main()

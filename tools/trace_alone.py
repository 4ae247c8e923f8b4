#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a multi-stream step: per kernel, the wall time during which it is the ONLY kernel in
flight ("alone") and the time it shares with others -- which kernels the streams do not overlap.

    python tools/trace_alone.py <kernel_trace.csv> [skip_first_ms | -last_ms]
"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]


def main():
    ev = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    ev.sort()
    t0 = ev[0][0]
    skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
    if skip < 0:                      # negative: only the last |skip| ms of the trace (the timed steps of a bench run)
        t_end = ev[-1][1]
        ev = [e for e in ev if t_end - e[0] <= -skip]
    else:
        ev = [e for e in ev if e[0] - t0 >= skip]
    pts = []
    for i, (a, b, n) in enumerate(ev):
        pts.append((a, 1, i))
        pts.append((b, 0, i))
    pts.sort()
    live = set()
    last = pts[0][0]
    alone = defaultdict(int)
    shared = defaultdict(int)
    idle = 0
    for t, kind, i in pts:
        dt = t - last
        if dt > 0:
            if len(live) == 1:
                alone[ev[next(iter(live))][2]] += dt
            elif len(live) > 1:
                for j in live:
                    shared[ev[j][2]] += dt
            else:
                idle += dt
        if kind:
            live.add(i)
        else:
            live.discard(i)
        last = t
    span = ev[-1][1] - ev[0][0]
    tot_alone = sum(alone.values())
    print(f"span {span / 1e6:.2f} ms; no kernel in flight {idle / 1e6:.2f} ms ({100 * idle / span:.1f} %); exactly one kernel {tot_alone / 1e6:.2f} ms ({100 * tot_alone / span:.1f} %)")
    print(f"{'kernel':62s} {'alone ms':>9s} {'% of span':>9s} {'shared ms':>10s}")
    for n in sorted(set(alone) | set(shared), key=lambda k: -alone[k])[:30]:
        print(f"{n:62s} {alone[n] / 1e6:9.3f} {100 * alone[n] / span:9.2f} {shared[n] / 1e6:10.3f}")


if __name__ == "__main__":
    main()

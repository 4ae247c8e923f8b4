#!/usr/bin/env python3
"""How the captured step is actually scheduled: from a rocprofv3 --kernel-trace CSV of `bench.py`, the kernels of ONE replayed step in start
order with their start / end offsets and how many other kernels were running when each started; plus the distribution of the concurrency
depth over the step and the queue ids the kernels ran on.

    python tools/trace_step_schedule.py <kernel_trace.csv> [step_index_from_the_end=3] [max_rows=80]
"""
import csv
import re
import sys


def short(name):
    m = re.search(r"(igemm3m|igemm3p|igemm2|igemm|wgrad3|wgrad2|wgrad_reduce|splitk_epilogue|adam_kernel|adam_advance|maxpool\w*|bilinear\w*|stem\w*|head\w*|"
                  r"dropout\w*|pack\w*|relu_bits|bias_partial|partial_reduce|split_dw_db|slab_rows_fold|ce_\w+|jsd_\w+|softmax\w*|finalize\w*|CatArray|copyBuffer)", name)
    return m.group(1) if m else name[:24]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows_max = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    ev.sort()
    # steps are delimited by the Adam launches: a step ends with the last adam_kernel of a pair
    adam_ends = [e[1] for e in ev if e[2] == "adam_kernel"]
    # two adam kernels per step (two models): boundaries at every second one
    bounds = adam_ends[1::2]
    if len(bounds) < back + 1:
        print("not enough steps in the trace"); return
    t0, t1 = bounds[-back - 1], bounds[-back]
    step = [e for e in ev if t0 < e[0] <= t1 or (e[0] <= t1 and e[1] > t0 and e[0] > t0)]
    print(f"step of {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels, queues {sorted(set(e[3] for e in step))}")
    depth_time = {}
    pts = sorted([(e[0], 1) for e in step] + [(e[1], -1) for e in step])
    d, last = 0, pts[0][0]
    for t, k in pts:
        depth_time[d] = depth_time.get(d, 0) + (t - last)
        d += k; last = t
    tot = sum(depth_time.values())
    print("concurrency depth: " + "  ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(depth_time.items())))
    print(f"{'start us':>9s} {'dur us':>7s} {'running':>7s} {'q':>3s}  kernel")
    for i, e in enumerate(step[:rows_max]):
        running = sum(1 for o in step if o[0] < e[0] < o[1])
        print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:7.1f} {running:7d} {e[3]:>3s}  {e[2]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Upper bound for a grouped two-model launch of the UNet's deep levels (missing #3) without building it.

A grouped launch (both models' identical layer in one grid) would give a deep-level convolution twice the blocks at an unsplit K.
Its best case is ONE launch over a batch of 2 x 16 images (same tile count, and -- optimistically -- one set of weights instead of
two).  Per layer, forward and data gradient, microseconds:

    pair    two launches of B = 16 with different weights on two streams (what the step's two model streams issue)
    serial  the same two launches on one stream
    b32     one launch of B = 32 (the grouped launch's upper bound)

    python tools/bench_grouped.py [--reps 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import dct_amd  # noqa: E402,F401
from dct_amd import hip_ops as K  # noqa: E402
from bench_conv import CONVS, plan_note  # noqa: E402

DEV = "cuda:0"
DEEP = ("dec4a", "dec4b", "cen_a", "cen_b", "enc4a", "enc4b", "enc3a", "enc3b")


def time_us(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dt = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    print(f"{'layer':7s} {'pass':6s} {'pair':>8s} {'serial':>8s} {'b32':>8s}  b32 / pair   planner (B = 16 | B = 32)")
    tot = {"pair": 0.0, "serial": 0.0, "b32": 0.0}
    for name, cin, hin, cout in CONVS:
        if name not in DEEP:
            continue
        ho = hin - 2
        mk = lambda *s: torch.randn(*s, device=DEV, generator=g)
        x = [mk(16, hin, hin, cin).to(dt) for _ in range(2)]
        x32 = torch.cat(x, 0)
        w = [(mk(cout, 3, 3, cin) / (3 * cin ** 0.5)).to(dt) for _ in range(2)]
        wd = [(mk(cin, 3, 3, cout) / (3 * cout ** 0.5)).to(dt) for _ in range(2)]
        bias = mk(cout)
        y = [torch.empty(16, ho, ho, cout, device=DEV, dtype=dt) for _ in range(2)]
        y32 = torch.empty(32, ho, ho, cout, device=DEV, dtype=dt)
        dy = [mk(16, ho, ho, cout).to(dt) for _ in range(2)]
        dy32 = torch.cat(dy, 0)
        dx = [torch.empty(16, hin, hin, cin, device=DEV, dtype=dt) for _ in range(2)]
        dx32 = torch.empty(32, hin, hin, cin, device=DEV, dtype=dt)

        def fwd(i):
            K.conv2d(x[i], w[i], bias, y[i], relu=True)

        def dgr(i):
            K.conv2d(dy[i], wd[i], None, dx[i], pad_h=2, pad_w=2, mask=x[i])

        for what, one, big in (("fwd", fwd, lambda: K.conv2d(x32, w[0], bias, y32, relu=True)),
                               ("dgrad", dgr, lambda: K.conv2d(dy32, wd[0], None, dx32, pad_h=2, pad_w=2, mask=x32))):
            def pair():
                s1.wait_stream(cur); s2.wait_stream(cur)
                with torch.cuda.stream(s1):
                    one(0)
                with torch.cuda.stream(s2):
                    one(1)
                cur.wait_stream(s1); cur.wait_stream(s2)

            def serial():
                one(0); one(1)
            one(0); n16 = plan_note()
            big(); n32 = plan_note()
            tp, ts, tb = time_us(pair, args.reps), time_us(serial, args.reps), time_us(big, args.reps)
            tot["pair"] += tp; tot["serial"] += ts; tot["b32"] += tb
            print(f"{name:7s} {what:6s} {tp:8.1f} {ts:8.1f} {tb:8.1f}  {tb / tp:10.2f}   {n16.split(':')[0]} | {n32.split(':')[0]}")
    print(f"TOTAL          {tot['pair']:8.1f} {tot['serial']:8.1f} {tot['b32']:8.1f}  {tot['b32'] / tot['pair']:10.2f}")


if __name__ == "__main__":
    main()

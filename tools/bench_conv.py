#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv GEMM kernels on the UNet layer shapes of cfg2
(256x256 input, batch B): forward, data-gradient and weight-gradient, TFLOP/s per layer and
FLOP-weighted totals.  Knobs (dct_tune_set) let one process A/B kernel variants.

    python tools/bench_conv.py [--batch 16] [--reps 20] [--what fwd,dgrad,wgrad] [--ab]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dct_amd  # noqa: E402,F401
from dct_amd import _lib, hip_ops as K  # noqa: E402

DEV = "cuda:0"

# (name, Cin, Hin, Cout) for the 3x3 valid convs; convT as (name, Cin, Hin, Cout) 2x2 s2
CONVS = [("dec1b", 64, 254, 64), ("dec2a", 64, 126, 128), ("dec2b", 128, 124, 128), ("dec3a", 128, 61, 256),
         ("dec3b", 256, 59, 256), ("dec4a", 256, 29, 512), ("dec4b", 512, 27, 512), ("cen_a", 512, 13, 1024),
         ("cen_b", 1024, 11, 1024), ("enc4a", 1024, 18, 512), ("enc4b", 512, 16, 512), ("enc3a", 512, 28, 256),
         ("enc3b", 256, 26, 256), ("enc2a", 256, 48, 128), ("enc2b", 128, 46, 128), ("enc1a", 128, 88, 64),
         ("enc1b", 64, 86, 64)]
CONVT = [("cenT", 1024, 9, 512), ("enc4T", 512, 14, 256), ("enc3T", 256, 24, 128), ("enc2T", 128, 44, 64)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


PLAN = False             # --plan: the planner's choice per launch (dct_debug_last_plan)


def plan_note():
    import ctypes
    f = _lib.load().dct_debug_last_plan
    f.restype = ctypes.c_char_p
    return f().decode()


TRAINER_FORM = True      # the launches as the trainer makes them: forward leaves ReLU-gate bits, the data gradient masks by such bits
                         # (1/16 of the bytes of the activation), the weight gradient carries the bias gradient (--plain: none of these)


def run(B, reps, what, label, only=None, use_mask=True):
    dt = torch.bfloat16
    rows = []
    tot = {k: [0.0, 0.0] for k in ("fwd", "dgrad", "wgrad")}
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, cin, hin, cout in CONVS:
        if only and name not in only:
            continue
        ho = hin - 2
        x = torch.randn(B, hin, hin, cin, device=DEV, generator=g).to(dt)
        w = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).to(dt)
        wd = (torch.randn(cin, 3, 3, cout, device=DEV, generator=g) / (3 * cout ** 0.5)).to(dt)
        bias = torch.randn(cout, device=DEV, generator=g)
        y = torch.empty(B, ho, ho, cout, device=DEV, dtype=dt)
        dy = torch.randn(B, ho, ho, cout, device=DEV, generator=g).to(dt)
        dx = torch.empty(B, hin, hin, cin, device=DEV, dtype=dt)
        dw = torch.zeros(cout * 9 * cin, device=DEV)
        fl = 2.0 * B * ho * ho * 9 * cin * cout
        r = {"name": name, "flops": fl}
        ybits = K.relu_bits_like(y) if TRAINER_FORM else None
        xbits = K.relu_bits_like(x) if TRAINER_FORM and use_mask else None
        db = torch.zeros(cout, device=DEV) if TRAINER_FORM else None
        if "fwd" in what:
            t = timeit(lambda: K.conv2d(x, w, bias, y, relu=True, relu_bits_out=ybits), reps)
            r["fwd"] = t; tot["fwd"][0] += fl; tot["fwd"][1] += t; r["fwd_plan"] = plan_note()
        if "dgrad" in what:
            t = timeit(lambda: K.conv2d(dy, wd, None, dx, pad_h=2, pad_w=2, mask=x if use_mask else None, mask_bits=xbits), reps)
            r["dgrad"] = t; tot["dgrad"][0] += fl; tot["dgrad"][1] += t; r["dgrad_plan"] = plan_note()
        if "wgrad" in what:
            t = timeit(lambda: K.conv2d_wgrad(dy, x, dw, accumulate=True, db=db), reps)
            r["wgrad"] = t; tot["wgrad"][0] += fl; tot["wgrad"][1] += t; r["wgrad_plan"] = plan_note()
        rows.append(r)
    for name, cin, hin, cout in CONVT:
        if only and name not in only:
            continue
        x = torch.randn(B, hin, hin, cin, device=DEV, generator=g).to(dt)
        wf = (torch.randn(4 * cout, cin, device=DEV, generator=g) / cin ** 0.5).to(dt)
        wd = (torch.randn(cin, 4 * cout, device=DEV, generator=g) / cout ** 0.5).to(dt)
        bias = torch.randn(cout, device=DEV, generator=g)
        y = torch.empty(B, 2 * hin, 2 * hin, cout, device=DEV, dtype=dt)
        dy = torch.randn(B, 2 * hin, 2 * hin, cout, device=DEV, generator=g).to(dt)
        dx = torch.empty(B, hin, hin, cin, device=DEV, dtype=dt)
        dw = torch.zeros(cin * 4 * cout, device=DEV)
        fl = 2.0 * B * hin * hin * 4 * cin * cout
        r = {"name": name, "flops": fl}
        if "fwd" in what:
            t = timeit(lambda: K.conv2d(x, wf, bias, y, R=1, S=1, relu=True, scatter2x2=True), reps)
            r["fwd"] = t; tot["fwd"][0] += fl; tot["fwd"][1] += t; r["fwd_plan"] = plan_note()
        if "dgrad" in what:
            t = timeit(lambda: K.conv2d(dy, wd, None, dx, R=2, S=2, stride=2, mask=x if use_mask else None), reps)
            r["dgrad"] = t; tot["dgrad"][0] += fl; tot["dgrad"][1] += t; r["dgrad_plan"] = plan_note()
        if "wgrad" in what:
            t = timeit(lambda: K.conv2d_wgrad(x, dy, dw, R=2, S=2, stride=2, accumulate=True), reps)
            r["wgrad"] = t; tot["wgrad"][0] += fl; tot["wgrad"][1] += t; r["wgrad_plan"] = plan_note()
        rows.append(r)
    print(f"--- {label}  (B={B}; us / TFLOP/s)")
    for r in rows:
        s = f"{r['name']:7s} {r['flops'] / 1e9:8.2f} GF"
        for k in ("fwd", "dgrad", "wgrad"):
            if k in r:
                s += f" | {k} {r[k] * 1e6:8.1f} {r['flops'] / r[k] / 1e12:7.1f}"
        print(s)
    for k, (f, t) in tot.items():
        if t > 0:
            print(f"TOTAL {k}: {t * 1e3:.3f} ms, {f / t / 1e12:.1f} TFLOP/s")
    if PLAN:
        print("--- planner's choice per launch")
        for r in rows:
            for k in ("fwd", "dgrad", "wgrad"):
                if k in r:
                    print(f"{r['name']:7s} {k:5s} {r[k] * 1e6:7.1f} us {r['flops'] / r[k] / 1e12:6.1f} TF  {r[k + '_plan']}")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--only", default="", help="comma-separated layer names")
    ap.add_argument("--ab-packed", action="store_true", help="A/B the packed-rows shared-halo kernel (knob 10) on the deep levels")
    ap.add_argument("--ab-wgrad", action="store_true", help="A/B the filter-row weight-gradient kernel (knob 8) against the per-tap kernel")
    ap.add_argument("--ab", action="store_true", help="A/B the shared-halo 3x3 kernel against the per-tap kernel, interleaved in one process")
    ap.add_argument("--ab-knob", action="append", default=[], metavar="KNOB=V0,V1[,...]",
                    help="A/B any dct_tune_set knob: interleaved rounds of the listed values in one process (repeatable)")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--plain", action="store_true", help="no gate bits / bias gradient riding along (the launches of rounds 1-3 of this table)")
    ap.add_argument("--plan", action="store_true", help="also print which kernel / grid the planner chose for every launch")
    ap.add_argument("--no-mask", action="store_true", help="data gradients without the ReLU mask re-read (timing study only)")
    args = ap.parse_args()
    what = args.what.split(",")
    lib = _lib.load()
    only = set(args.only.split(",")) if args.only else None
    global TRAINER_FORM, PLAN
    TRAINER_FORM = not args.plain
    PLAN = args.plan
    if args.no_mask:
        for rnd in range(args.rounds):
            run(args.batch, args.reps, ["dgrad"], f"round {rnd}: data gradient with the ReLU mask (re-reads the layer input)", only)
            run(args.batch, args.reps, ["dgrad"], f"round {rnd}: data gradient WITHOUT the mask (timing study)", only, use_mask=False)
        return
    run(args.batch, args.reps, what, "default", only)
    if args.ab_wgrad:
        ab_wgrad(args, lib, only)
    for spec in args.ab_knob:
        knob, vals = spec.split("=")
        vals = [int(v) for v in vals.split(",")]
        for rnd in range(args.rounds):
            for v in vals:
                assert lib.dct_tune_set(int(knob), v) == 0, (knob, v)
                run(args.batch, args.reps, what, f"round {rnd}: knob {knob} = {v}", only)
        lib.dct_tune_set(int(knob), vals[0])
    if args.ab_packed:
        for rnd in range(2):
            lib.dct_tune_set(10, 0)
            run(args.batch, args.reps, [w for w in what if w != "wgrad"], f"round {rnd}: per-tap kernel on the deep levels", only)
            lib.dct_tune_set(10, 1)
            run(args.batch, args.reps, [w for w in what if w != "wgrad"], f"round {rnd}: packed-rows shared-halo kernel", only)
    if args.ab:
        for rnd in range(2):       # interleaved rounds in ONE process (devices / DVFS differ between runs)
            lib.dct_tune_set(7, 0)
            run(args.batch, args.reps, [w for w in what if w != "wgrad"], f"round {rnd}: per-tap kernel (v2) everywhere", only)
            lib.dct_tune_set(7, 1)
            run(args.batch, args.reps, [w for w in what if w != "wgrad"], f"round {rnd}: shared-halo kernel (v3) where eligible", only)

def ab_wgrad(args, lib, only):
    for rnd in range(2):
        lib.dct_tune_set(8, 0)
        run(args.batch, args.reps, ["wgrad"], f"round {rnd}: per-tap weight gradient (v2)", only)
        lib.dct_tune_set(8, 1)
        run(args.batch, args.reps, ["wgrad"], f"round {rnd}: filter-row weight gradient (v3) where eligible", only)
        lib.dct_tune_set(9, 60)
        run(args.batch, args.reps, ["wgrad"], f"round {rnd}: v3 down to 60 % K-step fill", only)
        lib.dct_tune_set(9, 70)


if __name__ == "__main__":
    main()

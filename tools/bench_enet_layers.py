#!/usr/bin/env python3
"""Per-call kernel time of one Enet training pass (forward + backward), grouped by launch configuration: every C-ABI call of
the pass is recorded once (name, arguments) and then re-issued back to back with HIP events around the batch.

    python tools/bench_enet_layers.py [--H 200] [--B 8] [--C 2] [--dtype bf16] [--reps 20] [--top 40]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import dct_amd  # noqa: E402,F401
from dct_amd import _lib, hip_ops as K  # noqa: E402
from dct_amd.arch import get_arch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=200)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--C", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[args.dtype]
    dev = "cuda:0"
    torch.manual_seed(0)
    net = get_arch("enet", {"num_classes": args.C, "compute_dtype": dt}).to(dev).train()
    x = torch.rand(args.B, 1, args.H, args.H, device=dev)
    lp, tape = net.plan_forward(x, True)          # warm: allocations, flat buffers
    net.plan_backward(tape, torch.randn_like(lp) * 1e-3, need_dx=False, need_dw=True)
    torch.cuda.synchronize()

    calls = []
    orig_call = _lib.call

    def rec(name, *a):
        calls.append((name, a))
        return orig_call(name, *a)
    K.call = rec
    lp, tape = net.plan_forward(x, True)
    nf = len(calls)
    net.plan_backward(tape, torch.randn_like(lp) * 1e-3, need_dx=False, need_dw=True)
    K.call = orig_call
    torch.cuda.synchronize()
    print(f"# enet {args.dtype} B={args.B} {args.H}x{args.H}: {nf} forward + {len(calls) - nf} backward C-ABI calls per pass")

    def key(name, a):
        vs = []
        for v in a:
            if hasattr(v, "_obj") and isinstance(v._obj, _lib.View):
                o = v._obj
                vs.append(f"{o.n}x{o.h}x{o.w}x{o.c}")
            elif hasattr(v, "_obj") and isinstance(v._obj, _lib.ConvDesc):
                o = v._obj
                vs.append(f"k{o.R}x{o.S}s{o.stride}d{o.dil}")
        return name.replace("dct_enet_", "") + " " + " ".join(vs)

    stats = collections.OrderedDict()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for idx, (name, a) in enumerate(calls):
        k = ("F " if idx < nf else "B ") + key(name, a)
        for _ in range(3):
            orig_call(name, *a)
        ev0.record()
        for _ in range(args.reps):
            orig_call(name, *a)
        ev1.record()
        torch.cuda.synchronize()
        us = 1e3 * ev0.elapsed_time(ev1) / args.reps
        s = stats.setdefault(k, [0, 0.0])
        s[0] += 1
        s[1] += us
    tot = sum(v[1] for v in stats.values())
    print(f"# sum of back-to-back per-call times: {tot / 1e3:.2f} ms per pass ({len(calls)} calls)")
    byname = collections.Counter()
    for k, (n, us) in stats.items():
        byname[k.split()[1]] += us
    print("# by entry point:", ", ".join(f"{k} {v / 1e3:.2f} ms" for k, v in byname.most_common()))
    print(f"{'calls':>5s} {'total_us':>9s} {'avg_us':>8s}  configuration")
    for k, (n, us) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{n:5d} {us:9.1f} {us / n:8.1f}  {k}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Un-pooling on load, per encoder level of the cfg2 UNet (B = 16): {un-pooling launch + data gradient + weight gradient reading the
un-pooled tensor} against {data gradient + weight gradient expanding the pooled gradient while they stage}, microseconds per launch.

    python tools/bench_unpool.py [--reps 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dct_amd  # noqa: E402,F401
from dct_amd import hip_ops as K  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_conv import timeit, plan_note  # noqa: E402

DEV = "cuda:0"
LEVELS = [("dec1b", 64, 252), ("dec2b", 128, 122), ("dec3b", 256, 57)]      # (block's second conv, channels, extent of its output)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    B, dt = args.batch, torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(0)
    print(f"{'layer':7s} {'un-pool':>8s} {'dgrad':>8s} {'wgrad':>8s} {'sum':>8s} | {'dgrad^':>8s} {'wgrad^':>8s} {'sum':>8s}   (us; ^ = expands on load)")
    for name, c, h in LEVELS:
        hp = (h + 1) // 2
        d = torch.relu(torch.randn(B, h, h, c, device=DEV, generator=g)).to(dt)
        pooled = torch.empty(B, hp, hp, c, device=DEV, dtype=dt)
        codes = torch.empty(B, hp, hp, c, device=DEV, dtype=torch.uint8)
        K.maxpool_fwd(d, pooled, codes=codes)
        dp = torch.randn(B, hp, hp, c, device=DEV, generator=g).to(dt)
        dd = torch.empty(B, h, h, c, device=DEV, dtype=dt)
        a = torch.relu(torch.randn(B, h + 2, h + 2, c, device=DEV, generator=g)).to(dt)
        abits = K.relu_bits_like(a)
        wd = (torch.randn(c, 3, 3, c, device=DEV, generator=g) / (3 * c ** 0.5)).to(dt)
        da = torch.empty_like(a)
        dw = torch.zeros(c * 9 * c, device=DEV)
        db = torch.zeros(c, device=DEV)
        up = (codes, h, h)
        t_un = timeit(lambda: K.maxpool_bwd(None, dp, dd, relu_mask=True, scale=1.0, codes=codes), args.reps)
        t_dg = timeit(lambda: K.conv2d(dd, wd, None, da, pad_h=2, pad_w=2, mask=a, mask_bits=abits), args.reps); n1 = plan_note()
        t_wg = timeit(lambda: K.conv2d_wgrad(dd, a, dw, accumulate=True, db=db), args.reps); n2 = plan_note()
        try:
            t_dgu = timeit(lambda: K.conv2d(dp, wd, None, da, pad_h=2, pad_w=2, mask=a, mask_bits=abits, unpool=up), args.reps)
            t_wgu = timeit(lambda: K.conv2d_wgrad(dp, a, dw, accumulate=True, db=db, unpool=up), args.reps)
        except K.UnpoolOnLoadUnsupported:       # (small batches: the layer does not take the shared-halo / filter-row kernel)
            print(f"{name:7s} {t_un * 1e6:8.1f} {t_dg * 1e6:8.1f} {t_wg * 1e6:8.1f} {(t_un + t_dg + t_wg) * 1e6:8.1f} | not expanded on load at this batch")
            continue
        u = 1e6
        print(f"{name:7s} {t_un * u:8.1f} {t_dg * u:8.1f} {t_wg * u:8.1f} {(t_un + t_dg + t_wg) * u:8.1f} | {t_dgu * u:8.1f} {t_wgu * u:8.1f} "
              f"{(t_dgu + t_wgu) * u:8.1f}   [{n1.split(':')[0]}; {n2.split(':')[0]}]")


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()

#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernels around the UNet conv stack at cfg2 shapes (B images of 256x256):
time per launch and effective GB/s (algorithmic bytes: every operand read or written once).

    python tools/bench_pointwise.py [--batch 16] [--reps 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dct_amd  # noqa: E402,F401
from dct_amd import hip_ops as K  # noqa: E402

DEV = "cuda:0"


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    B, dt = args.batch, torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(0)
    rows = []

    img = torch.randn(B, 256, 256, 1, device=DEV, generator=g)
    w = torch.randn(64, 3, 3, 1, device=DEV, generator=g)
    b = torch.randn(64, device=DEV, generator=g)
    y = torch.empty(B, 254, 254, 64, device=DEV, dtype=dt)
    dy = torch.randn(B, 254, 254, 64, device=DEV, generator=g).to(dt)
    dw, db = torch.zeros(64 * 9, device=DEV), torch.zeros(64, device=DEV)
    dimg = torch.empty_like(img)
    rows.append(("stem_fwd", timeit(lambda: K.conv_cin1_fwd(img, w, b, y, relu=True), args.reps), y.numel() * 2 + img.numel() * 4))
    rows.append(("stem_wgrad", timeit(lambda: K.conv_cin1_wgrad(img, dy, dw, db), args.reps), dy.numel() * 2 + img.numel() * 4))
    rows.append(("stem_dgrad", timeit(lambda: K.conv_cin1_dgrad(dy, w, dimg), args.reps), dy.numel() * 2 + img.numel() * 4))
    for c, h in ((64, 252), (128, 122), (256, 57), (512, 25)):
        x = torch.randn(B, h, h, c, device=DEV, generator=g).to(dt)
        hp = (h + 1) // 2
        yp = torch.empty(B, hp, hp, c, device=DEV, dtype=dt)
        dyp = torch.randn(B, hp, hp, c, device=DEV, generator=g).to(dt)
        dx = torch.empty_like(x)
        rows.append((f"maxpool_fwd {c}x{h}", timeit(lambda: K.maxpool_fwd(x, yp), args.reps), (x.numel() + yp.numel()) * 2))
        rows.append((f"maxpool_bwd {c}x{h}", timeit(lambda: K.maxpool_bwd(x, dyp, dx, relu_mask=True), args.reps),
                     (2 * x.numel() + yp.numel()) * 2))
    # the step's own un-pooling: routed by the pooling codes, the skip connection's bilinear backward gathered on the way (cat res per level)
    for (c, h), cres in zip(((64, 252), (128, 122), (256, 57), (512, 25)), (88, 48, 28, 18)):
        x = torch.randn(B, h, h, c, device=DEV, generator=g).to(dt)
        hp = (h + 1) // 2
        yp = torch.empty(B, hp, hp, c, device=DEV, dtype=dt)
        codes = torch.empty(B, hp, hp, c, device=DEV, dtype=torch.uint8)
        K.maxpool_fwd(x, yp, codes=codes)
        dyp = torch.randn(B, hp, hp, c, device=DEV, generator=g).to(dt)
        dcat = torch.randn(B, cres, cres, 2 * c, device=DEV, generator=g).to(dt)
        dx = torch.empty_like(x)
        rows.append((f"unpool+skip {c}x{h}", timeit(lambda: K.maxpool_bwd(None, dyp, dx, relu_mask=True, codes=codes, skip=dcat[..., c:]), args.reps),
                     (x.numel() + yp.numel() + dcat.numel() // 2) * 2 + codes.numel()))
        cat = torch.empty(B, cres, cres, 2 * c, device=DEV, dtype=dt)
        rows.append((f"skip resize {c}x{hp}->{cres}", timeit(lambda: K.bilinear_fwd(yp, cat[..., c:]), args.reps), (yp.numel() + cat.numel() // 2) * 2))
    for name, t, nbytes in rows:
        print(f"{name:20s} {t * 1e6:8.1f} us  {nbytes / t / 1e9:8.1f} GB/s")


if __name__ == "__main__":
    main()

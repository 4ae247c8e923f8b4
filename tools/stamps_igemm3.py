#!/usr/bin/env python3
"""Per-phase cycle breakdown of igemm3_kernel (diagnostic build: `make -C <pkg>/csrc EXTRA=-DDCT_STAMPS`).
Per wave and K-step: DMA issue / fragment reads + MFMA issue / wait for the prefetched stage / barrier, plus
prologue and epilogue per block, from s_memtime stamps accumulated over every wave of the launch.

    python tools/stamps_igemm3.py [--layer dec2b]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dct_amd  # noqa: E402,F401
from dct_amd import _lib, hip_ops as K  # noqa: E402

LAYERS = {"dec1b": (64, 254, 64), "dec2a": (64, 126, 128), "dec2b": (128, 124, 128), "dec3b": (256, 59, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="dec2b")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--zeros", type=float, default=0.0, help="fraction of exactly-zero input activations (a ReLU network's operands)")
    ap.add_argument("--spin", type=float, default=2.0, help="seconds of back-to-back launches before the stamped ones (the clock settles)")
    args = ap.parse_args()
    cin, hin, cout = LAYERS[args.layer]
    lib = _lib.load()
    lib.dct_tune_set(11, 0)          # DCT_TUNE_IGEMM_MFMA16 = 0: the stamps sit in the 32x32x16 form (igemm3_kernel)
    fn = lib.dct_debug_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int]
    dev = "cuda:0"
    B, ho = args.batch, hin - 2
    x = torch.randn(B, hin, hin, cin, device=dev)
    if args.zeros > 0:
        x = x.abs() * (torch.rand_like(x) >= args.zeros)          # non-negative with the given share of exact zeros, like a ReLU output
    x = x.to(torch.bfloat16)
    w = (torch.randn(cout, 3, 3, cin, device=dev) / (3 * cin ** 0.5)).to(torch.bfloat16)
    bias = torch.randn(cout, device=dev)
    y = torch.empty(B, ho, ho, cout, device=dev, dtype=torch.bfloat16)
    import time
    t_end = time.time() + args.spin
    while time.time() < t_end:
        for _ in range(50):
            K.conv2d(x, w, bias, y, relu=True)
        torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    fn(buf, 1)
    n = 10
    for _ in range(n):
        K.conv2d(x, w, bias, y, relu=True)
    fn(buf, 1)
    waves, steps = buf[0], buf[8]
    if not waves:
        raise SystemExit("no stamps: not a DCT_STAMPS build, or the layer did not take the shared-halo kernel")
    spw = steps / waves
    print(f"{args.layer}: {waves // n} waves/launch, {spw:.0f} K-steps per wave")
    if buf[13]:
        print(f"  in-kernel shader clock: {buf[7] / buf[13] * 0.1:.3f} GHz  (s_memtime / s_memrealtime x 100 MHz over whole wave lifetimes; "
              f"input zeros {args.zeros:.0%})")
    print(f"  per wave: prologue {buf[1] / waves:8.0f}  loop {(buf[2] + buf[3] + buf[4] + buf[5]) / waves:8.0f}  "
          f"epilogue {buf[6] / waves:8.0f}  total {buf[7] / waves:8.0f} cycles")
    print(f"  epilogue: row table {buf[9] / waves:6.0f}  acc -> LDS {buf[10] / waves:6.0f}  barrier {buf[11] / waves:6.0f}  "
          f"LDS -> global {buf[12] / waves:6.0f}")
    print(f"  per K-step: DMA issue {buf[2] / steps:6.0f}  reads+MFMA issue {buf[3] / steps:6.0f}  "
          f"vmcnt wait {buf[4] / steps:6.0f}  barrier {buf[5] / steps:6.0f}  sum {(buf[2] + buf[3] + buf[4] + buf[5]) / steps:6.0f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One conv layer of the cfg2 UNet, launched N times -- the workload of a `rocprofv3 --pmc` pass on ONE kernel
(tools/gpu/pmc_layer.sh collects the SQ counters of the loop and tools/pmc_layer_report.py prints them per wave-cycle).

    python tools/pmc_layer.py --layer dec2b --what fwd --n 20"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dct_amd  # noqa: E402,F401
from dct_amd import hip_ops as K  # noqa: E402

LAYERS = {"dec1b": (64, 254, 64), "dec2a": (64, 126, 128), "dec2b": (128, 124, 128), "dec3b": (256, 59, 256), "dec4b": (512, 27, 512),
          "cen_b": (1024, 11, 1024), "enc3a": (512, 28, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="dec2b")
    ap.add_argument("--what", default="fwd", choices=["fwd", "dgrad", "wgrad"])
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE", help="dct_tune_set before the launches")
    a = ap.parse_args()
    from dct_amd import _lib
    for kv in a.tune:
        k, v = kv.split("=")
        assert _lib.load().dct_tune_set(int(k), int(v)) == 0, kv
    cin, hin, cout = LAYERS[a.layer]
    dev, dt, B, ho = "cuda:0", torch.bfloat16, a.batch, hin - 2
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, hin, hin, cin, device=dev, generator=g).to(dt)
    w = (torch.randn(cout, 3, 3, cin, device=dev, generator=g) / (3 * cin ** 0.5)).to(dt)
    wd = (torch.randn(cin, 3, 3, cout, device=dev, generator=g) / (3 * cout ** 0.5)).to(dt)
    bias = torch.randn(cout, device=dev, generator=g)
    y = torch.empty(B, ho, ho, cout, device=dev, dtype=dt)
    dy = torch.randn(B, ho, ho, cout, device=dev, generator=g).to(dt)
    dx = torch.empty(B, hin, hin, cin, device=dev, dtype=dt)
    dw = torch.zeros(cout * 9 * cin, device=dev)
    for _ in range(a.n):
        if a.what == "fwd":
            K.conv2d(x, w, bias, y, relu=True)
        elif a.what == "dgrad":
            K.conv2d(dy, wd, None, dx, pad_h=2, pad_w=2, mask=x)
        else:
            K.conv2d_wgrad(dy, x, dw, accumulate=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""igemm5_kernel (csrc/igemm5.hip) against the igemm.hip tiles: bit identity on the forward and data-gradient forms (bias + ReLU + gate
bits; mask + scale + accumulate; gate-bit mask), a repeat-launch race screen, and per-layer timing with the knob off / on.
With the diagnostic build (DCT_LIB_PATH=...libdct_hip_abl.so) a list of ablation variants may follow: i5_check.py time 0,1,4,8,64"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dct_amd  # noqa
from dct_amd import _lib, hip_ops as K

DEV = "cuda:0"
CASES = [(4, 64, 130, 130, 128, 0), (5, 128, 122, 90, 128, 2), (3, 192, 96, 96, 256, 0), (16, 256, 59, 59, 256, 0), (2, 64, 40, 300, 64, 0),
         (16, 64, 86, 86, 64, 2), (6, 128, 61, 61, 256, 0)]
LAYERS = {"dec1b": (64, 254, 64), "dec2a": (64, 126, 128), "dec2b": (128, 124, 128), "dec3a": (128, 61, 256), "dec3b": (256, 59, 256),
          "enc2b": (128, 46, 128), "enc1a": (128, 88, 64), "enc1b": (64, 86, 64)}


def timeit(fn, reps=30):
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


def check(lib, launches):
    bad = 0
    for (B, Cin, H, W, Cout, pad) in CASES:
        g = torch.Generator().manual_seed(14)
        x = torch.randn(B, H, W, Cin, generator=g).bfloat16().to(DEV)
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) / math.sqrt(Cin * 9)).bfloat16().to(DEV)
        b = torch.randn(Cout, generator=g).to(DEV)
        Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
        mask = torch.randn(B, Ho, Wo, Cout, generator=g).bfloat16().to(DEV)
        pos = (mask.float() > 0).to(torch.int32)
        mbits = (pos.view(B, Ho, Wo, Cout // 8, 8) << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8).contiguous()

        def run():
            y = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=DEV)
            bits = torch.zeros(B, Ho, Wo, Cout // 8, dtype=torch.uint8, device=DEV)
            K.conv2d(x, w, b, y, pad_h=pad, pad_w=pad, relu=True, relu_bits_out=bits)
            z = torch.ones(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=DEV)
            K.conv2d(x, w, None, z, pad_h=pad, pad_w=pad, mask=mask, mask_scale=2.0, accumulate=True)
            u = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=DEV)
            K.conv2d(x, w, None, u, pad_h=pad, pad_w=pad, mask=mask, mask_bits=mbits)
            return y, bits, z, u
        lib.dct_tune_set(38, 0); lib.dct_tune_set(35, 0); lib.dct_tune_set(10, 0)
        ref = run()
        lib.dct_tune_set(38, 1); lib.dct_tune_set(39, 1)
        got = run()
        same = [bool(torch.equal(a, r)) for a, r in zip(got, ref)]
        close = [(a.float() - r.float()).abs().max().item() for a, r in zip(got, ref)]
        races = 0
        for _ in range(launches):
            again = run()
            races += sum(0 if torch.equal(a, r) else 1 for a, r in zip(again, got))
        print(f"case {(B, Cin, H, W, Cout, pad)}: identical {same} maxdiff {close} races {races}", flush=True)
        bad += (not all(same)) + races
    lib.dct_tune_set(38, 0); lib.dct_tune_set(39, 200)
    return bad


def time_layers(lib, variants):
    B = 16
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, (cin, hin, cout) in LAYERS.items():
        ho = hin - 2
        x = torch.randn(B, hin, hin, cin, device=DEV, generator=g).bfloat16()
        w = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).bfloat16()
        bias = torch.randn(cout, device=DEV, generator=g)
        y = torch.empty(B, ho, ho, cout, device=DEV, dtype=torch.bfloat16)
        fl = 2.0 * B * ho * ho * 9 * cin * cout
        lib.dct_tune_set(38, 0)
        t_old = timeit(lambda: K.conv2d(x, w, bias, y, relu=True))
        lib.dct_tune_set(38, 1); lib.dct_tune_set(39, 1)
        line = f"{name}: igemm.hip {t_old:6.1f} us {fl / t_old / 1e6:6.1f} TF |"
        for v in variants:
            lib.dct_tune_set(1000, v)
            t = timeit(lambda: K.conv2d(x, w, bias, y, relu=True))
            line += f" i5[{v}] {t:6.1f} us {fl / t / 1e6:6.1f} TF |"
        lib.dct_tune_set(1000, 0)
        print(line, flush=True)
    lib.dct_tune_set(38, 0); lib.dct_tune_set(39, 200)


if __name__ == "__main__":
    lib = _lib.load()
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    bad = 0
    if what in ("all", "check"):
        bad = check(lib, 20)
    if what in ("all", "time"):
        time_layers(lib, [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0])
    sys.exit(1 if bad else 0)

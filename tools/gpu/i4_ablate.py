#!/usr/bin/env python3
"""Ablation timing of igemm4_kernel (diagnostic build: make EXTRA=-DDCT_I4_ABLATE BUILD=build_abl OUT=../libdct_hip_abl.so, run with
DCT_LIB_PATH=<that .so>): which part of a phase bounds the loop.  Results of ablated variants are wrong by construction."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dct_amd  # noqa
from dct_amd import _lib, hip_ops as K

DEV = "cuda:0"
LAYERS = {"dec2b": (128, 124, 128), "dec3b": (256, 59, 256), "dec2a": (64, 126, 128), "dec4b": (512, 27, 512)}
NAMES = {100: "full, reads in M", 0: "full", 1: "-Wdma", 2: "-Hdma", 3: "-W-H dma", 4: "-reads", 8: "-mfma", 12: "-reads-mfma", 7: "-dma-reads", 16: "-barriers",
         32: "-vmcnt", 35: "-dma-vmcnt", 39: "-dma-vmcnt-reads", 47: "only barriers", 63: "nothing"}


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


def main():
    lib = _lib.load()
    B = 16
    variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else sorted(NAMES)
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, (cin, hin, cout) in LAYERS.items():
        ho = hin - 2
        x = torch.randn(B, hin, hin, cin, device=DEV, generator=g).bfloat16()
        w = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).bfloat16()
        bias = torch.randn(cout, device=DEV, generator=g)
        y = torch.empty(B, ho, ho, cout, device=DEV, dtype=torch.bfloat16)
        fl = 2.0 * B * ho * ho * 9 * cin * cout
        lib.dct_tune_set(35, 0)
        t_old = timeit(lambda: K.conv2d(x, w, bias, y, relu=True))
        lib.dct_tune_set(35, 1)
        print(f"{name}: igemm.hip kernel {t_old:7.1f} us {fl / t_old / 1e6:7.1f} TF")
        for rnd in range(2):
            for v in variants:
                lib.dct_tune_set(1000, v)
                t = timeit(lambda: K.conv2d(x, w, bias, y, relu=True))
                print(f"  round {rnd} abl {v:2d} {NAMES.get(v, '?'):18s} {t:7.1f} us {fl / t / 1e6:7.1f} TF")
        lib.dct_tune_set(1000, 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where a wave of wgrad3_kernel (lean form) spends a K-step: diagnostic build with -DDCT_W3_STAMPS
(`make BUILD=build_stamps OUT=../libdct_hip_stamps.so EXTRA=-DDCT_W3_STAMPS`), selected by DCT_LIB_PATH:

    DCT_LIB_PATH=$PWD/deep-co-training-for-semi-supervised-image-segmentation_amd/libdct_hip_stamps.so python tools/gpu/w3_stamps.py

Segments per K-step (s_memtime at the step's top, behind the LDS-DMA issue of the next stage, behind the last MFMA's issue, behind the
barrier): read the SHARES, not the lengths -- the stamps' fences forbid overlaps the shipped kernel has."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dct_amd  # noqa: E402,F401
from dct_amd import _lib, hip_ops as K  # noqa: E402

DEV = "cuda:0"
LAYERS = {"dec1b": (64, 254, 64), "dec2b": (128, 124, 128), "dec3b": (256, 59, 256), "dec4b": (512, 27, 512), "enc2a": (256, 48, 128)}


def main():
    lib = _lib.load()
    lib.dct_debug_w3_stamps.argtypes = [C.c_void_p]
    g = torch.Generator(device=DEV).manual_seed(0)
    B = 16
    buf = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=DEV)
    for name, (cin, hin, cout) in LAYERS.items():
        ho = hin - 2
        x = torch.randn(B, hin, hin, cin, device=DEV, generator=g).bfloat16()
        dy = torch.randn(B, ho, ho, cout, device=DEV, generator=g).bfloat16()
        dw = torch.zeros(cout * 9 * cin, device=DEV)
        db = torch.zeros(cout, device=DEV)
        for _ in range(3):
            K.conv2d_wgrad(dy, x, dw, accumulate=False, db=db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.conv2d_wgrad(dy, x, dw, accumulate=False, db=db)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        buf.zero_()
        lib.dct_debug_w3_stamps(C.c_void_p(buf.data_ptr()))
        K.conv2d_wgrad(dy, x, dw, accumulate=False, db=db)
        torch.cuda.synchronize()
        lib.dct_debug_w3_stamps(None)
        r = buf.view(4096, 8, 8).cpu().double()
        live = r[:, :, 3] > 0
        nb = int(live[:, 0].sum().item())
        m = r[live].mean(0)
        steps = m[4].item()
        tot = m[3].item()
        print(f"{name}: {t * 1e6:6.1f} us with stamps; {nb} blocks x 8 waves, {steps:.1f} K-steps per wave; per K-step {tot / steps:7.0f} cycles = "
              f"next stage's LDS-DMA issue {m[0].item() / steps:6.0f} ({100 * m[0].item() / tot:4.1f} %) | reads + 12 MFMAs issued {m[1].item() / steps:6.0f} "
              f"({100 * m[1].item() / tot:4.1f} %) | vmcnt(0) + barrier {m[2].item() / steps:6.0f} ({100 * m[2].item() / tot:4.1f} %)")


if __name__ == "__main__":
    main()

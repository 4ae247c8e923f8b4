#!/usr/bin/env python3
"""Where a block of igemm4_kernel spends its cycles (diagnostic build with -DDCT_I4_ABLATE; DCT_LIB_PATH selects it)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dct_amd  # noqa
from dct_amd import _lib, hip_ops as K
from tools.gpu.i4_ablate import LAYERS, timeit

DEV = "cuda:0"
lib = _lib.load()
lib.dct_debug_i4_stamps.argtypes = [C.c_void_p]
g = torch.Generator(device=DEV).manual_seed(0)
B = 16
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=DEV)
for name, (cin, hin, cout) in LAYERS.items():
    ho = hin - 2
    x = torch.randn(B, hin, hin, cin, device=DEV, generator=g).bfloat16()
    w = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).bfloat16()
    bias = torch.randn(cout, device=DEV, generator=g)
    y = torch.empty(B, ho, ho, cout, device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * B * ho * ho * 9 * cin * cout
    lib.dct_debug_i4_stamps(None)
    t = timeit(lambda: K.conv2d(x, w, bias, y, relu=True))
    lib.dct_debug_i4_stamps(C.c_void_p(buf.data_ptr()))
    buf.zero_()
    K.conv2d(x, w, bias, y, relu=True)
    torch.cuda.synchronize()
    lib.dct_debug_i4_stamps(None)
    r = buf.view(256, 8, 8).cpu().double()
    tiles = r[:, 0, 4].mean().item()
    print(f"{name}: {t:6.1f} us {fl / t / 1e6:6.1f} TF; tiles/block {tiles:.2f}; cycles per block (mean over blocks), wave 0 / wave 4:")
    for wv in (0, 4):
        m = r[:, wv, :].mean(0)
        tot = m[5].item()
        print(f"   wave {wv}: total {tot:9.0f} ({tot / (t * 1e-6) / 1e9:4.2f} GHz if it spans the launch) | loop {m[0].item():9.0f} ({100 * m[0].item() / tot:4.1f} %) "
              f"| epilogue to staged tile {m[1].item():8.0f} ({100 * m[1].item() / tot:4.1f} %) | row stores {m[2].item():8.0f} ({100 * m[2].item() / tot:4.1f} %) "
              f"| next-tile set-up {m[3].item():8.0f} ({100 * m[3].item() / tot:4.1f} %)")

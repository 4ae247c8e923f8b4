#!/usr/bin/env python3
"""Where the in-step meters' time goes (diagnostic): DiceMeter.add / value on bench-shaped predictions, host and device time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dct_amd.metrics import DiceMeter
dev = "cuda:0"
B, C, H = 8, 4, 256
pred = torch.randn(B, H, H, C, device=dev).permute(0, 3, 1, 2)
gt = torch.randint(0, C, (B, 1, H, H), device=dev)
m = DiceMeter(method="2d", report_axises=[1, 2, 3], C=C)
for _ in range(5):
    m.add(pred, gt)
torch.cuda.synchronize()
for n, what in ((200, "add"), (50, "add+value")):
    t0 = time.perf_counter()
    for _ in range(n):
        m.add(pred, gt)
        if what != "add":
            float(m.value()[0][0])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    td = time.perf_counter() - t0
    print(f"{what}: host {1e6 * th / n:.1f} us/call, host+device {1e6 * td / n:.1f} us/call")

x = torch.zeros(4, device=dev)
torch.cuda.synchronize()
for what, fn in (("synchronize()", lambda: torch.cuda.synchronize()), (".item()", lambda: x[0].item()), (".cpu()", lambda: x.cpu()),
                 ("double ops + item", lambda: ((x.double() / 3).sqrt().float())[0].item())):
    t0 = time.perf_counter()
    for _ in range(50):
        x.add_(1.0)
        fn()
    print(f"{what}: {1e6 * (time.perf_counter() - t0) / 50:.1f} us per (tiny kernel + call)")

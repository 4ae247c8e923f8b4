"""Gradients of one Enet backward pass with the BatchNorm-backward sums written by the data-gradient convolutions' epilogues against
the separate reduction (identical forward).   python tools/debug_fuse_bn.py"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from dct_amd.arch import get_arch
DEV = "cuda:0"
def run(fuse, dt, scale, B=4, H=256):
    torch.manual_seed(5)
    net = get_arch("enet", {"num_classes": 4, "compute_dtype": dt}).to(DEV).train()
    net.fuse_bn_bwd_stats = fuse
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 1, H, H, generator=g).to(DEV)
    lp, tape = net.plan_forward(x, True)
    dl = (torch.randn(lp.shape, generator=g) * scale / (B * H * H)).to(DEV)
    net.flat_params.ensure_grads()
    net.flat_params.gflat.zero_()
    net.plan_backward(tape, dl, need_dx=False, need_dw=True)
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone().cpu() for k, p in net.named_parameters()}
for dt, scale in ((torch.bfloat16, 1.0), (torch.float16, 2.0 ** 18)):
    a, b = run(True, dt, scale), run(False, dt, scale)
    worst = []
    for k in a:
        d = (a[k].double() - b[k].double()).norm() / (b[k].double().norm() + 1e-30)
        worst.append((float(d), k, float(b[k].abs().max())))
    worst.sort(reverse=True)
    print(dt, "worst relative gradient differences, BatchNorm-backward sums fused into the data-gradient epilogue vs separate reduction:")
    for w in worst[:8]:
        print("   %.3e  %-50s max|g| %.3e" % w)

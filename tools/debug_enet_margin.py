#!/usr/bin/env python3
"""Margins of tests/test_enet_gpu.py::test_enet_vs_oracle_fwd_bwd_train in fp32 under the reduction knobs (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from dct_amd import _lib
import test_enet_gpu as T

lib = _lib.load()
for (B, H, W, C) in ((2, 64, 64, 2), (1, 96, 128, 4)):
    for ft, vec, skip in ((256, 0, False), (1024, 0, False), (1024, 1, False), (1024, 1, True)):
        lib.dct_tune_set(22, ft); lib.dct_tune_set(21, vec)
        onet = T._oracle_net(C, 7).train()
        net = T._hip_net(onet, C, torch.float32).train()
        net.skip_zero_bias_grads = skip
        g = torch.Generator().manual_seed(3)
        x = torch.rand(B, 1, H, W, generator=g); t = torch.randint(0, C, (B, H, W), generator=g)
        xo = x.clone().requires_grad_(True); yo = onet(xo); oracle.cross_entropy_2d(yo, t).backward()
        xd = x.to("cuda:0").requires_grad_(True); y = net(xd)
        yo2 = yo.detach().clone().requires_grad_(True)
        gl = torch.autograd.grad(oracle.cross_entropy_2d(yo2, t), yo2)[0]
        y.backward(gl.to("cuda:0"))
        errs = {k: T._rel2(p.grad.cpu().numpy(), po.grad.numpy()) for (k, p), (_, po) in zip(net.named_parameters(), onet.named_parameters()) if po.grad.norm() >= 1e-6}
        errs["grad_x"] = T._rel2(xd.grad.cpu().numpy(), xo.grad.numpy())
        v = np.array(list(errs.values()))
        print((B, H, W, C), "fold", ft, "vec", vec, "skip", skip, "logits", T._rel2(y.detach().cpu().numpy(), yo.detach().numpy()),
              "grad err max %.3e median %.3e n>5e-3: %d" % (v.max(), np.median(v), (v > 5e-3).sum()), flush=True)

#!/usr/bin/env python3
"""Per-block comparison of the Enet HIP plan with the CPU oracle (debug aid): L2-relative error of every
block output, with the oracle's block outputs rounded to bf16 when the plan runs in bf16."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dct_amd  # noqa: E402,F401
import oracle  # noqa: E402
from dct_amd.arch import get_arch  # noqa: E402

dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
C, B, H, W = 2, 2, 64, 64
torch.manual_seed(7)
onet = oracle.build_net("enet", C).train()
net = get_arch("enet", {"num_classes": C, "compute_dtype": dtype})
net.load_state_dict(onet.state_dict())
net = net.to("cuda:0").train()
outs = []


def hook(m, i, o):
    t = o[0] if isinstance(o, tuple) else o
    if dtype == torch.bfloat16:
        t = t.bfloat16().float()
    outs.append(t.detach())
    return (t,) + tuple(o[1:]) if isinstance(o, tuple) else t


for m in onet.modules():
    if m.__class__.__name__ in ("_Bottleneck", "_Initial"):
        m.register_forward_hook(hook)
x = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(3))
yo = onet(x).detach()
xd = x.to("cuda:0").requires_grad_(True)
net.flat_params.ensure()
logits, tape = net._run_forward(xd, True)
for k, (st, o) in enumerate(zip(tape[:-1], outs)):
    h = st["out"].float().cpu().permute(0, 3, 1, 2)
    err = ((h - o).norm() / o.norm()).item()
    kind = st.get("kind") or st["blk"].kind
    print(f"block {k:2d} {kind:8s} shape {tuple(o.shape)} rel-L2 {err:.3e}  max|d| {(h - o).abs().max().item():.3e}")
y = logits.permute(0, 3, 1, 2).float().cpu()
print("logits rel-L2", ((y - yo).norm() / yo.norm()).item())

"""N steps of a bench configuration, nothing else (for rocprofv3 timelines).   python tools/run_steps.py cfg4 12 [attr=0|1 ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

name, n = sys.argv[1], int(sys.argv[2])
cfg = bench.CONFIGS[name]
dev = torch.device("cuda:0")
tr, lab, unl = bench.make_trainer(cfg, torch.bfloat16, dev, 0, 1, None)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    setattr(tr, k, bool(int(v)))
S, nb = cfg["S"], len(unl)
for i in range(n):
    lb = [(lab[m][i % nb][0][0], lab[m][i % nb][0][1]) for m in range(S)]
    ub = (unl[i % nb][0][0], unl[i % nb][0][1])
    tr._run_step(lb, ub, True, cfg["train_adv"], (0, 1) if cfg["train_adv"] else None)
    torch.cuda.synchronize()
print("done")

"""__graft_entry__.smoke(): one small co-training step on cuda:0, checked against the CPU oracle."""
import os
import sys
import tempfile

import numpy as np
import torch


def run_smoke():
    import oracle
    from dct_amd.loss import get_loss_fn
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import FakeLoader, batches
    C, H, B = 4, 176, 1
    segs, omodels = [], []
    for seed in (1, 2):
        torch.manual_seed(seed)
        onet = oracle.build_net("unet", C, dropout_p=0.0).train()
        seg = Segmentator({"name": "unet", "num_classes": C, "compute_dtype": torch.float32, "dropout_p": 0.0},
                          {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4}, {"name": "StepLR", "step_size": 90, "gamma": 0.1})
        seg.torchnet.load_state_dict(onet.state_dict())
        segs.append(seg)
        omodels.append(oracle.OracleModel.make(onet))
    lab = [FakeLoader(batches(31 + i, 1, B, H, C), B) for i in range(2)]
    unl = FakeLoader(batches(41, 1, B, H, C), B)
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=tempfile.mkdtemp(), device="cuda:0", axises=[1, 2, 3],
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=1)
    for s in segs:
        s.train()
    lb = [(lab[i][0][0][0], lab[i][0][0][1]) for i in range(2)]
    ub = (unl[0][0][0], unl[0][0][1])
    out = tr._run_step(lb, ub, True, True, (0, 1))
    ref = oracle.cotrain_step(omodels, lb, ub[0], True, True, lam_cot=0.5, lam_adv=0.05, eps=0.03)
    np.testing.assert_allclose([s.item() for s in out["sup"]], [s.item() for s in ref["sup"]], rtol=1e-5)
    np.testing.assert_allclose(out["jsd"].item(), ref["jsd"].item(), rtol=1e-4)
    np.testing.assert_allclose(out["adv"].item(), ref["adv"].item(), rtol=2e-2)
    for seg, om in zip(segs, omodels):
        a = torch.cat([p.detach().flatten().cpu() for p in seg.torchnet.parameters()]).double()
        b = torch.cat([p.detach().flatten() for p in om.net.parameters()]).double()
        rel = ((a - b).norm() / b.norm()).item()
        assert rel < 1e-3, rel
    print("smoke ok: sup", [round(s.item(), 5) for s in out["sup"]], "jsd", out["jsd"].item(), "adv", out["adv"].item())

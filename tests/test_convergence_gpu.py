"""Training behaviour of the benchmarked bf16 mode against the fp32 oracle ("next" 3).

A randomly initialised train-mode Enet amplifies rounding (tests/test_enet_gpu.py: rounding the input image to bf16 alone moves
the oracle's own logits by 12 %), so tensor-by-tensor parity of a bf16 Enet says little.  What must hold is that the bf16 HIP
step TRAINS to the same place as the reference arithmetic: from identical initial weights, on identical blob-structured batches
(tests/helpers.py::blob_batches -- learnable, unlike i.i.d. random labels), the supervised loss must fall, and the loss level
and the foreground Dice of the last 20 steps must sit within a stated band of the oracle's fp32 run
(cotraining_totalloss.py:203-248 for the step, metrics/dice_meter.py:12-83 for the Dice).

    python tests/test_convergence_gpu.py        prints both loss curves (the committed evidence lives in profiles/)
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle  # noqa: E402
from helpers import FakeLoader, blob_batches  # noqa: E402

DEV = "cuda:0"


def run_curves(arch, H, B, C, steps, n_batches, dtype, adv=False, tmp="/tmp/dct_conv"):
    """-> dict(hip=..., ref=...) of per-step supervised losses (mean over the models) and foreground 2-D Dice."""
    from dct_amd.loss import get_loss_fn
    from dct_amd.metrics import DiceMeter
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    segs, oms = [], []
    for seed in (101, 102):
        torch.manual_seed(seed)
        onet = oracle.build_net(arch, C, **({"dropout_p": 0.0} if arch == "unet" else {})).train()
        arch_dict = {"name": arch, "num_classes": C, "compute_dtype": dtype}
        if arch == "unet":
            arch_dict["dropout_p"] = 0.0
        seg = Segmentator(arch_dict, {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4}, {"name": "StepLR", "step_size": 90, "gamma": 0.1})
        seg.torchnet.load_state_dict(onet.state_dict())
        segs.append(seg)
        oms.append(oracle.OracleModel.make(onet))
    lab = [FakeLoader(blob_batches(201 + i, n_batches, B, H, C), B) for i in range(2)]
    unl = FakeLoader(blob_batches(301, n_batches, B, H, C), B)
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    os.makedirs(tmp, exist_ok=True)
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=tmp, device=DEV, axises=list(range(1, C)),
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=steps)
    for s in segs:
        s.train()
    hip = dict(sup=[], dice=[])
    ref = dict(sup=[], dice=[])
    for k in range(steps):
        lb = [(lab[i][k % n_batches][0][0], lab[i][k % n_batches][0][1]) for i in range(2)]
        ub = (unl[k % n_batches][0][0], unl[k % n_batches][0][1])
        out = tr._run_step(lb, ub, True, adv, (0, 1) if adv else None)
        meter = DiceMeter(report_axises=list(range(1, C)), method='2d', C=C)
        for i in range(2):
            meter.add(out["preds"][i], lb[i][1].to(DEV))
        hip["sup"].append(float(sum(out["sup"]) / 2))
        hip["dice"].append(float(meter.value()[0][0]))
        r = oracle.cotrain_step(oms, lb, ub[0], True, adv, lam_cot=0.5, lam_adv=0.05, eps=0.03)
        ref["sup"].append(float(sum(r["sup"]) / 2))
        d = torch.cat([oracle.dice_2d(r["preds"][i], lb[i][1]) for i in range(2)])
        ref["dice"].append(float(d[:, 1:].mean()))
    return dict(hip=hip, ref=ref)


def _check(c, tail=20, loss_band=0.25, dice_band=0.10):
    hs, rs = np.array(c["hip"]["sup"]), np.array(c["ref"]["sup"])
    hd, rd = np.array(c["hip"]["dice"]), np.array(c["ref"]["dice"])
    assert np.isfinite(hs).all() and np.isfinite(hd).all()
    # the loss falls: the last `tail` steps sit well below the first 10, for both arithmetic modes
    assert hs[-tail:].mean() < 0.7 * hs[:10].mean(), (hs[:10].mean(), hs[-tail:].mean())
    assert rs[-tail:].mean() < 0.7 * rs[:10].mean(), (rs[:10].mean(), rs[-tail:].mean())
    # ... to the same level, with the same segmentation quality (band: |difference| of the tail means)
    assert abs(hs[-tail:].mean() - rs[-tail:].mean()) <= loss_band * rs[-tail:].mean(), (hs[-tail:].mean(), rs[-tail:].mean())
    # (Dice: not worse than the reference arithmetic by more than the band.  It may be better -- the bf16 Enet's 200-step Dice has
    # read 0.69-0.72 against the fp32 oracle's 0.60: rounding noise acts on this short run like a little regularisation -- and an
    # upper bound of twice the band still catches a meter that saturates for the wrong reason)
    assert -dice_band <= hd[-tail:].mean() - rd[-tail:].mean() <= 2 * dice_band, (hd[-tail:].mean(), rd[-tail:].mean())
    # and the Dice rises wherever the reference arithmetic's does (60 UNet steps at bs 1 + 1 only reach the background prior)
    if rd[-tail:].mean() > rd[:10].mean() + 0.1:
        assert hd[-tail:].mean() > hd[:10].mean() + 0.05, (hd[:10].mean(), hd[-tail:].mean())


def test_enet_bf16_trains_like_the_fp32_oracle():
    """2 x Enet, 96 x 96, C = 3, bs 4 + 4, CE + JSD, 200 steps over 10 distinct batches."""
    _check(run_curves("enet", 96, 4, 3, 200, 10, torch.bfloat16))


def test_unet_bf16_trains_like_the_fp32_oracle():
    """2 x UNet, 176 x 176, C = 3, bs 1 + 1, CE + JSD, 60 steps over 6 distinct batches."""
    _check(run_curves("unet", 176, 1, 3, 60, 6, torch.bfloat16), tail=15)


if __name__ == "__main__":
    from conftest import host_cores
    torch.set_num_threads(host_cores())
    res = {}
    for name, args in (("enet_bf16", ("enet", 96, 4, 3, 200, 10, torch.bfloat16)),
                       ("enet_f32", ("enet", 96, 4, 3, 200, 10, torch.float32)),
                       ("unet_bf16", ("unet", 176, 1, 3, 60, 6, torch.bfloat16))):
        c = run_curves(*args)
        res[name] = c
        for side in ("hip", "ref"):
            s, d = np.array(c[side]["sup"]), np.array(c[side]["dice"])
            print(name, side, "sup first10 %.4f last20 %.4f | dice first10 %.4f last20 %.4f" %
                  (s[:10].mean(), s[-20:].mean(), d[:10].mean(), d[-20:].mean()), flush=True)
    out = os.environ.get("DCT_CONV_OUT")
    if out:
        with open(out, "w") as f:
            json.dump(res, f)

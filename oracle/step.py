"""One co-training step on CPU fp32, restating the loop body
/root/reference/generalframework/trainer/cotraining_totalloss.py:203-248 and
``_FSGM_adv_training`` (:371-392,440-442).  TEST INFRASTRUCTURE ONLY (also the timed
``cpu_baseline`` of bench.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from .fgsm import fgsm_generate
from .losses import cross_entropy_2d, jsd_2d, kl_divergence_2d, softmax_channels


@dataclass
class OracleModel:
    """What the step needs from a reference ``Segmentator``: the net and its optimizer."""
    net: torch.nn.Module
    optimizer: torch.optim.Optimizer

    @classmethod
    def make(cls, net, lr=1e-3, weight_decay=1e-4):
        # models/segmentators.py:41 + config/ACDC_config_cotraing.yaml:5-8
        return cls(net, torch.optim.Adam(net.parameters(), lr=lr, weight_decay=weight_decay))


def cotrain_step(models: Sequence[OracleModel],
                 lab_batches: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                 unlab_img: Optional[torch.Tensor],
                 train_jsd: bool, train_adv: bool,
                 lam_cot: float = 0.0, lam_adv: float = 0.0, eps: float = 0.05,
                 adv_choice: Tuple[int, int] = (0, 1)) -> dict:
    S = len(models)
    sup: List[torch.Tensor] = []
    preds: List[torch.Tensor] = []
    total = 0
    for i in range(S):                                  # :208-218
        img, gt = lab_batches[i]
        pred = models[i].net(img)
        loss = cross_entropy_2d(pred, gt.squeeze(1))
        sup.append(loss)
        preds.append(pred)
        total = total + loss
    jsd = 0
    unlab_probs: List[torch.Tensor] = []
    if train_jsd:                                       # :219-227
        unlab_probs = [softmax_channels(m.net(unlab_img)) for m in models]
        jsd = jsd_2d(unlab_probs).mean()
    adv = 0
    extras = {}
    if train_adv:                                       # :233-244 -> :371-392
        a, b = adv_choice
        img_b, gt_b = lab_batches[b]
        x = torch.cat((img_b, unlab_img), dim=0)
        x_adv, noise, real, gx = fgsm_generate(models[b].net, x, gt_b, eps)
        adv_p = softmax_channels(models[a].net(x_adv))
        adv = kl_divergence_2d(adv_p, real.detach(), reduce=True)
        extras = dict(x_adv=x_adv, noise=noise, grad_x=gx, real=real.detach())
    for m in models:                                    # :245
        m.optimizer.zero_grad()
    total = total + lam_cot * jsd + lam_adv * adv       # :246
    total.backward()                                    # :247
    for m in models:                                    # :248
        m.optimizer.step()
    return dict(sup=[s.detach() for s in sup],
                jsd=jsd.detach() if train_jsd else 0,
                adv=adv.detach() if train_adv else 0,
                total=total.detach(), preds=[p.detach() for p in preds],
                unlab_probs=[p.detach() for p in unlab_probs], **extras)

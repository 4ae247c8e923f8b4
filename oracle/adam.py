"""Scalar-clear restatement of the Adam update the reference gets from torch.optim.Adam
(models/segmentators.py:37-43; config/ACDC_config_cotraing.yaml:5-8).  TEST INFRASTRUCTURE ONLY.

Follows torch 2.10 ``_single_tensor_adam`` op order (L2 weight decay folded into the grad).
"""
from __future__ import annotations

import math

import torch


def adam_reference_step(p, g, m, v, step: int, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """In-place fp32 update of (p, m, v) given grad g; ``step`` is the 1-based step count."""
    if weight_decay != 0:
        g = g.add(p, alpha=weight_decay)
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)
    return p, m, v

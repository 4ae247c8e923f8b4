"""Pixel-wise losses of the hot path, fp32 CPU restatement (TEST INFRASTRUCTURE ONLY).

All take NCHW tensors.  Citations are to /root/reference/generalframework/loss/loss.py.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn.functional as F


def softmax_channels(logits: torch.Tensor) -> torch.Tensor:
    # models/segmentators.py:50  F.softmax(pred_logit, 1)
    return F.softmax(logits, dim=1)


def cross_entropy_2d(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = 255) -> torch.Tensor:
    """loss.py:12-25: NLLLoss(mean over non-ignored pixels)(log_softmax(logits,1), target[B,H,W])."""
    return F.nll_loss(F.log_softmax(logits, dim=1), target, ignore_index=ignore_index, reduction="mean")


def entropy_2d(prob: torch.Tensor) -> torch.Tensor:
    """loss.py:70-84: -sum_c p*log(p+1e-16) -> [B,H,W]."""
    return -(prob * (prob + 1e-16).log()).sum(1)


def jsd_2d(probs: Sequence[torch.Tensor]) -> torch.Tensor:
    """loss.py:183-196: H(mean_i p_i) - mean_i H(p_i) -> [B,H,W] (caller takes .mean())."""
    mean_p = sum(probs[1:], probs[0]) / len(probs)
    mean_h = sum(entropy_2d(p) for p in probs) / len(probs)
    return entropy_2d(mean_p) - mean_h


def kl_divergence_2d(p_prob: torch.Tensor, y_prob: torch.Tensor, reduce: bool = False,
                     eps: float = 1e-10) -> torch.Tensor:
    """loss.py:110-134: sum_c y*(log(y+eps) - log(p+eps)); mean over B,H,W when reduce."""
    logp = (p_prob + eps).log()
    logy = (y_prob + eps).log()
    kl = (y_prob * logy).sum(1) - (y_prob * logp).sum(1)
    return kl.mean() if reduce else kl

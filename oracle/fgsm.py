"""FGSM adversarial example generation, restating utils/AEGenerator.py:16-51 (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import torch

from .losses import cross_entropy_2d, softmax_channels


def fgsm_generate(net, img: torch.Tensor, gt: torch.Tensor, eps: float, criterion=cross_entropy_2d):
    """Returns (x_adv, noise, softmax(pred), grad_x).

    gt is [B_l,1,H,W]; when img has more samples than gt the tail is pseudo-labelled with
    argmax(pred) (AEGenerator.py:24-25).  No clamp (AEGenerator.py:47).  Net grads are
    cleared before and after, as in the reference (:22,:30).
    """
    assert img.dim() == 4 and img.shape[0] >= gt.shape[0]
    x = img.detach().clone().requires_grad_(True)
    net.zero_grad()
    pred = net(x)
    if x.shape[0] > gt.shape[0]:
        gt = torch.cat((gt, pred.max(1)[1][gt.shape[0]:].unsqueeze(1)), dim=0)
    loss = criterion(pred, gt.squeeze(1))
    loss.backward()
    g = x.grad.detach().clone()
    noise = eps * g.sign()
    x_adv = (x + noise).detach()
    net.zero_grad()
    return x_adv, noise.detach(), softmax_channels(pred), g

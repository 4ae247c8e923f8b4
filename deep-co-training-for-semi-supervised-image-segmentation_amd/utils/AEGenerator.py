"""FGSM adversarial-example generator with the reference's interface
(/root/reference/generalframework/utils/AEGenerator.py:9-51).

``FSGMGenerator(net, eplision)(img, gt, criterion) -> (adv_img, noise, softmax(pred))``.
Differences that are not observable by the caller:
  * the backward to the input skips the weight-gradient GEMMs (the reference computes them and
    then ``net.zero_grad()``s them away, :27-30) -- parameters are frozen for the duration;
  * ``x + eps*sign(g)`` is one HIP kernel (K11), the pseudo-label argmax another (K10).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import hip_ops as K
from ..loss.loss import softmax_channels, _pc


class FSGMGenerator(object):
    def __init__(self, net: nn.Module, eplision: float = 0.05) -> None:
        super().__init__()
        self.net = net
        self.eplision = eplision

    def __call__(self, img: Tensor, gt: Tensor, criterion: nn.Module) -> Tuple[Tensor, Tensor, Tensor]:
        assert img.shape.__len__() == 4
        assert img.shape[0] >= gt.shape[0]
        img.requires_grad = True
        if img.grad is not None:
            img.grad.zero_()
        self.net.zero_grad()
        params = [p for p in self.net.parameters() if p.requires_grad]
        for p in params:
            p.requires_grad_(False)
        try:
            pred = self.net(img)
            if img.shape[0] > gt.shape[0]:
                lp = _pc(pred.detach())
                pseudo = K.argmax(lp, lp.shape[3]).view(img.shape[0], 1, img.shape[2], img.shape[3])
                gt = torch.cat((gt, pseudo[gt.shape[0]:]), dim=0)
            loss = criterion(pred, gt.squeeze(1))
            (grad,) = torch.autograd.grad(loss, img, retain_graph=False)
        finally:
            for p in params:
                p.requires_grad_(True)
        adv_img, noise = self.adversarial_fgsm(img, grad, epsilon=self.eplision)
        self.net.zero_grad()
        return adv_img.detach(), noise.detach(), softmax_channels(pred.detach())

    @staticmethod
    def adversarial_fgsm(image: Tensor, data_grad: Tensor, epsilon: float = 0.01) -> Tuple[Tensor, Tensor]:
        """perturbed = image + epsilon*sign(grad); no clamp (AEGenerator.py:35-51)."""
        x = image.detach().to(torch.float32).contiguous()
        g = data_grad.detach().to(torch.float32).contiguous()
        xa, noise = K.fgsm_step(x, g, epsilon)
        return xa, noise

"""Architecture registry for the hot path (reference: generalframework/arch/__init__.py:17-81).

Only the networks the co-training scripts use are provided, as HIP execution plans:
``unet`` (arch/network.py:196-240), ``unet_bn`` (:243-290) and ``enet`` (arch/enet.py:234-243)."""
from __future__ import annotations

import torch
from torch import nn

from .enet import Enet
from .unet import UNet, UNet_bn, _ConvP

__all__ = ['weights_init', 'get_arch', 'ARCH_CALLABLES']

ARCH_CALLABLES = {}


def _register_arch(arch, callable, alias=None):
    if arch in ARCH_CALLABLES:
        raise ValueError('{} already exists!'.format(arch))
    ARCH_CALLABLES[arch] = callable


_register_arch('unet', UNet)
_register_arch('unet_bn', UNet_bn)
_register_arch('enet', Enet)


def weights_init(m):
    """arch/__init__.py:60-65: xavier_normal_ on conv / transposed-conv weights, BN gamma ~ N(1, .02), beta = 0."""
    if isinstance(m, _ConvP):
        nn.init.xavier_normal_(m.weight.data)
    elif type(m) == nn.BatchNorm2d or getattr(m, "is_dct_batchnorm", False):
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def get_arch(arch, kwargs) -> nn.Module:
    """Get the architecture (arch/__init__.py:68-81).  Returns a torch.nn.Module."""
    arch_callable = ARCH_CALLABLES.get(arch)
    kwargs = dict(kwargs)
    kwargs.pop('arch', None)
    assert arch_callable, "Architecture {} is not found!".format(arch)
    net = arch_callable(**kwargs)
    net.apply(weights_init)
    if hasattr(net, "mark_weights_updated"):
        net.mark_weights_updated()
    return net

"""``Segmentator``: network + optimizer + LR scheduler bundle with the reference's interface
(/root/reference/generalframework/models/segmentators.py:17-116).

What is different underneath:
  * ``arch_dict['name']`` resolves to a HIP execution plan (dct_amd.arch) instead of an ATen
    module graph; ``torchnet`` may also be injected (any nn.Module) via ``torchnet=``;
  * ``optim_dict['name'] == 'Adam'`` on a flat-parameter HIP net becomes ``FusedAdam`` (one
    kernel per step, same maths and ``state_dict``); other optimizer names resolve in
    ``torch.optim`` exactly as the reference does;
  * no ``nn.DataParallel``: multi-GPU is one process per GPU with an RCCL all-reduce of the flat
    gradient buffer (dct_amd.ddp), which preserves DataParallel's per-replica BN statistics.
"""
from __future__ import annotations

from typing import List

import torch
from torch import Tensor, nn
from torch import optim
from torch.optim import lr_scheduler

from .. import ModelMode
from ..arch import get_arch


class Segmentator(object):
    def __init__(self, arch_dict: dict, optim_dict: dict, scheduler_dict: dict, torchnet: nn.Module = None,
                 softmax_fn=None) -> None:
        super().__init__()
        self._softmax_fn = softmax_fn   # host-logic tests inject one for injected (non-HIP) nets
        self.arch_dict = arch_dict
        self.optim_dict = optim_dict
        self.scheduler_dict = scheduler_dict
        self.torchnet, self.optimizer, self.scheduler = self.__setup(torchnet)

    def __setup(self, torchnet):
        self.arch_name = self.arch_dict['name']
        self.arch_params = {k: v for k, v in self.arch_dict.items() if k != 'name'}
        self.optim_name = self.optim_dict['name']
        self.optim_params = {k: v for k, v in self.optim_dict.items() if k != 'name'}
        self.scheduler_name = self.scheduler_dict['name']
        self.scheduler_params = {k: v for k, v in self.scheduler_dict.items() if k != 'name'}
        if torchnet is None:
            torchnet = get_arch(self.arch_name, self.arch_params)
        flat = getattr(torchnet, "flat_params", None)
        if self.optim_name == "Adam" and flat is not None:
            from ..optim import FusedAdam
            optimizer = FusedAdam(torchnet.parameters(), flat=flat,
                                  on_step=getattr(torchnet, "mark_weights_updated", None), **self.optim_params)
        else:
            optimizer = getattr(optim, self.optim_name)(torchnet.parameters(), **self.optim_params)
        scheduler = getattr(lr_scheduler, self.scheduler_name)(optimizer, **self.scheduler_params)
        return torchnet, optimizer, scheduler

    def predict(self, img: Tensor, logit=True) -> Tensor:
        pred_logit = self.torchnet(img)
        if logit:
            return pred_logit
        if self._softmax_fn is not None:
            return self._softmax_fn(pred_logit)
        from ..loss.loss import softmax_channels
        return softmax_channels(pred_logit)   # HIP kernel; rejects CPU tensors

    @property
    def training(self):
        return self.torchnet.training

    def update(self, img: Tensor, gt: Tensor, criterion, mode=ModelMode.TRAIN) -> List[Tensor]:
        assert img.shape.__len__() == 4
        assert gt.shape.__len__() == 4
        if mode == ModelMode.TRAIN:
            self.train()
            self.optimizer.zero_grad()
            pred = self.predict(img)
            loss = criterion(pred, gt.squeeze(1))
            loss.backward()
            self.optimizer.step()
        else:
            self.eval()
            with torch.no_grad():
                pred = self.predict(img)
                loss = criterion(pred, gt.squeeze(1))
        self.train()
        return [pred.data, loss.data]

    def schedulerStep(self):
        self.scheduler.step()

    @property
    def state_dict(self):
        return {'arch_dict': self.arch_dict, 'optim_dict': self.optim_dict, 'scheduler_dict': self.scheduler_dict,
                'net_state_dict': self.torchnet.state_dict(), 'optim_state_dict': self.optimizer.state_dict(),
                'scheduler_state_dict': self.scheduler.state_dict()}

    def load_state_dict(self, state_dict: dict):
        net_sd = state_dict['net_state_dict']
        if any(k.startswith("module.") for k in net_sd):   # nn.DataParallel checkpoints (segmentators.py:88-93)
            net_sd = {k.replace("module.", ""): v for k, v in net_sd.items()}
        self.torchnet.load_state_dict(net_sd)
        self.optimizer.load_state_dict(state_dict['optim_state_dict'])
        self.scheduler.load_state_dict(state_dict['scheduler_state_dict'])

    def to(self, device: torch.device):
        self.torchnet.to(device)
        for state in self.optimizer.state.values():
            for k, v in state.items():
                if isinstance(v, torch.Tensor) and v.dim() > 0:
                    state[k] = v.to(device)

    def set_mode(self, mode):
        assert mode in (ModelMode.TRAIN, ModelMode.EVAL) or mode in ('train', 'eval')
        if mode in (ModelMode.TRAIN, 'train'):
            self.train()
        elif mode in (ModelMode.EVAL, 'eval'):
            self.eval()

    def eval(self):
        self.torchnet.eval()

    def train(self):
        self.torchnet.train()

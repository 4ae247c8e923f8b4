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
        """One supervised step (TRAIN) or one loss evaluation (anything else) on a batch; returns [prediction, loss] detached and
        leaves the network in training mode, as the reference's ``update`` does (segmentators.py:60-77)."""
        if img.dim() != 4 or gt.dim() != 4:
            raise AssertionError(f"expected [B, C, H, W] image and [B, 1, H, W] target, got {tuple(img.shape)} / {tuple(gt.shape)}")
        learn = mode == ModelMode.TRAIN
        self.torchnet.train(learn)
        with torch.set_grad_enabled(learn):
            pred = self.predict(img)
            loss = criterion(pred, gt.squeeze(1))
        if learn:
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
        self.torchnet.train(True)
        return [pred.detach(), loss.detach()]

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
        """network AND optimizer state (moment buffers; not the 0-d step counters) onto ``device``"""
        self.torchnet.to(device)
        for slot in self.optimizer.state.values():
            moved = {name: t.to(device) for name, t in slot.items() if torch.is_tensor(t) and t.dim() > 0}
            slot.update(moved)

    _MODES = {ModelMode.TRAIN: True, 'train': True, ModelMode.EVAL: False, 'eval': False}

    def set_mode(self, mode):
        if mode not in self._MODES:
            raise AssertionError(f"mode must be ModelMode.TRAIN / ModelMode.EVAL (or 'train' / 'eval'), got {mode!r}")
        self.torchnet.train(self._MODES[mode])

    def eval(self):
        self.torchnet.eval()

    def train(self):
        self.torchnet.train()

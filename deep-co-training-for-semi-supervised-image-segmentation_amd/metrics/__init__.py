"""Meters used inside the step (reference: generalframework/metrics/dice_meter.py:12-83,
averagemeter.py:3-48), kept on the device: ``add`` launches one counting kernel and never
synchronises; ``value`` is where the (tiny) results are read."""
from __future__ import annotations

import math

import torch

from .. import hip_ops as K

__all__ = ["DiceMeter", "AverageValueMeter"]


class DiceMeter(object):
    def __init__(self, method='2d', report_axises='all', C=4) -> None:
        assert method in ('2d', '3d')
        assert report_axises == 'all' or isinstance(report_axises, list)
        self.method = method
        self.report_axis = report_axises
        self.diceLog = []
        self.C = C
        self._acc = None        # float64 [2, C + 1] on the device: running sum / sum of squares per class and of the report mean
        self._n = 0
        self._cache = None

    def reset(self):
        self.diceLog = []
        self._acc = None
        self._n = 0
        self._cache = None

    def add(self, pred_logit: torch.Tensor, gt: torch.Tensor, smooth: float = 1e-8):
        """pred_logit [B,C,H,W] (logits or probabilities: only the argmax matters, dice_meter.py:28-32),
        gt [B,1,H,W] int64."""
        if not pred_logit.is_cuda:
            raise RuntimeError("dct_amd DiceMeter counts on the HIP device only (no CPU fallback)")
        B, C = pred_logit.shape[0], pred_logit.shape[1]
        lp = pred_logit.detach().permute(0, 2, 3, 1)
        if lp.dtype != torch.float32 or not lp.is_contiguous():
            lp = lp.to(torch.float32).contiguous()
        g = gt.reshape(B, -1)
        if g.dtype != torch.int64 or not g.is_contiguous():
            g = g.to(torch.int64).contiguous()
        inter, ps, gs = K.dice_counts(lp, g, B, C)
        # the Dice rows and the running moments behind value() in one more launch: three launches per add, nothing that scales
        # with the history (the reference re-concatenates the whole log 4 S times per reported step, :251-264)
        if self._acc is None or self._acc.device != inter.device:
            self._acc = torch.zeros(2, C + 1, dtype=torch.float64, device=inter.device)
        axes = range(C) if self.report_axis == 'all' else self.report_axis
        mask = sum(1 << int(a) for a in axes)
        if B <= 64:
            dice = K.dice_update(inter, ps, gs, self.method == '3d', mask, smooth, self._acc)
        else:       # patient batches of more than 64 slices: the same arithmetic in torch ops
            if self.method == '3d':
                inter, ps, gs = inter.sum(0, keepdim=True), ps.sum(0, keepdim=True), gs.sum(0, keepdim=True)
            dice = (2 * inter.float() + smooth) / ((ps + gs).float() + smooth)
            rep = dice[:, list(axes)].mean(1, keepdim=True)
            row = torch.cat((dice, rep), dim=1).double()
            self._acc += torch.stack((row.sum(0), (row * row).sum(0)))
        self.diceLog.append(dice)
        self._n += dice.shape[0]

    @property
    def log(self):
        if self.diceLog:
            log = torch.cat(self.diceLog)
        else:
            log = torch.zeros(1, self.C)
        return log

    def value(self, **kwargs):
        if self._acc is None:               # nothing added yet: the reference reports over one row of zeros
            log = self.log
            report_means = log.mean(1) if self.report_axis == 'all' else log[:, self.report_axis].mean(1)
            return (report_means.mean(), report_means.std()), (log.mean(0), log.std(0))
        # ONE device->host copy of the 2 x (C + 1) running sums per reading (cached until the next add): the per-class floats
        # the progress bar then takes (4 S readings of C values each, cotraining_totalloss.py:251-264) cost no further
        # synchronisation -- a blocking read of a device scalar is ~ms on this stack, the arithmetic below is nothing
        if self._cache is None or self._cache[0] != self._n:
            acc = self._acc.cpu()
            n = self._n
            mean = acc[0] / n
            var = (acc[1] - n * mean * mean).clamp_min(0.0) / (n - 1) if n > 1 else torch.full_like(mean, float('nan'))
            mean, std = mean.float(), var.sqrt().float()
            self._cache = (n, ((mean[-1], std[-1]), (mean[:-1], std[:-1])))
        return self._cache[1]

    def detailed_summary(self) -> dict:
        _, (means, _) = self.value()
        return {f'DSC{i}': means[i].item() for i in range(len(means))}

    def summary(self) -> dict:
        (means, var), (_, _) = self.value()
        return {'mDSC': means.item(), 'mVars': var.item()}


class AverageValueMeter(object):
    """Running mean/std of scalars.  Device scalars are kept as tensors and only read in value()."""

    def __init__(self, name='Average Meter'):
        self.name = name
        self.reset()

    def reset(self):
        self._vals = []

    def add(self, value, n=1):
        self._vals.append(value)

    def _floats(self):
        if self._vals and any(isinstance(v, torch.Tensor) for v in self._vals):
            ts = [v.detach().float().reshape(()) if isinstance(v, torch.Tensor) else torch.tensor(float(v)) for v in self._vals]
            dev = next(t.device for t in ts if t.is_cuda) if any(t.is_cuda for t in ts) else ts[0].device
            self._vals = torch.stack([t.to(dev) for t in ts]).cpu().tolist()
        return self._vals

    def value(self):
        v = self._floats()
        n = len(v)
        if n == 0:
            return math.nan, math.nan
        mean = sum(v) / n
        if n == 1:
            return mean, math.inf
        var = sum((x - mean) ** 2 for x in v) / (n - 1.0)
        return mean, math.sqrt(var)

    def summary(self) -> dict:
        return {'mean': self.value()[0], 'val': self.value()[1]}

    detailed_summary = summary

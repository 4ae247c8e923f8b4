"""Checkpoint ensemble summary (reference: /root/reference/Summary.py:70-252, the SURVEY.md 8f "next" row 3).

``load_model`` / ``load_models`` rebuild ``Segmentator``s from ``best_{i}.pth`` checkpoints exactly as Summary.py:70-79 does;
``Ensembleway`` is the soft / hard voting of :92-126; ``summarize`` is the evaluation loop of :148-172 + the result tables of
:176-205 for the Dice part (2-D per slice and 3-D per patient batch, per model and for the ensemble).  Hausdorff distance
(needs the external ``deepclustering`` package, absent from the reference tree) and the kappa table are out of scope.

Predictions come from the HIP networks; voting and the Dice counting run on the device (``dct_dice_counts``)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import Tensor

from .metrics import DiceMeter
from .models import Segmentator


def load_model(checkpoint, map_location="cpu") -> Segmentator:
    """Summary.py:70-74 (+ :140-146: the weights are loaded too)."""
    state = torch.load(checkpoint, map_location=torch.device(map_location), weights_only=False)
    sd = state['segmentator']
    model = Segmentator(arch_dict=sd['arch_dict'], optim_dict=sd['optim_dict'], scheduler_dict=sd['scheduler_dict'])
    model.load_state_dict(sd)
    return model


def load_models(checkpoints) -> List[Segmentator]:
    return [load_model(c) for c in checkpoints]


class Ensembleway(object):
    def __init__(self, ensembleway: str, num_classes: Optional[int] = None) -> None:
        assert ensembleway in ('soft', 'hard'), ensembleway
        self.ensembleway = ensembleway
        self.num_classes = num_classes

    def __call__(self, predicts: List[Tensor]) -> Tensor:
        return self._softVoting(predicts) if self.ensembleway == 'soft' else self._hardVoting(predicts, self.num_classes)

    @staticmethod
    def _softVoting(predicts: List[Tensor]) -> Tensor:
        assert isinstance(predicts, list), type(predicts)
        out = torch.stack(predicts, dim=0).mean(0)
        assert out.shape == predicts[0].shape
        return out

    @staticmethod
    def _hardVoting(predicts: List[Tensor], num_classes: Optional[int] = None) -> Tensor:
        """Majority of the argmax maps, ties to the smallest class (np.bincount(...).argmax(), Summary.py:120), as a one-hot
        float map.  Like the reference it votes over the concatenated batch axis, i.e. it expects single-slice batches."""
        assert isinstance(predicts, list), type(predicts)
        C = num_classes or predicts[0].shape[1]
        votes = torch.cat([p.max(1)[1] for p in predicts], 0)
        counts = torch.stack([(votes == c).sum(0) for c in range(C)])
        winner = counts.max(0)[1]
        return torch.nn.functional.one_hot(winner, C).permute(2, 0, 1).unsqueeze(0).float()


@torch.no_grad()
def summarize(models: List[Segmentator], val_dataloader, device, ensemble_method: str = 'soft',
              report_axises: Optional[List[int]] = None) -> Dict[str, dict]:
    """Per-model and ensemble 2-D / 3-D Dice over a validation loader (batches ``[(img, gt), meta, names]``)."""
    device = torch.device(device)
    C = models[0].arch_params['num_classes']
    axes = report_axises if report_axises is not None else list(range(C))
    ens = Ensembleway(ensemble_method, C)
    for m in models:
        m.to(device)
        m.eval()
    d2 = [DiceMeter(method='2d', report_axises=axes, C=C) for _ in models]
    d3 = [DiceMeter(method='3d', report_axises=axes, C=C) for _ in models]
    e2, e3 = DiceMeter(method='2d', report_axises=axes, C=C), DiceMeter(method='3d', report_axises=axes, C=C)
    for (img, gt), _, _ in val_dataloader:
        img, gt = img.to(device), gt.to(device)
        preds = [m.predict(img, logit=False) for m in models]
        for j, p in enumerate(preds):
            d2[j].add(p, gt)
            d3[j].add(p, gt)
        v = ens(preds)
        e2.add(v, gt)
        e3.add(v, gt)

    def table(meter):
        (_, _), (means, stds) = meter.value()
        return {f'DSC{j}': float(means[j]) for j in range(C)}, {f'DSC{j}': float(stds[j]) for j in range(C)}

    out: Dict[str, dict] = {}
    for name, meters in (("2d", (d2, e2)), ("3d", (d3, e3))):
        res = {f'model_{i}': table(m)[0] for i, m in enumerate(meters[0])}
        res['ensemble'] = table(meters[1])[0]
        res['ensemble_std'] = table(meters[1])[1]
        out[name] = res
    return out

"""Dice scores as DiceMeter computes them (metrics/dice_meter.py:12-32; utils/utils.py:187-207).
TEST INFRASTRUCTURE ONLY."""
import torch


def _one_hots(pred_logit: torch.Tensor, gt: torch.Tensor):
    C = pred_logit.shape[1]
    cls = pred_logit.argmax(1)  # argmax(softmax(x)) == argmax(x)
    ohp = torch.stack([cls == c for c in range(C)], 1).to(torch.int32)
    ohg = torch.stack([gt.squeeze(1) == c for c in range(C)], 1).to(torch.int32)
    return ohp, ohg


def dice_2d(pred_logit, gt, smooth: float = 1e-8) -> torch.Tensor:
    """per-slice dice -> [B,C]  ("bcwh->bc")."""
    p, g = _one_hots(pred_logit, gt)
    inter = (p & g).sum((2, 3)).float()
    sizes = (p.sum((2, 3)) + g.sum((2, 3))).float()
    return (2 * inter + smooth) / (sizes + smooth)


def dice_3d(pred_logit, gt, smooth: float = 1e-8) -> torch.Tensor:
    """per-batch (patient) dice -> [C]  ("bcwh->c")."""
    p, g = _one_hots(pred_logit, gt)
    inter = (p & g).sum((0, 2, 3)).float()
    sizes = (p.sum((0, 2, 3)) + g.sum((0, 2, 3))).float()
    return (2 * inter + smooth) / (sizes + smooth)

"""Voting ensembles of the reference's offline summary (TEST INFRASTRUCTURE ONLY).

Restates ``Ensembleway`` of /root/reference/Summary.py:92-126 (soft voting :104-110, hard voting :112-126).  Pinned by
tests/golden/g7_eval.npz, captured from that class taken out of the reference script (tools/capture_golden.py::g7_eval).
"""
from __future__ import annotations

from typing import List

import torch


def soft_vote(probs: List[torch.Tensor]) -> torch.Tensor:
    """Mean of the S probability maps [B,C,H,W] (Summary.py:104-110)."""
    return torch.stack(probs, dim=0).mean(0)


def hard_vote(probs: List[torch.Tensor], num_classes: int) -> torch.Tensor:
    """Per-pixel majority of the S argmax maps, ties to the smallest class index (np.bincount(...).argmax()), returned
    one-hot as float [1,C,H,W].  The reference concatenates the argmax maps along the BATCH axis and votes over it
    (Summary.py:113-125), i.e. it is defined for single-slice batches; the same is done here."""
    votes = torch.cat([p.max(1)[1] for p in probs], 0)                      # [S*B, H, W]
    counts = torch.stack([(votes == c).sum(0) for c in range(num_classes)])  # [C, H, W]
    winner = counts.max(0)[1]                                                # first maximum = smallest class index
    return torch.nn.functional.one_hot(winner, num_classes).permute(2, 0, 1).unsqueeze(0).float()

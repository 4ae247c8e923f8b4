"""Shared test scaffolding: synthetic loaders with the batch format the trainer consumes
(``[[img, gt], meta, names]``, cotraining_totalloss.py:209,221,292)."""
import numpy as np
import torch


class FakeDataset:
    def __init__(self):
        from dct_amd import ModelMode
        self.training = ModelMode.EVAL

    def set_mode(self, mode):
        self.training = mode


class FakeLoader(list):
    def __init__(self, batches, batch_size):
        super().__init__(batches)
        self.batch_size = batch_size
        self.dataset = FakeDataset()


def batches(seed, n, B, H, C, W=None):
    """same generator protocol as tools/capture_golden.py::_batches"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        img = torch.rand(B, 1, H, W or H, generator=g)
        gt = torch.randint(0, C, (B, 1, H, W or H), generator=g)
        out.append([[img, gt], None, [f"s{seed}_{i}_{j}" for j in range(B)]])
    return out


def digest(t):
    t = t.detach().double().flatten().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), t.norm().item(), t.abs().max().item()])

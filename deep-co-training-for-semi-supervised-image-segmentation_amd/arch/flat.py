"""Flat parameter / gradient storage for one network.

All parameters of a net live in ONE fp32 buffer (and all gradients in another) so that
  * Adam is a single launch over the flat buffers (K12),
  * the data-parallel gradient exchange is a handful of large RCCL all-reduces straight out of
    the gradient buffer (no bucket copies),
  * weight-gradient kernels accumulate in place ("+=" across the 2-3 backward passes a model
    sees per step) instead of materialising one temporary per pass.
Each nn.Parameter stays a normal tensor *view* (same logical shape, same strides) of the flat
buffer, so ``state_dict`` / ``load_state_dict`` / ``.to(device)`` keep working; ``ensure()``
re-flattens transparently after anything that re-allocated the parameters.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

ALIGN = 64  # elements: 256-byte aligned slots (16-B vector loads in every kernel)


def _dense_span_ok(t: torch.Tensor) -> bool:
    # dense, non-overlapping in *some* dim order (contiguous or channels_last)
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class FlatParams:
    def __init__(self, params: List[nn.Parameter]):
        self.params = list(params)
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat: torch.Tensor = None
        self.gflat: torch.Tensor = None
        self.version = 0          # bumps whenever the flat buffer is rebuilt
        # optional bf16 image of the flat buffer (same offsets), written by the fused Adam in the same pass
        # that updates the masters: the forward weight "packs" of a bf16 network are views of it
        self.want_shadow = False
        self.shadow: torch.Tensor = None

    # -- parameters -------------------------------------------------------------------------
    def is_flat(self) -> bool:
        if self.flat is None:
            return False
        base = self.flat.data_ptr()
        for p, off in zip(self.params, self.offsets):
            if p.device != self.flat.device or p.data_ptr() != base + 4 * off:
                return False
        return True

    def ensure(self) -> bool:
        """Make every parameter a view of the flat buffer.  Returns True when it had to rebuild."""
        if self.is_flat():
            return False
        dev = self.params[0].device
        flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            assert p.dtype == torch.float32 and _dense_span_ok(p.data), "flat params must be dense fp32"
            v = flat.as_strided(p.shape, p.stride(), off)
            v.copy_(p.data)
            p.data = v
        self.flat, self.gflat = flat, None
        self.shadow = None
        self.version += 1
        return True

    def ensure_shadow(self) -> torch.Tensor:
        if self.shadow is None or self.shadow.device != self.flat.device:
            self.shadow = torch.empty(self.total, dtype=torch.bfloat16, device=self.flat.device)
        return self.shadow

    def shadow_dense(self, i: int) -> torch.Tensor:
        off = self.offsets[i]
        return self.shadow[off:off + self.params[i].numel()]

    def dense(self, i: int) -> torch.Tensor:
        """1-D view of parameter i's storage span (physical order)."""
        off = self.offsets[i]
        return self.flat[off:off + self.params[i].numel()]

    # -- gradients --------------------------------------------------------------------------
    def _grad_view(self, i: int) -> torch.Tensor:
        p = self.params[i]
        return self.gflat.as_strided(p.shape, p.stride(), self.gflat.storage_offset() + self.offsets[i])   # gflat may be a slice (ddp arena)

    def grads_attached(self) -> bool:
        if self.gflat is None or self.gflat.device != self.flat.device:
            return False
        base = self.gflat.data_ptr()
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return False
        return True

    def ensure_grads(self) -> bool:
        """Attach ``p.grad`` views of the flat gradient buffer (zeroed) unless already attached.
        ``optimizer.zero_grad()`` (set_to_none) detaches them, so the next backward starts from
        a zeroed buffer -- exactly the semantics of accumulating into fresh grads."""
        if self.grads_attached():
            return False
        if self.gflat is None or self.gflat.device != self.flat.device:
            self.gflat = torch.zeros(self.total, dtype=torch.float32, device=self.flat.device)
        else:
            self.gflat.zero_()
        for i, p in enumerate(self.params):
            p.grad = self._grad_view(i)
        return True

    def grad_dense(self, i: int) -> torch.Tensor:
        off = self.offsets[i]
        return self.gflat[off:off + self.params[i].numel()]

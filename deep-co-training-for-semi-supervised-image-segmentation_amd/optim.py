"""Fused Adam over flat parameter buffers (K12): drop-in for ``torch.optim.Adam`` as the
reference constructs it (models/segmentators.py:37-43; config/ACDC_config_cotraing.yaml:5-8).

Same hyper-parameters, same update rule (L2 weight decay folded into the gradient, bias
correction as torch 2.x), same ``state_dict`` layout (per-parameter ``step`` / ``exp_avg`` /
``exp_avg_sq``) -- but ONE kernel launch per network instead of hundreds of tiny ones.
"""
from __future__ import annotations

import math
from typing import List

import torch

from . import hip_ops
from .arch.flat import FlatParams


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, flat: FlatParams = None,
                 on_step=None):
        if flat is None:
            raise ValueError("FusedAdam needs the network's FlatParams (use net.flat_params)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        plist = [p for g in self.param_groups for p in g["params"]]
        assert len(self.param_groups) == 1 and len(plist) == len(flat.params) and all(a is b for a, b in zip(plist, flat.params)), \
            "FusedAdam covers exactly one network's parameters, in order"
        self.flat = flat
        self._m = None
        self._v = None
        self._steps = 0
        self._import_steps = False
        self._flat_version = -1
        self._on_step = on_step
        self.grad_scale = 1.0       # multiplies the gradients inside the update: the inverse loss scale of an fp16 step
        # {step count, lr} as float64[2] ON THE DEVICE: the kernel forms the bias corrections itself
        # (dct_adam_flat_dev), so step() has no per-step host scalar and can be replayed from a HIP graph.
        self._dev_state = None
        self._dev_table = None
        self._dev_lr = None
        self._table_base = 0

    def _state_views(self):
        f = self.flat
        for p, off in zip(f.params, f.offsets):
            st = self.state[p]
            st["exp_avg"] = self._m.as_strided(p.shape, p.stride(), off)
            st["exp_avg_sq"] = self._v.as_strided(p.shape, p.stride(), off)
            st["step"] = torch.tensor(float(self._steps))

    TABLE_STEPS = 2048

    def _ensure_dev_state(self, dev, lr):
        """Device state of dct_adam_flat_dev: {t, lr, table base, table length} + the table of host-computed
        {1 - beta1^t, sqrt(1 - beta2^t)} (python doubles) for the next TABLE_STEPS steps; the kernel forms
        lr / (1 - beta1^t) in double and rounds to fp32 -- exactly what torch.optim.Adam hands its kernels.  The table is
        rebuilt when it runs out or the moments were re-imported; a learning-rate change is one async fill of state[1]
        (a per-step schedule costs nothing more).  Never inside a captured graph."""
        lr = float(lr)
        fresh = self._dev_state is None or self._dev_state.device != dev
        if not fresh and self._steps + 1 <= self._table_base + self.TABLE_STEPS:
            if lr != self._dev_lr:
                self._dev_state[1:2].fill_(lr)
                self._dev_lr = lr
            return self._dev_state
        b1, b2 = self.param_groups[0]["betas"]
        base = self._steps
        rows = [(1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t)) for t in range(base + 1, base + 1 + self.TABLE_STEPS)]
        table = torch.tensor(rows, dtype=torch.float64)
        state = torch.tensor([float(self._steps), lr, float(base), float(self.TABLE_STEPS)], dtype=torch.float64)
        if fresh:
            self._dev_state, self._dev_table = state.to(dev), table.to(dev)
        else:                                     # in place: captured graphs hold these addresses
            self._dev_table.copy_(table)
            self._dev_state[1:].copy_(state[1:])  # the device owns the step count
        self._dev_lr, self._table_base = lr, base
        return self._dev_state

    def refresh_lr(self):
        """Push a changed learning rate / an exhausted table to the device (call between graph replays)."""
        if self._dev_state is not None:
            self._ensure_dev_state(self._dev_state.device, self.param_groups[0]["lr"])

    def note_replayed_steps(self, n: int = 1):
        """A captured graph containing step() was replayed n times: keep the host step count in line with the
        device counter (state_dict() reports it)."""
        self._steps += int(n)

    def _ensure_state(self):
        f = self.flat
        f.ensure()
        dev = f.flat.device
        ok = self._m is not None and self._m.device == dev
        if ok:
            base_m = self._m.data_ptr()
            for p, off in zip(f.params, f.offsets):
                st = self.state.get(p)
                if not st or "exp_avg" not in st or st["exp_avg"].data_ptr() != base_m + 4 * off:
                    ok = False
                    break
        if ok:
            return
        # (re)build flat moment buffers, importing whatever per-parameter state exists
        # (e.g. after load_state_dict of a reference checkpoint or after .to(device))
        m = torch.zeros(f.total, dtype=torch.float32, device=dev)
        v = torch.zeros(f.total, dtype=torch.float32, device=dev)
        steps = self._steps
        if self._import_steps:          # load_state_dict: the loaded moments come with THEIR step count (torch.optim.Adam
            steps = 0                   # takes the loaded one too); one flat kernel = one count, the largest loaded
        for p, off in zip(f.params, f.offsets):
            st = self.state.get(p)
            if st and "exp_avg" in st:
                m.as_strided(p.shape, p.stride(), off).copy_(st["exp_avg"])
                v.as_strided(p.shape, p.stride(), off).copy_(st["exp_avg_sq"])
                if self._import_steps:
                    steps = max(steps, int(float(st.get("step", 0))))
        self._import_steps = False
        self._m, self._v, self._steps = m, v, steps
        self._dev_state = None
        self._state_views()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._ensure_state()
        f = self.flat
        if not f.grads_attached():
            raise RuntimeError("FusedAdam.step(): gradients are not in the network's flat gradient buffer "
                               "(backward of a dct_amd network attaches them)")
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        state = self._ensure_dev_state(f.flat.device, g["lr"])
        self._steps += 1                    # the kernel increments the device copy in stream order
        shadow = f.ensure_shadow() if f.want_shadow else None
        hip_ops.adam_flat_dev(f.flat, f.gflat, self._m, self._v, state, self._dev_table, b1, b2, g["eps"],
                              g["weight_decay"], bf16_shadow=shadow, grad_scale=self.grad_scale)
        if self._on_step is not None:
            try:
                self._on_step(shadow_fresh=shadow is not None)
            except TypeError:
                self._on_step()
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._m = None  # force re-import of the loaded per-parameter moments
        self._import_steps = True
        self._dev_state = None
        steps = [int(float(st.get("step", 0))) for st in self.state.values() if isinstance(st, dict)]
        self._steps = max(steps) if steps else 0

    def state_dict(self):
        if self._m is not None:
            self._state_views()         # refreshes the per-parameter "step" entries from the host count
        return super().state_dict()

"""Loss modules of the hot path with the reference's signatures
(/root/reference/generalframework/loss/loss.py) on top of the fused HIP kernels (K10).

``CrossEntropyLoss2d`` (:12-25), ``Entropy_2D`` (:70-84), ``KL_Divergence_2D`` (:110-134),
``JSD_2D`` (:183-196).  Inputs/outputs are logical NCHW torch tensors exactly as in the
reference; internally every module works on the physical NHWC fp32 image of its input (free
for tensors produced by dct_amd networks, which are channels_last already).

The reference asserts ``simplex(p)`` (a host-synchronising ``allclose``) on every call; here
that check runs only when ``dct_amd.loss.DEBUG_ASSERTS`` is True, so the stream is never
drained in production.  HIP only -- CPU tensors are rejected.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import hip_ops as K

DEBUG_ASSERTS = False


def _pc(t: torch.Tensor) -> torch.Tensor:
    """logical [B,C,H,W] -> physical NHWC fp32 dense (no copy when already so)."""
    if not t.is_cuda:
        raise RuntimeError("dct_amd losses run on the HIP device only (no CPU fallback)")
    p = t.permute(0, 2, 3, 1)
    if p.dtype != torch.float32 or not p.is_contiguous():
        p = p.to(torch.float32).contiguous()
    return p


def _nchw(p: torch.Tensor) -> torch.Tensor:
    return p.permute(0, 3, 1, 2)


def _simplex(t: torch.Tensor, axis=1) -> bool:
    s = t.sum(axis).type(torch.float32)
    return torch.allclose(s, torch.ones_like(s))


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        lp = _pc(logits)
        C = lp.shape[3]
        t = targets.reshape(-1)
        if t.dtype != torch.int64 or not t.is_contiguous():
            t = t.to(torch.int64).contiguous()
        out = K.ce_fwd(lp, t, C, ignore_index)
        ctx.save_for_backward(lp, t, out)
        ctx.ignore_index = ignore_index
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lp, t, out = ctx.saved_tensors
        dl = torch.empty_like(lp)
        g = g.to(torch.float32).contiguous()
        K.ce_bwd(lp, t, lp.shape[3], out[1:2], dl, gscale=g, ignore_index=ctx.ignore_index)
        return _nchw(dl), None, None


class CrossEntropyLoss2d(nn.Module):
    def __init__(self, weight=None, reduce=True, size_average=True, ignore_index=255):
        super().__init__()
        if weight is not None and any(float(w) != 1.0 for w in weight):
            raise NotImplementedError("dct_amd CrossEntropyLoss2d: class weights are not on the co-training path "
                                      "(train_ACDC_cotraining.py:48 uses weight=None)")
        if not (reduce and size_average):
            raise NotImplementedError("dct_amd CrossEntropyLoss2d: only the mean reduction of the reference path")
        self.weight = weight
        self.ignore_index = ignore_index

    def forward(self, outputs, targets):
        assert outputs.dim() == 4 and targets.dim() == 3, (outputs.shape, targets.shape)
        return _CEFn.apply(outputs, targets, self.ignore_index)


class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        lp = _pc(logits)
        probs = K.softmax_fwd(lp, lp.shape[3])
        ctx.save_for_backward(probs)
        return _nchw(probs)

    @staticmethod
    def backward(ctx, g):
        (probs,) = ctx.saved_tensors
        return _nchw(K.softmax_bwd(probs, _pc(g), probs.shape[3]))


def softmax_channels(logits: torch.Tensor) -> torch.Tensor:
    """F.softmax(logits, 1) of models/segmentators.py:50 on the HIP kernel."""
    return _SoftmaxFn.apply(logits)


class _EntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs):
        pp = _pc(probs)
        ctx.save_for_backward(pp)
        return K.entropy_fwd(pp, pp.shape[3]).view(pp.shape[:3])

    @staticmethod
    def backward(ctx, g):
        (pp,) = ctx.saved_tensors
        return _nchw(K.entropy_bwd(pp, g.to(torch.float32).contiguous().view(-1), pp.shape[3]))


class Entropy_2D(nn.Module):
    def forward(self, input: torch.Tensor):
        assert input.shape.__len__() == 4
        if DEBUG_ASSERTS:
            assert _simplex(input)
        return _EntropyFn.apply(input)


class _KLMapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, y, eps):
        pp, yp = _pc(p), _pc(y)
        ctx.save_for_backward(pp, yp)
        ctx.eps = eps
        return K.kl_map_fwd(pp, yp, pp.shape[3], eps).view(pp.shape[:3])

    @staticmethod
    def backward(ctx, g):
        pp, yp = ctx.saved_tensors
        dp = K.kl_map_bwd(pp, yp, g.to(torch.float32).contiguous().view(-1), pp.shape[3], ctx.eps)
        return _nchw(dp), None, None


class KL_Divergence_2D(nn.Module):
    """sum_c y*(log(y+eps) - log(p+eps)); the target ``y_prob`` is a constant on the co-training
    path (cotraining_totalloss.py:392 passes ``real_preds.detach()``), so only ``p_prob`` gets a gradient."""

    def __init__(self, reduce=False, eps=1e-10):
        super().__init__()
        self.reduce = reduce
        self.eps = eps

    def forward(self, p_prob: torch.Tensor, y_prob: torch.Tensor):
        if y_prob.requires_grad:
            raise NotImplementedError("dct_amd KL_Divergence_2D: pass y_prob.detach() (as the co-training step does)")
        if DEBUG_ASSERTS:
            assert _simplex(p_prob, 1) and _simplex(y_prob, 1)
        kl = _KLMapFn.apply(p_prob, y_prob, self.eps)
        return kl.mean() if self.reduce else kl


class _JSDMapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *probs):
        pps = [_pc(p) for p in probs]
        ctx.save_for_backward(*pps)
        return K.jsd_map_fwd(pps, pps[0].shape[3]).view(pps[0].shape[:3])

    @staticmethod
    def backward(ctx, g):
        pps = list(ctx.saved_tensors)
        dps = K.jsd_map_bwd(pps, g.to(torch.float32).contiguous().view(-1), pps[0].shape[3])
        return tuple(_nchw(d) for d in dps)


MAX_VIEWS = 8     # csrc/loss.hip MAXS


class JSD_2D(nn.Module):
    """H(mean_i p_i) - mean_i H(p_i) -> [B,H,W]; up to 8 views per call (the reference's sweeps run 2, 4 and 6:
    script/GM/run_multiview.sh:2-6, script/ACDC/5_run_multiple_view.sh:27-33)."""

    def __init__(self):
        super().__init__()
        self.entropy = Entropy_2D()

    def forward(self, input: List[torch.Tensor]):
        assert 1 <= len(input) <= MAX_VIEWS, f"dct_amd JSD_2D handles up to {MAX_VIEWS} models per call"
        if DEBUG_ASSERTS:
            for inprob in input:
                assert _simplex(inprob, 1)
        return _JSDMapFn.apply(*input)

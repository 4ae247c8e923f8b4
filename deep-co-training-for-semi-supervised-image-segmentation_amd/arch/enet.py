"""Enet of the reference (arch/enet.py:8-243) as one HIP execution plan.

Same parameters / buffers, ``state_dict`` keys and logical shapes as the reference module
(``encoder.initial.*``, ``encoder.bottleneck_{1_0..3_8}.{block1x1_1,middle_block,block1x1_2}.*``,
``decoder.layers.{0..5}.*``; ``enet.py:191-192`` registers the encoder blocks by name), but the
forward/backward are fixed sequences of the fused Enet kernels of include/dct.h:

  * a conv writes its RAW output once; its BatchNorm statistics come from one reduction pass and
    the normalise + PReLU/ReLU is applied by whoever reads it next ("normalise on load");
  * the bottleneck tail ``relu(main + ext)`` (enet.py:146-149) is one kernel that also does the
    max-pool-with-indices / zero channel pad (down) or max-unpool (up) of the main branch;
  * backward mirrors it: one BN/activation backward (reduce + apply) per conv, data gradient and
    weight gradient through the same generic small-channel kernels.

``Dropout2d`` is constructed by the reference (enet.py:122) but never applied in ``forward``, so
Enet is deterministic in train mode apart from the BatchNorm batch statistics -- three separate
statistics batches per model per step (labeled, unlabeled, adversarial), which is why the trainer
never merges passes for this network (``batch_independent = False``).

Data layout in HBM: activations NHWC in ``compute_dtype`` (bf16 default / fp32 parity mode); the
pooling argmax codes are uint8 [N,h,w,C]; weights are the fp32 masters of the flat parameter
buffer, physically K-major ``[Cout][kh][kw][Cin]`` (``[Cin][kh][kw][Cout]`` for transposed convs).
No CPU path.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import hip_ops as K
from ..hip_ops import Tf
from .flat import FlatParams
from .unet import _ConvP, _Slots

_STAGE23 = [("regular", 1), ("dilated", 2), ("asym", 1), ("dilated", 4),
            ("regular", 1), ("dilated", 8), ("asym", 1), ("dilated", 16)]


class _EConv(_ConvP):
    """nn.Conv2d(cin, cout, (kh, kw), stride, padding, dilation, bias) parameters, K-major storage."""
    transposed = False

    def __init__(self, cin, cout, kh, kw, stride=1, pad=(0, 0), dil=1, bias=True):
        nn.Module.__init__(self)
        self.cin, self.cout, self.kh, self.kw = cin, cout, kh, kw
        self.stride, self.pad, self.dil = stride, pad, dil
        self.weight = nn.Parameter(torch.empty(cout, kh, kw, cin).permute(0, 3, 1, 2))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout))
        else:
            self.register_parameter("bias", None)
        fan_in = cin * kh * kw
        bound = 1.0 / fan_in ** 0.5
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if bias:
                self.bias.uniform_(-bound, bound)

    @property
    def taps(self):
        return self.kh * self.kw


class _EConvT(_EConv):
    """nn.ConvTranspose2d(cin, cout, k, stride=2, padding, output_padding): logical [cin,cout,k,k],
    physical [cin][k][k][cout]."""
    transposed = True

    def __init__(self, cin, cout, k, stride=2, pad=0, out_pad=0, bias=True):
        nn.Module.__init__(self)
        self.cin, self.cout, self.kh, self.kw = cin, cout, k, k
        self.stride, self.pad, self.dil, self.out_pad = stride, (pad, pad), 1, out_pad
        self.weight = nn.Parameter(torch.empty(cin, k, k, cout).permute(0, 3, 1, 2))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout))
        else:
            self.register_parameter("bias", None)
        bound = 1.0 / (cout * k * k) ** 0.5
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if bias:
                self.bias.uniform_(-bound, bound)


class _BN(nn.Module):
    """nn.BatchNorm2d(c, eps=1e-3) parameters and buffers (enet.py:22,55,...)."""
    is_dct_batchnorm = True

    def __init__(self, c, eps=1e-3, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _PReLU(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.full((c,), 0.25))


class _ReLU(nn.Module):
    pass


def _act(c, relu):
    return _ReLU() if relu else _PReLU(c)


class _Bottleneck(nn.Module):
    def __init__(self, cin, cout, kind="regular", dilation=1, relu=False):
        super().__init__()
        self.kind, self.cin, self.cout = kind, cin, cout
        mid = cout // 4
        k = 2 if kind == "down" else 1
        self.block1x1_1 = _Slots({0: _EConv(cin, mid, k, k, stride=k, bias=False), 1: _BN(mid), 2: _act(mid, relu)})
        if kind == "up":
            self.conv_before_unpool = _Slots({0: _EConv(cin, cout, 1, 1, bias=False), 1: _BN(cout)})
            core = _EConvT(mid, mid, 3, stride=2, pad=1, out_pad=1)
        elif kind == "dilated":
            core = _EConv(mid, mid, 3, 3, pad=(dilation, dilation), dil=dilation)
        elif kind == "asym":
            core = _Slots({0: _EConv(mid, mid, 5, 1, pad=(2, 0), bias=False), 1: _EConv(mid, mid, 1, 5, pad=(0, 2))})
        else:
            core = _EConv(mid, mid, 3, 3, pad=(1, 1))
        self.middle_block = _Slots({0: core, 1: _BN(mid), 2: _act(mid, relu)})
        self.block1x1_2 = _Slots({0: _EConv(mid, cout, 1, 1, bias=False), 1: _BN(cout), 2: _act(cout, relu)})


class _Initial(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = _EConv(1, 13, 3, 3, stride=2, pad=(1, 1))
        self.batch_norm = _BN(13)
        self.prelu = _PReLU(13)


class _Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.order: List[str] = []

        def add(name, mod):
            self.add_module(name, mod)
            self.order.append(name)

        add("initial", _Initial())
        add("bottleneck_1_0", _Bottleneck(14, 64, "down"))
        for i in range(1, 5):
            add(f"bottleneck_1_{i}", _Bottleneck(64, 64))
        add("bottleneck_2_0", _Bottleneck(64, 128, "down"))
        for stage in (2, 3):
            for i, (kind, dil) in enumerate(_STAGE23, start=1):
                add(f"bottleneck_{stage}_{i}", _Bottleneck(128, 128, kind, dil))


class _Decoder(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.layers = nn.ModuleList([
            _Bottleneck(128, 64, "up", relu=True), _Bottleneck(64, 64, relu=True), _Bottleneck(64, 64, relu=True),
            _Bottleneck(64, 14, "up", relu=True), _Bottleneck(14, 14, relu=True),
            _EConvT(14, num_classes, 2, stride=2)])


def _empty(*a, **k):
    """Allocation of a plan: referenced until the launches are issued while the pass records into a K.PassGroup."""
    return K.keep(torch.empty(*a, **k))


def _empty_like(t):
    return K.keep(torch.empty_like(t))


class _Draw(object):
    """The BatchNorm-backward result of a layer (gradient wrt its raw conv output), possibly not materialised: the data-gradient
    convolution it feeds computes it on load (K.enet_conv_bwd_in), and only a weight gradient needs the tensor -- as a leaf."""
    __slots__ = ("rec", "g", "g_mask", "scratch", "tensor")

    def __init__(self, rec, g, g_mask, scratch, tensor=None):
        self.rec, self.g, self.g_mask, self.scratch, self.tensor = rec, g, g_mask, scratch, tensor


class _Rec(object):
    """what one conv+BN(+act) unit leaves behind for its consumers and for the backward"""
    __slots__ = ("raw", "tf", "mean", "invstd", "bn", "act", "conv", "src", "src_tf")


class Enet(nn.Module):
    """Drop-in for the reference ``Enet(num_classes)`` (enet.py:234-243)."""
    batch_independent = False     # BatchNorm batch statistics couple the samples of a pass

    def __init__(self, num_classes: int = 2, compute_dtype=torch.bfloat16):
        super().__init__()
        if not 2 <= num_classes <= 8:
            raise ValueError("dct_amd Enet: 2 <= num_classes <= 8")
        self.num_classes = num_classes
        self.compute_dtype = compute_dtype
        self.encoder = _Encoder()
        self.decoder = _Decoder(num_classes)
        self.flat_params = FlatParams(list(self.parameters()))
        self._pidx = {id(p): i for i, p in enumerate(self.flat_params.params)}
        self._grad_target = None
        # Batch statistics of a training-mode forward pass: ONE flat buffer per pass, layer k's five vectors (scale, shift, mean,
        # invstd, unbiased variance; c floats each) at element 5 * _bn_off[k] -- so that the running-statistics bookkeeping of all
        # 84 layers is one launch over a static record table (dct_bn_running_update) instead of four torch._foreach launches
        self._bn_list = [m for m in self.modules() if isinstance(m, _BN)]
        self._bn_off, off = {}, 0
        for m in self._bn_list:
            self._bn_off[id(m)] = off
            off += (m.num_features + 3) // 4 * 4            # (16-byte aligned vectors)
        self._bn_total = off
        self._bn_stats = None                # flat statistics buffer of the forward pass being planned
        self._bn_table = None                # (key, device record table)
        self._defer_running = False
        self.fuse_bn_stats = os.environ.get("DCT_ENET_FUSE_BN_STATS", "1") != "0"     # BatchNorm partial sums from the conv epilogue
        # ... and the backward sums from the data-gradient epilogue (dct_enet_conv_bnbwd_stats): -270 launches per cfg4 step.  Level
        # on the four-queue step of round 2 (17.05 vs 17.15 ms), -3..5 % with the grouped layout of round 3 (15.2-15.6 vs 16.0-16.2 ms
        # on one box; cfg5 level) -- on; tests/test_enet_kernels_gpu.py pins the kernel, tools/debug_fuse_bn.py compares a
        # whole backward pass (3e-3 in fp16 / 3e-2 in bf16 on the smallest gradients: rounding ties of the 16-bit draw tensors)
        # (With the library still containing packed-FP32 instructions the fp16 ACDC run of tests/test_acdc_dsc_gpu.py ended at twice the
        # reference's loss with this on -- 0.144 vs 0.070 after 150 steps -- and at 0.07 without: the epilogue's sums were hit by the
        # hazard of DESIGN 4.3.  Built without them, it passes in both modes.)
        self.fuse_bn_bwd_stats = os.environ.get("DCT_ENET_FUSE_BN_BWD", "1") == "1"
        # BatchNorm-backward apply computed ON LOAD by the data-gradient convolution it feeds (K.enet_conv_bwd_in), the tensor only
        # materialised -- as a leaf -- for the weight gradient: bit-identical, 60-90 launches off the adversarial chain, and LEVEL on
        # cfg4 (15.98-16.18 vs 15.91-16.03 ms on one box), +0.5 % on cfg5: the convolution pays in loads (raw fp32 + g + mask instead
        # of one 16-bit tensor) what the chain saves in launches.  Off; kept for A/B.
        self.denorm_on_load = os.environ.get("DCT_ENET_DENORM_ON_LOAD", "0") == "1"
        self.denorm_on_load_all = False      # (A/B: also for 3x3 / dilated / asymmetric kernels)
        # most 32-pixel tiles (= partial rows the one-block finalize must fold) a fused layer may have.  1024 / 1280 / 2560 / 5120 on one
        # box: cfg4 15.79 / 15.52 / 15.43 / 15.54 ms, cfg5 37.50 / - / 37.48 / 38.82 (profiles/r03_cfg4_grouped_passes.txt)
        self.stats_tiles_cap = int(os.environ.get("DCT_ENET_STATS_TILES", "2560"))
        self.skip_zero_bias_grads = os.environ.get("DCT_ENET_BIAS_GRADS", "0") != "1"    # see _conv_wgrad

    supports_pass_streams = True         # plan_backward(grad_buffer=...): concurrent backward passes of one model (trainer)
    prefers_segmented_graphs = True      # ~780 launches of ~8 us per pass: the step is captured as one HIP graph per stream segment
                                         # so that the models / passes run on different hardware queues (trainer/stream_sched.py)

    @property
    def _nbt(self):
        return [m.num_batches_tracked for m in self.modules() if isinstance(m, _BN)]

    def mark_weights_updated(self):
        pass    # no packed copies: the kernels read the fp32 masters

    # ------------------------------------------------------------------------------ helpers
    def _w(self, p):
        return self.flat_params.dense(self._pidx[id(p)])

    def _g(self, p):
        i = self._pidx[id(p)]
        if self._grad_target is not None:       # a backward pass on its own stream accumulates into its own flat buffer
            off = self.flat_params.offsets[i]
            return self._grad_target[off:off + self.flat_params.params[i].numel()]
        return self.flat_params.grad_dense(i)

    def _check_input(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("dct_amd Enet runs on the HIP device only (no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"expected [B,1,H,W], got {tuple(x.shape)}")
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise RuntimeError("Enet needs H and W divisible by 8 (three stride-2 stages, enet.py:26,132)")

    # Autograd-free entry points of the execution plan (see arch/unet.py::plan_forward)
    def plan_forward(self, x: torch.Tensor, save: bool = True, defer_running: bool = False):
        """``defer_running`` (training mode, ``save``): the running statistics are NOT updated; the batch statistics of the 84
        BatchNorms travel in the tape and ``apply_running_updates`` folds them in later.  The forward passes a model sees per
        step (labeled, unlabeled, FGSM, adversarial) then have no order among themselves and can run on different streams; the
        caller applies their updates in the reference's order."""
        self._check_input(x)
        self.flat_params.ensure()
        self._defer_running = bool(defer_running and save and self.training)
        try:
            return self._run_forward(x, save)
        finally:
            self._defer_running = False

    supports_deferred_running_stats = True
    supports_pass_groups = True          # the plan's launches record into a hip_ops.PassGroup (grouped passes)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._bn_table = None                # the buffers may have moved
        return r

    def _apply_running(self, stats):
        """r <- (1 - momentum) r + momentum b for every BatchNorm of one forward pass and num_batches_tracked += 1, in one launch
        (csrc/bn.hip::bn_running_update_kernel).  Every training-mode forward goes through here (immediately, or deferred), so
        the arithmetic does not depend on the schedule.  ``stats``: the pass's flat statistics buffer."""
        if stats is None:
            return
        bns = self._bn_list
        key = tuple(b.running_mean.data_ptr() for b in bns[:2]) + (bns[-1].running_var.data_ptr(), stats.device)
        if self._bn_table is None or self._bn_table[0] != key:
            mom = bns[0].momentum
            assert all(bn.momentum == mom for bn in bns)
            recs = []
            for bn in bns:
                c, base = bn.num_features, 5 * self._bn_off[id(bn)]
                cp = (c + 3) // 4 * 4
                recs.append((bn.running_mean, bn.running_var, bn.num_batches_tracked, c, base + 2 * cp, base + 4 * cp))
            self._bn_table = (key, K.bn_running_table(recs, stats.device), mom)
        K.bn_running_update(self._bn_table[1], len(bns), stats, self._bn_table[2])

    def apply_running_updates(self, tapes):
        """The deferred running-statistics updates of ``tapes`` (plan_forward(defer_running=True)), in the order given."""
        for tape in tapes:
            stats = tape[-1].pop("bn_stats", None)
            if stats is not None:
                self._apply_running(stats)

    def plan_backward(self, tape, dlogits: torch.Tensor, need_dx: bool = False, need_dw: bool = True, grad_buffer=None,
                      leaf_hook=None, leaf_every: int = 7):
        """``grad_buffer``: flat fp32 tensor (flat_params.total elements, zeroed by the caller) that receives this pass's
        parameter gradients instead of the network's gradient buffer -- the 2-3 backward passes a model sees per step are
        independent apart from that accumulation, so the trainer runs them on separate streams and adds the buffers in a fixed
        order afterwards."""
        if need_dw:
            self.flat_params.ensure_grads()
        self._grad_target = grad_buffer if need_dw else None
        self._leaf_hook = (leaf_hook, max(1, int(leaf_every))) if (leaf_hook is not None and need_dw) else None
        try:
            dx = self._run_backward(tape, dlogits, need_dx, need_dw)
        finally:
            self._grad_target = None
            self._leaf_hook = None
        return dx.reshape(dx.shape[0], 1, dx.shape[1], dx.shape[2]) if dx is not None else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input(x)
        self.flat_params.ensure()
        params = self.flat_params.params
        save = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _EnetFn.apply(self, save, x, *params)

    # ------------------------------------------------------------------------------ forward plan
    def _cba(self, src, src_tf, conv, bn, act, out_hw, save) -> _Rec:
        """conv (+bias) -> raw; BatchNorm statistics -> consumer transform."""
        dt, dev = self.compute_dtype, src.device
        B = src.shape[0]
        raw = _empty(B, out_hw[0], out_hw[1], conv.cout, dtype=torch.float32, device=dev)   # raw: always fp32
        stats, rows = None, 0
        tiles = (B * out_hw[0] * out_hw[1] + 31) // 32
        if (self.training and self.fuse_bn_stats and dt != torch.float32 and conv.cin >= 16 and conv.cin % 16 == 0 and
                conv.cout <= 128 and tiles <= self.stats_tiles_cap):
            # MFMA convolution: its epilogue writes the BatchNorm partial sums (one launch and one read of `raw` less per seam)
            stats = _empty(tiles * conv.cout * 3, dtype=torch.float64, device=dev)
            fused = True
        else:
            fused = False
        rec = _Rec()
        rec.raw, rec.conv, rec.bn, rec.act, rec.src, rec.src_tf = raw, conv, bn, act, src, src_tf
        c = conv.cout
        if self.training:
            # batch statistics only: the running statistics are updated by _apply_running from (mean, unbiased variance)
            if self._bn_stats is None:
                self._bn_stats = _empty(5 * self._bn_total, dtype=torch.float32, device=dev)
            cp, base = (c + 3) // 4 * 4, 5 * self._bn_off[id(bn)]
            vec = self._bn_stats[base:base + 5 * cp].view(5, cp)[:, :c]
            if fused:       # the convolution's epilogue leaves the BatchNorm's partial sums where the library can (-> rows)
                rows = self._conv_fwd(src, src_tf, conv, raw, stats=stats)
            else:
                self._conv_fwd(src, src_tf, conv, raw)
            K.enet_bn_fwd_stats(raw, self._w(bn.weight), self._w(bn.bias), bn.eps, bn.momentum, None, None,
                                True, vec[0], vec[1], vec[2], vec[3], save_var=vec[4], partial=stats, partial_rows=rows)
        else:
            self._conv_fwd(src, src_tf, conv, raw)
            vec = _empty(4, c, dtype=torch.float32, device=dev)
            K.enet_bn_fwd_stats(raw, self._w(bn.weight), self._w(bn.bias), bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                False, vec[0], vec[1], vec[2], vec[3])
        if isinstance(act, _PReLU):
            rec.tf = Tf(vec[0], vec[1], self._w(act.weight), Tf.PRELU)
        elif isinstance(act, _ReLU):
            rec.tf = Tf(vec[0], vec[1], None, Tf.RELU)
        else:
            rec.tf = Tf(vec[0], vec[1], None, Tf.AFFINE)
        rec.mean, rec.invstd = vec[2], vec[3]
        return rec

    def _conv_fwd(self, src, src_tf, conv, dst, stats=None):
        """``stats`` (float64 scratch): the convolution's epilogue also writes the BatchNorm partial sums of ``dst`` where it can;
        -> number of partial rows (0: not written)."""
        w = self._w(conv.weight)
        b = self._w(conv.bias) if conv.bias is not None else None
        t = conv.taps
        if stats is not None:
            kw = dict(R=conv.kh, S=conv.kw, stride=conv.stride, pad_h=conv.pad[0], pad_w=conv.pad[1], compute=self.compute_dtype)
            if conv.transposed:
                return K.enet_conv_stats(src, w, b, src_tf, dst, stats, transposed=True, ws=(1, conv.cout, t * conv.cout), **kw)
            return K.enet_conv_stats(src, w, b, src_tf, dst, stats, dil=conv.dil, ws=(t * conv.cin, conv.cin, 1), **kw)
        if conv.transposed:
            K.enet_conv(src, w, b, src_tf, dst, R=conv.kh, S=conv.kw, stride=conv.stride, pad_h=conv.pad[0], pad_w=conv.pad[1],
                        transposed=True, ws=(1, conv.cout, t * conv.cout), compute=self.compute_dtype)
        else:
            K.enet_conv(src, w, b, src_tf, dst, R=conv.kh, S=conv.kw, stride=conv.stride, dil=conv.dil, pad_h=conv.pad[0],
                        pad_w=conv.pad[1], ws=(t * conv.cin, conv.cin, 1), compute=self.compute_dtype)

    def _conv_dgrad(self, g, conv, dst, accumulate=False, resid=None, bn_of_dst=None, need_dw=True):
        """dst (+)= d(loss)/d(conv input) given g = d/d(conv output) -- a tensor, or a _Draw that the convolution computes on load
        where the library has that form (else it is materialised here, on the chain).

        ``bn_of_dst`` (a _Rec): dst is the gradient wrt act(BN(rec.raw)); where the MFMA form runs (and fuse_bn_bwd_stats is on), its
        epilogue also writes that BatchNorm's backward partial sums -> (dst, (stats, scratch) | None, rows) for _bn_bwd (rows = 0: not
        written)."""
        w = self._w(conv.weight)
        t = conv.taps
        rg, rm = resid if resid is not None else (None, None)
        kw = dict(R=conv.kh, S=conv.kw, stride=conv.stride, pad_h=conv.pad[0], pad_w=conv.pad[1], compute=self.compute_dtype)
        if conv.transposed:
            kw.update(ws=(t * conv.cout, conv.cout, 1))
        else:
            kw.update(dil=conv.dil, transposed=True, ws=(1, conv.cin, t * conv.cin))
        cin_g = g.rec.raw.shape[3] if isinstance(g, _Draw) else g.shape[3]
        stats = scratch = None
        if bn_of_dst is not None:
            tiles = (dst.shape[0] * dst.shape[1] * dst.shape[2] + 31) // 32
            if (self.fuse_bn_bwd_stats and self._tape_training and self.compute_dtype != torch.float32 and cin_g >= 16 and cin_g % 16 == 0 and
                    dst.shape[3] <= 128 and tiles <= self.stats_tiles_cap and not accumulate and resid is None):
                stats = _empty(tiles * dst.shape[3] * 3, dtype=torch.float64, device=dst.device)
                scratch = _empty(2 * dst.shape[3], dtype=torch.float32, device=dst.device)
        rec = bn_of_dst
        # on load only where every input element is read ONCE (1x1, or a kernel as large as its stride): a 3x3 would repeat the
        # transform -- and its three loads -- nine times per element (measured: the chain got 92 launches shorter and 0.5 ms slower)
        once = conv.kh * conv.kw == 1 or (conv.kh == conv.stride and conv.kw == conv.stride and conv.dil == 1)
        if isinstance(g, _Draw) and g.tensor is None and (once or self.denorm_on_load_all):
            d = g
            rows = K.enet_conv_bwd_in(d.rec.raw, w, d.rec.tf, d.g, d.g_mask, d.rec.mean, d.rec.invstd, d.scratch, dst,
                                      resid_grad=rg, resid_mask=rm, accumulate=accumulate,
                                      bn=(stats, rec.raw, rec.tf, rec.mean, rec.invstd) if stats is not None else None, **kw)
            if rows is not None:
                return (dst, (stats, scratch) if stats is not None else None, rows) if bn_of_dst is not None else dst
        gt = self._tensor_of(g, leaf=False)
        if stats is not None:
            rows = K.enet_conv_bnbwd_stats(gt, w, dst, stats, rec.raw, rec.tf, rec.mean, rec.invstd, **kw)
            return dst, (stats, scratch), rows
        K.enet_conv(gt, w, None, None, dst, accumulate=accumulate, resid_grad=rg, resid_mask=rm, **kw)
        return (dst, None, 0) if bn_of_dst is not None else dst

    def _conv_wgrad(self, g, conv, src, src_tf, before_bn=False):
        """dW (+ db) += for a conv whose input was src (read through src_tf) and output gradient is g.

        ``before_bn``: the conv feeds a train-mode BatchNorm.  Its bias then has an identically zero gradient (the batch mean
        absorbs it: sum over pixels of the BatchNorm input gradient is 0), which the reference evaluates as fp32 rounding noise
        ~1e-8 under a weight-decay term ~1e-5 (chaotic parity points; the step tests bound these biases by n * lr).
        It is taken as the exact zero here -- nothing to accumulate -- which saves one reduction + one fold launch per conv
        and pass (~860 of a cfg4 step's ~5600 launches).  Eval-mode BatchNorm (running statistics) does pass a bias gradient."""
        dw = self._g(conv.weight)
        g = self._tensor_of(g, leaf=True)          # (a _Draw not needed by its data-gradient convolution is materialised as a leaf)
        if conv.transposed:
            K.enet_wgrad(src, src_tf, g, None, dw, R=conv.kh, S=conv.kw, stride=conv.stride, pad_h=conv.pad[0], pad_w=conv.pad[1])
        else:
            K.enet_wgrad(g, None, src, src_tf, dw, R=conv.kh, S=conv.kw, stride=conv.stride, dil=conv.dil, pad_h=conv.pad[0],
                         pad_w=conv.pad[1])
        if conv.bias is not None and not (before_bn and self.skip_zero_bias_grads and self._tape_training):
            K.enet_channel_sum(g, self._g(conv.bias))

    def _bottleneck_fwd(self, blk: _Bottleneck, x, idx_in, save):
        dt, dev = self.compute_dtype, x.device
        B, h, w, _ = x.shape
        st: Dict[str, object] = {"blk": blk, "x": x}
        if blk.kind == "down":
            oh, ow = h // 2, w // 2
        elif blk.kind == "up":
            oh, ow = 2 * h, 2 * w
        else:
            oh, ow = h, w
        s1 = blk.block1x1_1
        r1 = self._cba(x, None, s1.at(0), s1.at(1), s1.at(2), (oh, ow) if blk.kind == "down" else (h, w), save)
        mb = blk.middle_block
        core = mb.at(0)
        if blk.kind == "asym":
            c5, c15 = core.at(0), core.at(1)
            mid_raw = _empty(B, oh, ow, c5.cout, dtype=torch.float32, device=dev)
            self._conv_fwd(r1.raw, r1.tf, c5, mid_raw)
            r2 = self._cba(mid_raw, None, c15, mb.at(1), mb.at(2), (oh, ow), save)
            st["mid_raw"] = mid_raw
        else:
            r2 = self._cba(r1.raw, r1.tf, core, mb.at(1), mb.at(2), (oh, ow), save)
        s3 = blk.block1x1_2
        r3 = self._cba(r2.raw, r2.tf, s3.at(0), s3.at(1), s3.at(2), (oh, ow), save)
        out = _empty(B, oh, ow, blk.cout, dtype=dt, device=dev)
        idx_out = None
        if blk.kind == "down":
            idx_out = _empty(B, oh, ow, blk.cin, dtype=torch.uint8, device=dev)
            K.enet_tail_fwd(r3.raw, r3.tf, x, None, None, idx_out, blk.cin, 1, out)
        elif blk.kind == "up":
            cb = blk.conv_before_unpool
            rm = self._cba(x, None, cb.at(0), cb.at(1), None, (h, w), save)
            K.enet_tail_fwd(r3.raw, r3.tf, None, rm.raw, rm.tf, idx_in, blk.cout, 2, out)
            st["rm"], st["idx"] = rm, idx_in
        else:
            K.enet_tail_fwd(r3.raw, r3.tf, x, None, None, None, 0, 0, out)
        if idx_out is not None:
            st["idx"] = idx_out
        st.update(r1=r1, r2=r2, r3=r3, out=out)
        return out, idx_out, (st if save else None)

    def _run_forward(self, x, save):
        dt, dev = self.compute_dtype, x.device
        B, _, H, W = x.shape
        xs = x.detach().to(torch.float32).reshape(B, H, W, 1)
        if not xs.is_contiguous():
            xs = xs.contiguous()
        ini = self.encoder.initial
        r0 = self._cba(xs, None, ini.conv, ini.batch_norm, ini.prelu, (H // 2, W // 2), save)   # the image stays fp32
        h = _empty(B, H // 2, W // 2, 14, dtype=dt, device=dev)
        K.enet_tail_fwd(r0.raw, r0.tf, xs, None, None, None, 13, 3, h)
        tape = [{"kind": "initial", "x": xs, "r0": r0, "out": h}] if save else None
        stack = []
        for name in self.encoder.order[1:]:
            blk = getattr(self.encoder, name)
            h, idx, st = self._bottleneck_fwd(blk, h, None, save)
            if idx is not None:
                stack.append(idx)
            if save:
                tape.append(st)
        for blk in list(self.decoder.layers)[:5]:
            idx = stack.pop() if blk.kind == "up" else None
            h, _, st = self._bottleneck_fwd(blk, h, idx, save)
            if save:
                tape.append(st)
        fin = self.decoder.layers[5]
        logits = _empty(B, H, W, self.num_classes, dtype=torch.float32, device=dev)
        self._conv_fwd(h, None, fin, logits)
        stats, self._bn_stats = self._bn_stats, None
        if save:
            tape.append({"kind": "final", "x": h, "training": self.training})
        if self.training:
            if getattr(self, "_defer_running", False):
                tape[-1]["bn_stats"] = stats
            else:
                self._apply_running(stats)
        return logits, tape

    # ------------------------------------------------------------------------------ backward plan
    def _bn_bwd(self, rec: _Rec, g, g_mask, need_dw, partial=None, rows=0):
        """grad wrt act(bn(raw)) -> grad wrt raw; accumulates dgamma / dbeta / dslope.  ``partial`` / ``rows``: the reduction's
        partial sums, already written by the data-gradient convolution that produced g (_conv_dgrad(bn_of_dst=rec))."""
        dev = rec.raw.device
        c = rec.raw.shape[3]
        if partial is not None:          # (_conv_dgrad: the partial rows and the scratch for the two means)
            partial, scratch = partial
        else:
            scratch = _empty(2 * c, dtype=torch.float32, device=dev)
        dg = self._g(rec.bn.weight) if need_dw else None
        db = self._g(rec.bn.bias) if need_dw else None
        ds = self._g(rec.act.weight) if (need_dw and isinstance(rec.act, _PReLU)) else None
        # FGSM pass (need_dw False): the finalize kernel skips null parameter gradients -- no throw-away zero buffers
        if self.denorm_on_load and self.compute_dtype != torch.float32:
            # sums only: the elementwise apply leaves the chain -- the data-gradient convolution computes the result on load
            # (_conv_dgrad), a weight gradient materialises it as a leaf (_tensor_of)
            K.enet_bn_bwd_sums(rec.raw, g, g_mask, rec.tf, rec.mean, rec.invstd, dg, db, ds, scratch, training=self._tape_training,
                               partial=partial, partial_rows=rows)
            return _Draw(rec, g, g_mask, scratch)
        draw = _empty(rec.raw.shape, dtype=self.compute_dtype, device=dev)
        K.enet_bn_bwd(rec.raw, g, g_mask, rec.tf, rec.mean, rec.invstd, dg, db, ds, scratch, draw, training=self._tape_training,
                      partial=partial, partial_rows=rows)
        return _Draw(rec, g, g_mask, scratch, draw)

    def _tensor_of(self, d, leaf: bool):
        """The gradient tensor behind ``d`` (a plain tensor, or a _Draw that is materialised now if it was not)."""
        if not isinstance(d, _Draw):
            return d
        if d.tensor is None:
            rec = d.rec
            d.tensor = K.enet_bn_bwd_apply(rec.raw, d.g, d.g_mask, rec.tf, rec.mean, rec.invstd, d.scratch,
                                           _empty(rec.raw.shape, dtype=self.compute_dtype, device=rec.raw.device), leaf=leaf)
        return d.tensor

    def _bottleneck_bwd(self, st, dout, need_dw, need_dx=True):
        blk: _Bottleneck = st["blk"]
        x, out = st["x"], st["out"]
        r1, r2, r3 = st["r1"], st["r2"], st["r3"]
        dt = self.compute_dtype
        # ---- extension branch, last to first
        # (every data gradient is issued BEFORE the weight gradient of the same layer: it computes the BatchNorm-backward result on
        #  load, the weight gradient materialises it -- a leaf, like itself)
        d3 = self._bn_bwd(r3, dout, out, need_dw)
        g2, p2, n2 = self._conv_dgrad(d3, r3.conv, _empty(r2.raw.shape, dtype=dt, device=r2.raw.device), bn_of_dst=r2, need_dw=need_dw)
        if need_dw:
            self._conv_wgrad(d3, r3.conv, r2.raw, r2.tf, before_bn=True)
        d2 = self._bn_bwd(r2, g2, None, need_dw, partial=p2, rows=n2)
        if blk.kind == "asym":
            c5, c15 = blk.middle_block.at(0).at(0), blk.middle_block.at(0).at(1)
            mid_raw = st["mid_raw"]
            gmid = self._conv_dgrad(d2, c15, _empty(mid_raw.shape, dtype=dt, device=mid_raw.device))
            if need_dw:
                self._conv_wgrad(d2, c15, mid_raw, None, before_bn=True)
                self._conv_wgrad(gmid, c5, r1.raw, r1.tf)
            g1, p1, n1 = self._conv_dgrad(gmid, c5, _empty(r1.raw.shape, dtype=dt, device=r1.raw.device), bn_of_dst=r1, need_dw=need_dw)
        else:
            g1, p1, n1 = self._conv_dgrad(d2, r2.conv, _empty(r1.raw.shape, dtype=dt, device=r1.raw.device), bn_of_dst=r1, need_dw=need_dw)
            if need_dw:
                self._conv_wgrad(d2, r2.conv, r1.raw, r1.tf, before_bn=True)
        d1 = self._bn_bwd(r1, g1, None, need_dw, partial=p1, rows=n1)
        # ---- input gradient = extension branch + main branch
        dx = _empty_like(x)
        if blk.kind == "down":
            K.enet_tail_bwd(dout, out, st["idx"], blk.cin, 1, dx)
            self._conv_dgrad(d1, r1.conv, dx, accumulate=True)
        elif blk.kind == "up":
            rm: _Rec = st["rm"]
            gm = K.enet_tail_bwd(dout, out, st["idx"], blk.cout, 2, _empty(rm.raw.shape, dtype=dt, device=rm.raw.device))
            dm = self._bn_bwd(rm, gm, None, need_dw)
            self._conv_dgrad(dm, rm.conv, dx)
            if need_dw:
                self._conv_wgrad(dm, rm.conv, x, None, before_bn=True)
            self._conv_dgrad(d1, r1.conv, dx, accumulate=True)
        else:
            self._conv_dgrad(d1, r1.conv, dx, resid=(dout, out))
        if need_dw:
            self._conv_wgrad(d1, r1.conv, x, None, before_bn=True)
        return dx

    def _run_backward(self, tape, dlogits, need_dx, need_dw):
        dt, dev = self.compute_dtype, dlogits.device
        fin = self.decoder.layers[5]
        st = tape[-1]
        self._tape_training = st["training"]
        dl = dlogits if dlogits.dtype == dt else K.cast(dlogits, _empty(dlogits.shape, dtype=dt, device=dev))
        if need_dw:
            self._conv_wgrad(dl, fin, st["x"], None)
        g = _empty_like(st["x"])
        self._conv_dgrad(dl, fin, g)
        hook = getattr(self, "_leaf_hook", None)
        for k, st in enumerate(reversed(tape[1:-1])):
            g = self._bottleneck_bwd(st, g, need_dw)
            if hook is not None and (k + 1) % hook[1] == 0:
                hook[0]()           # (K.LeafSide: the weight gradients held back so far go out on another queue)
        st = tape[0]
        r0 = st["r0"]
        d0 = self._bn_bwd(r0, g[..., :13], None, need_dw)
        ini = self.encoder.initial
        dx = None
        if need_dx:
            dx = _empty_like(st["x"])
            self._conv_dgrad(d0, ini.conv, dx)
            K.enet_tail_bwd(g, st["x"], None, 13, 3, dx, accumulate=True)
        if need_dw:
            self._conv_wgrad(d0, ini.conv, st["x"], None, before_bn=True)
        return dx


class _EnetFn(torch.autograd.Function):
    """One autograd node for the whole network (see arch/unet.py::_UNetFn)."""

    @staticmethod
    def forward(ctx, net: Enet, save: bool, x: torch.Tensor, *params):
        need_dw = any(p.requires_grad for p in params)
        logits, tape = net._run_forward(x, save)
        ctx.net, ctx.tape, ctx.need_dw = net, tape, need_dw
        ctx.set_materialize_grads(False)
        return logits.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        net: Enet = ctx.net
        n_params = len(net.flat_params.params)
        if g is None or ctx.tape is None:
            return (None, None, None) + (None,) * n_params
        dl = g.permute(0, 2, 3, 1)
        if dl.dtype != torch.float32 or not dl.is_contiguous():
            dl = dl.to(torch.float32).contiguous()
        need_dx = ctx.needs_input_grad[2]
        need_dw = ctx.need_dw and any(ctx.needs_input_grad[3:])
        if need_dw:
            net.flat_params.ensure_grads()
        dx = net._run_backward(ctx.tape, dl, need_dx, need_dw)
        ctx.tape = None
        gx = dx.reshape(dx.shape[0], 1, dx.shape[1], dx.shape[2]) if dx is not None else None
        return (None, None, gx) + (None,) * n_params

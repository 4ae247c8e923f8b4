"""UNet of the reference (arch/network.py:115-130,153-171,196-240) as one HIP execution plan.

Same parameters, same ``state_dict`` keys and logical shapes as the reference module, but
the forward/backward are not a graph of ATen ops: ``forward`` runs a fixed sequence of
hand-written gfx950 kernels over NHWC activations (include/dct.h), and a single
``torch.autograd.Function`` node replays the matching backward sequence.

Data layout in HBM
  activations : NHWC, bf16 (default) or fp32 (parity mode); skip-concats are channel slices of
                one buffer (the producer writes its slice, nothing is copied);
  weights     : fp32 masters in one flat buffer, physically K-major ``[Cout][kh][kw][Cin]``
                (``[Cin][a][b][Cout]`` for ConvTranspose2d) exposed as strided views with the
                reference's logical shapes; per-step packs in the compute dtype for the forward
                and the data-gradient GEMMs;
  gradients   : weight/bias gradients are accumulated straight into the flat fp32 gradient
                buffer that ``p.grad`` views alias (see arch/flat.py).

No CPU path: inputs must be on the HIP device.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import hip_ops as K
from .flat import FlatParams

_WIDTHS = (64, 128, 256, 512)


class _Slots(nn.Module):
    def __init__(self, layers: Dict[int, nn.Module]):
        super().__init__()
        for idx, layer in layers.items():
            self.add_module(str(idx), layer)

    def at(self, idx: int) -> nn.Module:
        return getattr(self, str(idx))


class _Holder(nn.Module):
    pass


class _ShapeOf:
    """Stands in the tape for an activation that is no longer stored: the backward plan sizes its gradient by it."""
    __slots__ = ("shape",)

    def __init__(self, t: torch.Tensor):
        self.shape = tuple(t.shape)


class _ConvP(nn.Module):
    """Parameters of nn.Conv2d(cin, cout, k): logical [cout,cin,k,k], physical [cout][k][k][cin]."""
    transposed = False

    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.weight = nn.Parameter(torch.empty(cout, k, k, cin).permute(0, 3, 1, 2))
        self.bias = nn.Parameter(torch.empty(cout))
        self.reset_parameters()

    def reset_parameters(self):
        # torch's nn.Conv2d default (kaiming_uniform a=sqrt(5)); the reference then applies
        # weights_init (arch/__init__.py:60-65) on top, as get_arch() does here.
        fan_in = self.weight.shape[1] * self.k * self.k
        bound = 1.0 / fan_in ** 0.5
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            self.bias.uniform_(-bound, bound)


class _ConvTP(_ConvP):
    """Parameters of nn.ConvTranspose2d(cin, cout, 2, stride=2): logical [cin,cout,2,2],
    physical [cin][a][b][cout]."""
    transposed = True

    def __init__(self, cin, cout):
        nn.Module.__init__(self)
        self.cin, self.cout, self.k = cin, cout, 2
        self.weight = nn.Parameter(torch.empty(cin, 2, 2, cout).permute(0, 3, 1, 2))
        self.bias = nn.Parameter(torch.empty(cout))
        self.reset_parameters()

    def reset_parameters(self):
        fan_in = self.weight.shape[1] * 4  # torch uses weight.size(1) * receptive field
        bound = 1.0 / fan_in ** 0.5
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            self.bias.uniform_(-bound, bound)


class _BNP(nn.Module):
    """nn.BatchNorm2d(c) parameters and buffers (eps 1e-5, momentum 0.1: torch defaults, as network.py:138,181 use)."""
    is_dct_batchnorm = True

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class UNet(nn.Module):
    """Drop-in for the reference ``UNet(in_channels=1, num_classes=2)`` (network.py:196) and, with ``batchnorm=True``
    (class ``UNet_bn`` below), for ``UNet_bn`` (network.py:243-290).

    ``compute_dtype``: torch.bfloat16 (MFMA bf16, fp32 accumulate) or torch.float32
    (v_mfma_f32_32x32x2_f32; the parity mode).  ``dropout_p`` is the p of the two
    ``nn.Dropout`` sites (network.py:165,210)."""

    batch_independent = True   # no BatchNorm: samples of a batch never interact

    def __init__(self, in_channels: int = 1, num_classes: int = 2, compute_dtype=torch.bfloat16,
                 dropout_p: float = 0.5, batchnorm: bool = False):
        super().__init__()
        self.batchnorm = bool(batchnorm)
        if self.batchnorm:
            self.batch_independent = False     # batch statistics couple the samples of a pass
        if in_channels != 1:
            raise ValueError("dct_amd UNet: the hot path is single-channel slices (in_channels=1)")
        if not 2 <= num_classes <= 8:
            raise ValueError("dct_amd UNet: 2 <= num_classes <= 8")
        if compute_dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("dct_amd UNet: compute_dtype is torch.bfloat16 (MFMA bf16) or torch.float32; fp16 is an Enet mode")
        self.num_classes = num_classes
        self.compute_dtype = compute_dtype
        self.dropout_p = float(dropout_p)
        cin = in_channels
        bn = self.batchnorm
        # Sequential indices of the reference modules (they are the state_dict keys): UNetDec / UNetDec_bn (network.py:153-193),
        # centre (:204-212 / :251-261), UNetEnc / UNetEnc_bn (:115-150), enc1 (:216-221 / :265-271).  The plan addresses the
        # layers by role through self._roles: (first conv, its BatchNorm or None, second conv, its BatchNorm or None[, convT]).
        self._roles: Dict[str, tuple] = {}
        for lvl, width in enumerate(_WIDTHS, start=1):
            blk = _Holder()
            if bn:
                blk.down = _Slots({0: _ConvP(cin, width, 3), 1: _BNP(width), 3: _ConvP(width, width, 3)})
                self._roles[f"dec{lvl}"] = (blk.down.at(0), blk.down.at(1), blk.down.at(3), None)
            else:
                blk.down = _Slots({0: _ConvP(cin, width, 3), 2: _ConvP(width, width, 3)})
                self._roles[f"dec{lvl}"] = (blk.down.at(0), None, blk.down.at(2), None)
            setattr(self, f"dec{lvl}", blk)
            cin = width
        if bn:
            self.center = _Slots({0: _ConvP(512, 1024, 3), 1: _BNP(1024), 3: _ConvP(1024, 1024, 3), 4: _BNP(1024), 7: _ConvTP(1024, 512)})
            c = self.center
            self._roles["center"] = (c.at(0), c.at(1), c.at(3), c.at(4), c.at(7))
        else:
            self.center = _Slots({0: _ConvP(512, 1024, 3), 2: _ConvP(1024, 1024, 3), 5: _ConvTP(1024, 512)})
            c = self.center
            self._roles["center"] = (c.at(0), None, c.at(2), None, c.at(5))
        for lvl, (ci, feat, co) in {4: (1024, 512, 256), 3: (512, 256, 128), 2: (256, 128, 64)}.items():
            blk = _Holder()
            if bn:
                blk.up = _Slots({0: _ConvP(ci, feat, 3), 1: _BNP(feat), 3: _ConvP(feat, feat, 3), 4: _BNP(feat), 6: _ConvTP(feat, co)})
                u = blk.up
                self._roles[f"enc{lvl}"] = (u.at(0), u.at(1), u.at(3), u.at(4), u.at(6))
            else:
                blk.up = _Slots({0: _ConvP(ci, feat, 3), 2: _ConvP(feat, feat, 3), 4: _ConvTP(feat, co)})
                u = blk.up
                self._roles[f"enc{lvl}"] = (u.at(0), None, u.at(2), None, u.at(4))
            setattr(self, f"enc{lvl}", blk)
        if bn:
            self.enc1 = _Slots({0: _ConvP(128, 64, 3), 1: _BNP(64), 3: _ConvP(64, 64, 3)})
            self._roles["enc1"] = (self.enc1.at(0), self.enc1.at(1), self.enc1.at(3), None)
        else:
            self.enc1 = _Slots({0: _ConvP(128, 64, 3), 2: _ConvP(64, 64, 3)})
            self._roles["enc1"] = (self.enc1.at(0), None, self.enc1.at(2), None)
        self.final = _ConvP(64, num_classes, 1)

        self._convs: List[_ConvP] = [m for m in self.modules() if isinstance(m, _ConvP)]
        self.flat_params = FlatParams(list(self.parameters()))
        self.flat_params.want_shadow = compute_dtype == torch.bfloat16
        self._shadow_fresh = False
        self._pidx = {id(p): i for i, p in enumerate(self.flat_params.params)}
        self._packs: Dict[int, dict] = {}
        self._pack_table = None
        self._pack_key = None
        self.external_dropout_masks: Optional[List[torch.Tensor]] = None  # parity replay hook
        self.last_dropout_masks: Optional[List[torch.Tensor]] = None
        self.record_dropout_masks = False
        self.dropout_mask_log: List[List[torch.Tensor]] = []   # with record_dropout_masks: one entry per forward (the caller clears it)
        self._drop_calls = 0                  # host mirror of the device call counter below
        self._drop_parity = 0
        self._drop_state: Optional[torch.Tensor] = None   # int64[2] on the device: the call counter (Philox offset = calls << 40), two words used in turn
        # Philox seed of the two dropout sites: drawn from torch's global generator at construction, so a user seed
        # (torch.manual_seed / fix_all_seed) controls it and co-trained models get decorrelated masks (the reference
        # draws every mask from that generator: network.py:165,210).  ddp.FlatGradSync mixes the rank in.
        self.dropout_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        self._debug: Optional[dict] = None   # tools/debug_unet_layers.py stashes backward intermediates here
        self._grad_hook = None               # data parallelism: called with a bucket index as gradient ranges complete
        self.relu_bits = True                # convolutions keep the ReLU-gate bits of activations whose consumer's data gradient masks by them
        self.pool_codes = True               # max-pool keeps its routing codes for the backward pass (which then does not re-read its input)
        self.wgrad_side_stream = False       # opt-in: weight gradients on a second HIP stream (see _run_backward)
        self.fuse_pool = True                # an encoder block's max pooling rides in its second convolution's call (dct_conv_desc.pool_out)
        self.pool_only = True                # ... whose full-resolution output is then not stored at all (nobody else reads it)
        self.fuse_stem_wgrad = True          # the stem's weight gradient from the epilogue of the data gradient that produces its dy (not stored then)
        self.fuse_skip_grad = True           # the skip connections' bilinear backward gathered by the un-pooling instead of summed in memory
        self.late_packs = True               # the transposed weight packs (first reader: the centre's up-convolution) are launched behind the encoder,
                                             # beside the other chain's kernels, instead of in front of the stem where nothing else runs
        self._pending_packs = None
        self.batch_bias_grads = True         # the four up-convolutions' bias gradients in one launch pair per gradient bucket
        self.batch_skip_resize = True        # the four skip connections' bilinear resizes in one launch, in front of the centre
        self.fuse_drop_pool = True           # the fourth level's dropout + max-pool in one launch (the dropped tensor is never written)
        self.unpool_max_level = 3            # the deepest level that does
        self.unpool_on_load = False          # levels 1..unpool_max_level: the un-pooled gradient of an encoder block is never written -- its two consumers
                                             # expand {pooled gradient + routing codes} while they stage (dct_conv_desc.unpool_codes).  Built, bit-identical,
                                             # and OFF: on the captured cfg2 step level 1 alone is 0.5 % slower and levels 1-3 are 1.2 % slower than the
                                             # un-pooling launches (profiles/r05_unpool_on_load_per_level.txt, r05_knob_sweep.txt) although 0.3 GB of
                                             # writes and 0.7 GB of reads per step are gone: the expansion's ~40 vector instructions per thread and
                                             # K-step land in the filter-row weight gradient's issue-bound loop (+10 % on that kernel)
        self._wgrad_stream = None

    # ------------------------------------------------------------------------------ weights
    def mark_weights_updated(self, shadow_fresh: bool = False):
        """fp32 masters changed; ``shadow_fresh``: the flat bf16 shadow was rewritten in the same pass (fused Adam)."""
        self._pack_key = None
        self._shadow_fresh = bool(shadow_fresh)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.mark_weights_updated()         # (also drops "the bf16 shadow is fresh": it mirrors the weights before the call)
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.mark_weights_updated()
        return r

    def _w(self, conv: _ConvP) -> torch.Tensor:
        return self.flat_params.dense(self._pidx[id(conv.weight)])

    def _gw(self, conv: _ConvP) -> torch.Tensor:
        return self.flat_params.grad_dense(self._pidx[id(conv.weight)])

    def _gb(self, conv: _ConvP) -> torch.Tensor:
        return self.flat_params.grad_dense(self._pidx[id(conv.bias)])

    def _ensure_packs(self, late: bool = False):
        """(Re)build compute-dtype weight packs when the fp32 masters changed.  ``late``: the batched transposing launch is left
        pending for `_run_forward`, which issues it in front of the packs' first reader (`_flush_packs`)."""
        fp = self.flat_params
        if fp.ensure():
            self._shadow_fresh = False
        key = (fp.version, self.compute_dtype, tuple(p._version for p in fp.params))
        if key == self._pack_key:
            return
        dt, dev = self.compute_dtype, fp.flat.device
        use_shadow = dt == torch.bfloat16
        if use_shadow:
            shadow = fp.ensure_shadow()
            if not self._shadow_fresh:          # weights changed outside the fused Adam (init, load, broadcast)
                K.pack_weight(fp.flat, shadow, 1, 1, fp.total)
            self._shadow_fresh = False
        jobs = []
        for conv in self._convs:
            if conv is self.final or conv is self._roles["dec1"][0]:
                continue  # stem and head read the fp32 masters directly
            w = self._w(conv)
            ent = self._packs.get(id(conv))
            if ent is None or ent["dev"] != dev or ent["dt"] != dt:
                ent = {"dev": dev, "dt": dt}
                self._packs[id(conv)] = ent
                self._pack_table = None
                n = w.numel()
                if conv.transposed:
                    ent["fwd"] = torch.empty(n, dtype=dt, device=dev)                    # [(a,b,co)][ci]
                else:
                    ent["dgrad"] = torch.empty(n, dtype=dt, device=dev)                  # [ci][flip taps][co]
            ws = fp.shadow_dense(self._pidx[id(conv.weight)]) if use_shadow else w       # K-major image in dt
            # the packs read the bf16 shadow when there is one (half the bytes of the fp32 masters, same values:
            # both are round-to-nearest of the same fp32 weights)
            if conv.transposed:
                jobs.append((ws, ent["fwd"], conv.cin, 4, conv.cout, 2, False))
                ent["dgrad"] = ws                                                        # [ci][a][b][co] as stored
            else:
                ent["fwd"] = ws                                                          # [co][r][s][ci] as stored
                jobs.append((ws, ent["dgrad"], conv.cout, conv.k * conv.k, conv.cin, 1, True))
        # every transposed pack of the network in one launch (dct_pack_weights_batched); the job table holds device
        # addresses, so it is rebuilt whenever the flat buffer or a pack was re-allocated
        tkey = (fp.version, fp.flat.data_ptr(), fp.shadow.data_ptr() if use_shadow else 0, dev, dt)
        if self._pack_table is None or self._pack_table[0] != tkey:
            self._pack_table = (tkey,) + K.pack_jobs_table(jobs, dev)
        _, table, njobs, tiles, edge = self._pack_table
        self._pending_packs = (table, njobs, tiles, dt, edge)
        if not late:
            self._flush_packs()
        self._pack_key = key

    def _flush_packs(self):
        job, self._pending_packs = self._pending_packs, None
        if job is not None:
            K.pack_weights_batched(*job)

    # ------------------------------------------------------------------------------ forward
    def _check_input(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("dct_amd UNet runs on the HIP device only (no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"expected [B,1,H,W], got {tuple(x.shape)}")
        if min(x.shape[2], x.shape[3]) < 176:
            # same failure the reference has (valid convs): SURVEY.md fact 3
            raise RuntimeError("Kernel size can't be greater than actual input size (UNet needs H,W >= 176)")

    # Autograd-free entry points of the execution plan (the fused trainer step drives them directly: no autograd
    # engine thread hop, and the whole step is a plain launch sequence that a HIP graph can capture).
    def plan_forward(self, x: torch.Tensor, save: bool = True, keep_predrop: bool = False, reuse=None):
        """-> (logits, tape): logits physical NHWC fp32 [B,H,W,C]; tape feeds plan_backward (None if not save).
        ``keep_predrop``: a training pass also keeps the fourth encoder level's output in front of its dropout in the tape.  ``reuse``: the tape
        of such a pass over THE SAME INPUT TENSOR with the same weights: everything in front of the first dropout -- the stem and the eight
        encoder convolutions, half of a forward pass -- is the same computation and is taken from that tape (shared, read-only) instead of
        run again; the pass starts at the fourth level's dropout with masks of its own.  (The FGSM generator's clean forward pass follows the
        joint forward pass of the same network over the same batch: AEGenerator.py:27 behind cotraining_totalloss.py:208-227.)"""
        self._check_input(x)
        self._ensure_packs(late=bool(self.late_packs))
        return self._run_forward(x, save, keep_predrop=keep_predrop, reuse=reuse)

    supports_grad_overwrite = True
    supports_forward_reuse = True        # plan_forward(keep_predrop=..., reuse=...)

    def plan_backward(self, tape, dlogits: torch.Tensor, need_dx: bool = False, need_dw: bool = True,
                      overwrite: bool = False):
        """dlogits: NHWC fp32 like the logits.  Parameter gradients accumulate into (``overwrite``: replace the
        contents of) the flat gradient buffer (attached as p.grad); returns d/dx as [B,1,H,W] when need_dx."""
        if need_dw:
            self.flat_params.ensure_grads()
        dx = self._run_backward(tape, dlogits, need_dx, need_dw, overwrite=overwrite and need_dw)
        return dx.reshape(dx.shape[0], 1, dx.shape[1], dx.shape[2]) if dx is not None else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input(x)
        self._ensure_packs()
        params = self.flat_params.params
        save = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _UNetFn.apply(self, save, x, *params)

    def grad_bucket_ranges(self):
        """[lo, hi) element ranges of the flat gradient buffer in the order the backward pass completes them:
        decoder side ("enc*" + final), centre, encoder side ("dec*").  The flat buffer is in registration order
        dec1..dec4, center, enc4..enc1, final, so each bucket is contiguous; the gradient exchange of a bucket can
        start while the rest of the backward still runs (_grad_hook, ddp.FlatGradSync.begin_bucket)."""
        fp = self.flat_params
        first_center = self._pidx[id(self._roles["center"][0].weight)]
        first_enc4 = self._pidx[id(self._roles["enc4"][0].weight)]
        return [(fp.offsets[first_enc4], fp.total), (fp.offsets[first_center], fp.offsets[first_enc4]),
                (0, fp.offsets[first_center])]

    def _side_stream(self, dev):
        if not self.wgrad_side_stream:
            return None
        if self._wgrad_stream is None or self._wgrad_stream.device != dev:
            self._wgrad_stream = torch.cuda.Stream(device=dev)
        return self._wgrad_stream

    # The plan itself ------------------------------------------------------------------------
    def _run_forward(self, x: torch.Tensor, save: bool, keep_predrop: bool = False, reuse=None):
        dt, dev = self.compute_dtype, x.device
        B, _, H, W = x.shape
        P = self._packs
        A: Dict[str, torch.Tensor] = {}

        def new(h, w, c, dtype=dt):
            return torch.empty(B, h, w, c, dtype=dtype, device=dev)

        bn_recs: Dict[str, tuple] = {}

        def bn_relu(raw, bn, name):
            """nn.BatchNorm2d + ReLU of unet_bn: statistics of the raw conv output -> y = relu(scale * raw + shift)."""
            vec = torch.empty(4, bn.num_features, dtype=torch.float32, device=dev)
            y = torch.empty_like(raw)
            K.bn_fwd(raw, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                     self.training, vec[0], vec[1], vec[2], vec[3], y=y, relu=True)
            bn_recs[name] = (raw, vec, bn)
            return y

        bits: Dict[int, torch.Tensor] = {}     # ReLU-gate bits of the activations whose consumer's data gradient masks by them

        def want_bits(t, bn):
            if not (save and self.relu_bits and bn is None):
                return None
            b = K.relu_bits_like(t)
            if b is not None:
                bits[id(t)] = b
            return b

        def conv3(src, conv, dst, bn=None, name=None, gate=False, pool_out=None, pool_codes=None, pool_only=False):
            """``gate``: dst feeds a convolution whose data gradient is masked by (dst > 0) -- keep the one-bit image of it."""
            K.conv2d(src, P[id(conv)]["fwd"], conv.bias, dst, relu=bn is None, relu_bits_out=want_bits(dst, bn) if gate else None,
                     pool_out=pool_out, pool_codes=pool_codes, pool_only=pool_only)
            return dst if bn is None else bn_relu(dst, bn, name)

        xs = x.detach().to(torch.float32).reshape(B, H, W, 1)
        if not xs.is_contiguous():
            xs = xs.contiguous()
        A["x"] = xs
        training = self.training and self.dropout_p > 0
        masks_out = [] if self.record_dropout_masks else None

        def dropout(src, which):
            if self.external_dropout_masks is not None:
                dst = torch.empty_like(src)
                K.dropout_apply(src, dst, self.external_dropout_masks[which], self.dropout_p)
                return dst
            if not training:
                return src
            dst = torch.empty_like(src)
            m = torch.empty(src.shape, dtype=torch.uint8, device=dev) if masks_out is not None else None
            state, parity = drop_turn()
            K.dropout_fwd(src, dst, self.dropout_p, self.dropout_seed, 0, mask_out=m, calls_dev=state, parity=parity)
            if masks_out is not None:
                masks_out.append(m)
            return dst

        def drop_turn():
            """-> (device call counter, the word this launch reads); the launch advances the device copy in stream order (graph-replayable)."""
            if self._drop_state is None or self._drop_state.device != dev:
                self._drop_state = torch.tensor([self._drop_calls, self._drop_calls], dtype=torch.int64, device=dev)
                self._drop_parity = 0
            self._drop_calls += 1
            parity = self._drop_parity
            self._drop_parity ^= 1      # (two sites per forward pass: a captured step holds an even number of launches, so its replays stay in turn)
            return self._drop_state, parity

        # encoder ("dec" in the reference's naming)
        h, w = H, W
        src = xs
        if reuse is not None:
            # same input tensor, same weights (the packs' key), a training pass with the pre-dropout tensor kept: else run everything
            ok = (training and not self.batchnorm and self.external_dropout_masks is None and masks_out is None and self._debug is None
                  and reuse.get("d4pre") is not None and reuse.get("pack_key") == self._pack_key
                  and reuse["x"].data_ptr() == xs.data_ptr() and reuse["x"].shape == xs.shape and all(reuse.get(f"pc{k}") is not None for k in (1, 2, 3)))
            if not ok:
                reuse = None
        for lvl, width in enumerate(_WIDTHS, start=1):
            if reuse is not None and lvl < 4:
                a, dd, p, codes = (reuse[f"{k}{lvl}"] for k in ("a", "d", "p", "pc"))
                if id(a) in reuse["bits"]:
                    bits[id(a)] = reuse["bits"][id(a)]
                h, w = p.shape[1], p.shape[2]
                A[f"a{lvl}"], A[f"d{lvl}"], A[f"p{lvl}"], A[f"pc{lvl}"] = a, dd, p, codes
                src = p
                continue
            ca, bna, cb, _ = self._roles[f"dec{lvl}"]
            if reuse is not None:                # level 4: the two convolutions from the tape, dropout + pool with this pass's mask
                a, d = reuse["a4"], reuse["d4pre"]
                if id(a) in reuse["bits"]:
                    bits[id(a)] = reuse["bits"][id(a)]
                hp, wp = (h - 4 + 1) // 2, (w - 4 + 1) // 2
                codes = torch.empty(B, hp, wp, width, dtype=torch.uint8, device=dev)
                p = new(hp, wp, width)
                state, parity = drop_turn()
                K.dropout_maxpool_fwd(d, p, codes, self.dropout_p, self.dropout_seed, state, parity)
                h, w = hp, wp
                A[f"a{lvl}"], A[f"d{lvl}"], A[f"p{lvl}"], A[f"pc{lvl}"] = a, _ShapeOf(d), p, codes
                src = p
                continue
            a = new(h - 2, w - 2, width)
            if lvl == 1:
                K.conv_cin1_fwd(xs, self._w(ca), ca.bias, a, relu=bna is None, relu_bits_out=want_bits(a, bna))
                if bna is not None:
                    a = bn_relu(a, bna, "a1")
            else:
                a = conv3(src, ca, a, bna, f"a{lvl}", gate=True)
            hp, wp = (h - 4 + 1) // 2, (w - 4 + 1) // 2
            codes = torch.empty(B, hp, wp, width, dtype=torch.uint8, device=dev) if (save and self.pool_codes) else None
            # the pool reads the convolution's output as it is (no dropout in between: every level but a training pass's fourth)
            fuse = self.fuse_pool and not (lvl == 4 and (training or self.external_dropout_masks is not None))
            p = new(hp, wp, width)
            # ... and alone: the backward pass routes by the codes and never reads the block's full-resolution output
            only = fuse and self.pool_only and (codes is not None or not save) and self._debug is None
            d = conv3(a, cb, new(h - 4, w - 4, width), pool_out=p if fuse else None, pool_codes=codes if fuse else None, pool_only=only)
            # the fourth level of a training pass: dropout and the pool behind it in one launch; the dropped tensor is never written
            drop_pool = (lvl == 4 and not fuse and training and self.fuse_drop_pool and self.external_dropout_masks is None and masks_out is None
                         and codes is not None and self._debug is None)
            if drop_pool:
                state, parity = drop_turn()
                K.dropout_maxpool_fwd(d, p, codes, self.dropout_p, self.dropout_seed, state, parity)
                if keep_predrop and save:
                    A["d4pre"], A["pack_key"] = d, self._pack_key        # for a later pass over the same input (plan_forward(reuse=...))
                d = dd = _ShapeOf(d)
            else:
                dd = dropout(d, 0) if lvl == 4 else d
            if only:
                # nobody reads the block's full-resolution output again (the backward pass routes by the codes and needs its SHAPE):
                # the buffer goes back to the allocator now instead of riding in the tape until the backward pass (130 MB at level 1)
                d = dd = _ShapeOf(d)
            h, w = hp, wp
            if not fuse and not drop_pool:
                K.maxpool_fwd(dd, p, codes=codes)
            A[f"a{lvl}"], A[f"d{lvl}"], A[f"p{lvl}"], A[f"pc{lvl}"] = a, dd, p, codes
            src = p
        # center
        self._flush_packs()                  # first reader of a transposed pack: the centre's up-convolution below
        # The four skip connections' resizes (pooled tensor -> second half of the decoder level's concatenation) in ONE launch here, where all
        # four pooled tensors exist, instead of one launch per decoder level (UNet.batch_skip_resize; dct_bilinear_fwd_batched)
        cats = {}
        if self.batch_skip_resize:
            ch, cw = h, w
            for lvl, co in ((4, 512), (3, 256), (2, 128), (1, 64)):
                ch, cw = 2 * (ch - 4), 2 * (cw - 4)
                cats[lvl] = new(ch, cw, 2 * co)
            K.bilinear_fwd_batched([A[f"p{lvl}"] for lvl in (4, 3, 2, 1)], [cats[lvl][..., cats[lvl].shape[3] // 2:] for lvl in (4, 3, 2, 1)])
        ca, bna, cb, bnb, ct = self._roles["center"]
        c1 = conv3(src, ca, new(h - 2, w - 2, 1024), bna, "c1", gate=True)
        c2 = conv3(c1, cb, new(h - 4, w - 4, 1024), bnb, "c2")
        c2d = dropout(c2, 1)
        h, w = 2 * (h - 4), 2 * (w - 4)
        cat = cats[4] if cats else new(h, w, 1024)
        K.conv2d(c2d, P[id(ct)]["fwd"], ct.bias, cat[..., :512], R=1, S=1, relu=True, scatter2x2=True)
        if not cats:
            K.bilinear_fwd(A["p4"], cat[..., 512:])
        A["c1"], A["c2"], A["cat4"] = c1, c2d, cat
        # decoder ("enc")
        for lvl, feat, co in ((4, 512, 256), (3, 256, 128), (2, 128, 64)):
            ca, bna, cb, bnb, ct = self._roles[f"enc{lvl}"]
            ea = conv3(cat, ca, new(h - 2, w - 2, feat), bna, f"e{lvl}a", gate=True)
            eb = conv3(ea, cb, new(h - 4, w - 4, feat), bnb, f"e{lvl}b", gate=True)
            h, w = 2 * (h - 4), 2 * (w - 4)
            cat = cats[lvl - 1] if cats else new(h, w, 2 * co)
            K.conv2d(eb, P[id(ct)]["fwd"], ct.bias, cat[..., :co], R=1, S=1, relu=True, scatter2x2=True)
            if not cats:
                K.bilinear_fwd(A[f"p{lvl - 1}"], cat[..., co:])
            A[f"e{lvl}a"], A[f"e{lvl}b"], A[f"cat{lvl - 1}"] = ea, eb, cat
        ca, bna, cb, _ = self._roles["enc1"]
        e1a = conv3(cat, ca, new(h - 2, w - 2, 64), bna, "e1a", gate=True)
        e1b = conv3(e1a, cb, new(h - 4, w - 4, 64))
        f = K.head_fwd(e1b, self._w(self.final), self.final.bias, new(h - 4, w - 4, self.num_classes, torch.float32))
        logits = K.bilinear_fwd(f, new(H, W, self.num_classes, torch.float32))
        A["bn"], A["bn_training"], A["bits"] = bn_recs, bool(self.training), bits
        if bn_recs and self.training:      # nn.BatchNorm2d bookkeeping: one multi-tensor launch for all thirteen layers
            torch._foreach_add_([rec[2].num_batches_tracked for rec in bn_recs.values()], 1)
        A["e1a"], A["e1b"] = e1a, e1b
        A["fshape"] = (h - 4, w - 4)
        if masks_out is not None:
            self.last_dropout_masks = masks_out
            self.dropout_mask_log.append(masks_out)
        drop_scale = 1.0 / (1.0 - self.dropout_p) if (training or self.external_dropout_masks is not None) else 1.0
        A["drop_scale"] = drop_scale
        return logits, (A if save else None)

    def _run_backward(self, A, dlogits: torch.Tensor, need_dx: bool, need_dw: bool, overwrite: bool = False):
        """dlogits: fp32 NHWC [B,H,W,C].  Accumulates parameter grads in place (``overwrite``: writes them -- every
        parameter of the network receives a gradient in every backward pass, so the first pass of a step can replace
        the zero fill of the 124 MB gradient buffer and the read-modify-write of the folds); returns dx or None."""
        gacc = not overwrite
        dt, dev = self.compute_dtype, dlogits.device
        B = dlogits.shape[0]
        P = self._packs
        C = self.num_classes
        ds = A["drop_scale"]

        def new_like(t, c=None):
            return torch.empty(t.shape[:3] + ((c,) if c else t.shape[3:]), dtype=dt, device=dev)

        # Weight gradients are off the critical path (only the data-gradient chain feeds the next layer): they
        # are queued on a second stream, ordered after the producer of dy by an event, and joined at the end.
        # Their operands stay referenced until the join (`keep`), so the caching allocator cannot hand a dy
        # buffer to a later data-gradient while a weight-gradient kernel still reads it.
        gate_bits = A.get("bits") or {}
        cur = torch.cuda.current_stream(dev)
        side = self._side_stream(dev) if need_dw else None
        keep: List[torch.Tensor] = []

        def on_side(*operands):
            if side is None:
                return contextlib.nullcontext()
            side.wait_stream(cur)
            keep.extend(operands)
            return torch.cuda.stream(side)

        stem_fused = [False]

        def conv_bwd(conv, x_in, dy, dx_out, mask=None, mask_channels=0, mask_scale=1.0, accumulate=False, stem=None, unpool=None):
            """dy: grad wrt the conv's pre-activation output (already ReLU-masked).  ``stem``: (x, dw, db, accumulate) -- try to take
            the stem's weight gradient from this data gradient's output tile (dx_out is then not written; stem_fused[0] tells).
            ``unpool`` = (codes, H, W): dy is the gradient at the POOLED tensor; the kernels expand it to the un-pooled gradient while
            they stage where they can, else it is un-pooled into a buffer here, once, for whoever still needs it."""
            def materialised():
                codes, uh, uw = unpool
                full = torch.empty(dy.shape[0], uh, uw, dy.shape[3], dtype=dt, device=dev)
                return K.maxpool_bwd(None, dy, full, relu_mask=True, scale=1.0, codes=codes)
            if need_dw:
                with on_side(dy, x_in):
                    if dt == torch.bfloat16:     # bias gradient rides along in the weight-gradient launch
                        try:
                            K.conv2d_wgrad(dy, x_in, self._gw(conv), accumulate=gacc, db=self._gb(conv), unpool=unpool)
                        except K.UnpoolOnLoadUnsupported:
                            dy, unpool = materialised(), None
                            K.conv2d_wgrad(dy, x_in, self._gw(conv), accumulate=gacc, db=self._gb(conv))
                    else:
                        if unpool is not None:
                            dy, unpool = materialised(), None
                        K.conv2d_wgrad(dy, x_in, self._gw(conv), accumulate=gacc)
                        K.bias_grad(dy, self._gb(conv), accumulate=gacc)
            if dx_out is not None:
                mb = gate_bits.get(id(mask)) if mask is not None else None
                if stem is not None and mb is not None:
                    try:
                        K.conv2d(dy, P[id(conv)]["dgrad"], None, dx_out, pad_h=2, pad_w=2, mask=mask, mask_bits=mb, stem=stem, unpool=unpool)
                        stem_fused[0] = True
                        return dx_out
                    except K.StemFusionUnsupported:
                        pass                     # (the stem's weight gradient then runs as its own launch)
                if unpool is not None:
                    try:
                        K.conv2d(dy, P[id(conv)]["dgrad"], None, dx_out, pad_h=2, pad_w=2, mask=mask, mask_channels=mask_channels,
                                 mask_scale=mask_scale, accumulate=accumulate, mask_bits=mb, unpool=unpool)
                        return dx_out
                    except K.UnpoolOnLoadUnsupported:
                        dy, unpool = materialised(), None
                K.conv2d(dy, P[id(conv)]["dgrad"], None, dx_out, pad_h=2, pad_w=2, mask=mask,
                         mask_channels=mask_channels, mask_scale=mask_scale, accumulate=accumulate, mask_bits=mb)
            return dx_out

        # The up-convolutions' bias gradients (column sums of their dy, which is the OTHER operand of their weight-gradient GEMM: they cannot ride
        # along there) are collected and summed by ONE launch pair per gradient bucket (dct_bias_grad_batched) instead of one pair per level:
        # six launches of ~5 us less on every model's backward chain.
        pending_bias: List[tuple] = []

        def flush_bias():
            if pending_bias:
                dys, dbs = [t[0] for t in pending_bias], [t[1] for t in pending_bias]
                pending_bias.clear()
                with on_side(*dys):
                    K.bias_grad_batched(dys, dbs, accumulate=gacc)

        def convT_bwd(conv, x_in, dy, dx_out, mask, mask_scale=1.0):
            if need_dw:
                with on_side(dy, x_in):
                    K.conv2d_wgrad(x_in, dy, self._gw(conv), R=2, S=2, stride=2, accumulate=gacc)
                pending_bias.append((dy, self._gb(conv)))
                if not self.batch_bias_grads:
                    flush_bias()
            K.conv2d(dy, P[id(conv)]["dgrad"], None, dx_out, R=2, S=2, stride=2, mask=mask, mask_scale=mask_scale,
                     mask_bits=gate_bits.get(id(mask)))
            return dx_out

        def bn_back(name, g):
            """g: gradient wrt the ReLU output of a BatchNorm'd conv -> gradient wrt the raw conv output (in place)."""
            rec = A["bn"].get(name)
            if rec is None:
                return g
            raw, vec, bn = rec
            c1c2 = torch.empty(2 * bn.num_features, dtype=torch.float32, device=dev)
            K.bn_bwd(raw, g, vec[0], vec[1], vec[2], vec[3],
                     self.flat_params.grad_dense(self._pidx[id(bn.weight)]) if need_dw else None,
                     self.flat_params.grad_dense(self._pidx[id(bn.bias)]) if need_dw else None,
                     c1c2, g, training=A["bn_training"], relu=True, accumulate=gacc)
            return g

        fh, fw = A["fshape"]
        df = K.bilinear_bwd(dlogits, torch.empty(B, fh, fw, C, dtype=torch.float32, device=dev))
        e1b, e1a = A["e1b"], A["e1a"]
        de1b = new_like(e1b)
        # (the classifier's weight gradient is a leaf, but launched later -- behind enc1's gradients, out of the stretch where nothing else runs --
        # the step is 0.3 % SLOWER: profiles/r05_dead_zone_ab.txt)
        K.head_bwd(e1b, df, self._w(self.final), de1b,
                   self._gw(self.final) if need_dw else None, self._gb(self.final) if need_dw else None,
                   relu_mask=True, accumulate=gacc)
        ca, _, cb, _ = self._roles["enc1"]
        de1a = bn_back("e1a", conv_bwd(cb, e1a, de1b, new_like(e1a), mask=e1a))
        cat = A["cat1"]
        dcat = conv_bwd(ca, cat, de1a, new_like(cat), mask=cat, mask_channels=64)
        dp: Dict[int, torch.Tensor] = {}
        skip_g: Dict[int, torch.Tensor] = {}
        skip_fused = bool(self.fuse_skip_grad and all(A.get(f"pc{k}") is not None for k in (1, 2, 3, 4)))
        # Levels whose un-pooled gradient is expanded on load (no un-pooling launch to gather the skip gradient in): the bilinear
        # backward of the skip connection is written to dp and the next block's data gradient accumulates onto it.
        up_levels = {k for k in (1, 2, 3) if self.unpool_on_load and k <= int(self.unpool_max_level) and dt == torch.bfloat16 and A.get(f"pc{k}") is not None and
                     isinstance(A[f"d{k}"], _ShapeOf) and self._debug is None}
        for lvl, co in ((2, 64), (3, 128), (4, 256)):
            ca, _, cb, _, ct = self._roles[f"enc{lvl}"]
            ea, eb = A[f"e{lvl}a"], A[f"e{lvl}b"]
            p = A[f"p{lvl - 1}"]
            if skip_fused and (lvl - 1) not in up_levels:
                skip_g[lvl - 1] = dcat[..., co:]          # gathered by level (lvl - 1)'s un-pooling (K.maxpool_bwd(..., skip=))
            else:
                dp[lvl - 1] = K.bilinear_bwd(dcat[..., co:], new_like(p))
            deb = bn_back(f"e{lvl}b", convT_bwd(ct, eb, dcat[..., :co], new_like(eb), mask=eb))
            dea = bn_back(f"e{lvl}a", conv_bwd(cb, ea, deb, new_like(ea), mask=ea))
            if self._debug is not None:
                self._debug[f"de{lvl}b"], self._debug[f"de{lvl}a"], self._debug[f"dcat{lvl - 1}"] = deb, dea, dcat
            cat = A[f"cat{lvl}"]
            dcat = conv_bwd(ca, cat, dea, new_like(cat), mask=cat, mask_channels=2 * co)
        hook = self._grad_hook if need_dw else None
        if hook is not None and side is None:
            flush_bias()
            hook(0)                           # decoder-side gradients are complete
        # center (cat4: 512 convT channels + 512 skip channels)
        ca, _, cb, _, ct = self._roles["center"]
        p4 = A["p4"]
        if skip_fused:
            skip_g[4] = dcat[..., 512:]
            dp[4] = new_like(p4)
        else:
            dp[4] = K.bilinear_bwd(dcat[..., 512:], new_like(p4))
        c2d, c1 = A["c2"], A["c1"]
        dc2 = bn_back("c2", convT_bwd(ct, c2d, dcat[..., :512], new_like(c2d), mask=c2d, mask_scale=ds))
        dc1 = bn_back("c1", conv_bwd(cb, c1, dc2, new_like(c1), mask=c1))
        conv_bwd(ca, p4, dc1, dp[4], accumulate=not skip_fused)
        flush_bias()                          # (the centre's up-convolution was the last one)
        if hook is not None and side is None:
            hook(1)                           # centre gradients are complete
        # encoder
        dx = None
        for lvl in (4, 3, 2, 1):
            ca, _, cb, _ = self._roles[f"dec{lvl}"]
            a, d = A[f"a{lvl}"], A[f"d{lvl}"]
            unpool = None
            if lvl in up_levels:
                dd, unpool = dp[lvl], (A[f"pc{lvl}"], d.shape[1], d.shape[2])      # the pooled gradient stands for the un-pooled one
            else:
                dd = K.maxpool_bwd(d, dp[lvl], new_like(d), relu_mask=True, scale=ds if lvl == 4 else 1.0, codes=A.get(f"pc{lvl}"),
                                   skip=skip_g.get(lvl))
            stem = None
            if (lvl == 1 and need_dw and not need_dx and self.fuse_stem_wgrad and dt == torch.bfloat16 and side is None and
                    A["bn"].get("a1") is None and self._debug is None):
                stem = (A["x"], self._gw(ca), self._gb(ca), gacc)       # the stem's dy has no other reader: see dct_conv_desc.stem_x
            da = bn_back(f"a{lvl}", conv_bwd(cb, a, dd, new_like(a), mask=a, stem=stem, unpool=unpool))
            if lvl > 1:
                summed = not skip_fused or (lvl - 1) in up_levels          # dp[lvl - 1] already holds the skip connection's share
                if not summed:
                    dp[lvl - 1] = new_like(A[f"p{lvl - 1}"])
                conv_bwd(ca, A[f"p{lvl - 1}"], da, dp[lvl - 1], accumulate=summed)
            else:
                c0 = ca
                if need_dw and not stem_fused[0]:
                    with on_side(da):
                        K.conv_cin1_wgrad(A["x"], da, self._gw(c0), self._gb(c0), accumulate=gacc)
                if need_dx:
                    dx = K.conv_cin1_dgrad(da, self._w(c0), torch.empty_like(A["x"]), pad_h=0, pad_w=0)
        if side is not None:
            cur.wait_stream(side)
        elif hook is not None:
            hook(2)                           # encoder-side gradients: the whole buffer is final
        return dx


class UNet_bn(UNet):
    """Drop-in for the reference ``UNet_bn`` (network.py:243-290; registry name ``unet_bn``): BatchNorm2d + ReLU after the
    first convolution of every encoder block and of ``enc1``, and after both convolutions of the centre and of the three
    decoder blocks.  Same plan as ``UNet`` with one statistics + one apply pass per BatchNorm (csrc/bn.hip)."""

    def __init__(self, in_channels: int = 1, num_classes: int = 2, compute_dtype=torch.bfloat16, dropout_p: float = 0.5):
        super().__init__(in_channels, num_classes, compute_dtype, dropout_p, batchnorm=True)


class _UNetFn(torch.autograd.Function):
    """One autograd node for the whole network.  Parameter gradients are accumulated directly
    into the net's flat gradient buffer (``p.grad`` views), so ``backward`` returns None for
    them; only d/dx (FGSM, AEGenerator.py:27-28) is returned as a tensor."""

    @staticmethod
    def forward(ctx, net: UNet, save: bool, x: torch.Tensor, *params):
        need_dw = any(p.requires_grad for p in params)
        logits, acts = net._run_forward(x, save)
        ctx.net, ctx.acts, ctx.need_dw = net, acts, need_dw
        ctx.set_materialize_grads(False)
        # logical NCHW view of the physical NHWC logits (channels_last strides)
        return logits.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        net: UNet = ctx.net
        n_params = len(net.flat_params.params)
        if g is None or ctx.acts is None:
            return (None, None, None) + (None,) * n_params
        dl = g.permute(0, 2, 3, 1)
        if dl.dtype != torch.float32 or not dl.is_contiguous():
            dl = dl.to(torch.float32).contiguous()
        need_dx = ctx.needs_input_grad[2]
        need_dw = ctx.need_dw and any(ctx.needs_input_grad[3:])
        if need_dw:
            net.flat_params.ensure_grads()
        dx = net._run_backward(ctx.acts, dl, need_dx, need_dw)
        ctx.acts = None
        gx = dx.reshape(dx.shape[0], 1, dx.shape[1], dx.shape[2]) if dx is not None else None
        return (None, None, gx) + (None,) * n_params

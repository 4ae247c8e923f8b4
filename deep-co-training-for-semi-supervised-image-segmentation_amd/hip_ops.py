"""Functional wrappers over the C ABI (one per kernel family), operating on physical-NHWC
torch tensors that only provide device memory and the stream.  Used by the networks
(arch/), the loss modules, the optimizer and the parity tests.  HIP only."""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import BF16, DTYPE_OF, F32, EnetTf, call, conv_desc, ptr, stream, view


# The open pass group / leaf side of THIS thread (the library keeps its recording state per thread too: csrc/enet.hip g_grp)
_tls = threading.local()


def _group():
    return getattr(_tls, "group", None)


_WS_CACHE = {}      # (device index, stream handle) -> uint8 workspace, grown geometrically


def _ws(nbytes: int, device) -> Optional[torch.Tensor]:
    """Scratch for one launch (split-K slabs, partial sums).  Eager launches reuse ONE buffer per stream -- launches of a
    stream run in order, so the next one may overwrite it -- instead of hitting the allocator per call.  Under stream
    capture the buffer comes from the graph's private pool as before: a cached buffer baked into a graph could be
    replaced (and freed) by a later, larger request while the graph still replays."""
    nbytes = max(int(nbytes), 16)
    if _group() is not None:        # recorded launches of different members run at the same time: scratch of their own, kept until then
        return keep(torch.empty(nbytes, dtype=torch.uint8, device=device))
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 2 * buf.numel() if buf is not None else 1 << 20), dtype=torch.uint8, device=dev)
        _WS_CACHE[key] = buf
    return buf


class PassGroup(object):
    """Records the Enet launches of ``members`` independent passes and issues them as one chain of grouped launches
    (include/dct.h "grouped passes").  Use::

        with K.PassGroup(n) as grp:
            for m in range(n):
                grp.member(m)
                ... plan_forward / plan_backward of pass m ...
        # leaving the block launches on the current stream

    Everything a member allocates while recording (`keep`) -- activations, scratch -- stays referenced until the launches
    are issued: the caching allocator would otherwise hand member m + 1 the blocks member m has already "freed" although
    none of m's kernels has run."""

    def __init__(self, members: int):
        self.n = int(members)
        self.kept: list = []
        self.grouped = self.single = 0

    def __enter__(self):
        assert _group() is None, "pass groups do not nest"
        call("dct_group_begin", self.n)
        _tls.group = self
        _lib.set_group_open(True)
        return self

    def member(self, m: int) -> None:
        call("dct_group_member", int(m))

    def __exit__(self, et, ev, tb):
        _tls.group = None
        _lib.set_group_open(False)
        if et is not None:
            _lib.load().dct_group_abort()
            self.kept.clear()
            return False
        g, s = C.c_int(0), C.c_int(0)
        call("dct_group_end", stream(), C.byref(g), C.byref(s))
        self.grouped, self.single = g.value, s.value
        self.kept.clear()
        return False


class LeafSide(object):
    """Holds back the weight- and bias-gradient launches of a backward pass (the leaves of its data-gradient chain) so that
    they can be issued on ANOTHER queue (include/dct.h dct_leaves_*)::

        with K.LeafSide() as side:
            net.plan_backward(..., leaf_hook=hook)     # hook(): event on the chain's stream; side stream waits; side.flush() there
        keep = side.kept                                # stays referenced until the side stream has been joined

    ``flush`` issues what is held so far on the CURRENT stream.  Leaving the block issues any rest on the current stream."""

    def __init__(self):
        self.kept: list = []
        self.launches = 0

    def __enter__(self):
        assert _group() is None, "pass groups / leaf sides do not nest"
        call("dct_leaves_begin")
        _tls.group = self
        return self

    def flush(self) -> None:
        n = C.c_int(0)
        call("dct_leaves_flush", stream(), C.byref(n))
        self.launches += n.value

    def __exit__(self, et, ev, tb):
        _tls.group = None
        if et is not None:
            _lib.load().dct_group_abort()
            return False
        n = C.c_int(0)
        call("dct_leaves_end", stream(), C.byref(n))
        self.launches += n.value
        return False



def keep(t):
    """Tensor ``t`` (allocated by a pass that may be recording into a PassGroup) -> t, referenced until the group launches."""
    if _group() is not None:
        _group().kept.append(t)
    return t


def group_max() -> int:
    return int(_lib.load().dct_group_max())


def _dt(t: torch.Tensor) -> int:
    try:
        return DTYPE_OF[t.dtype]
    except KeyError:
        raise RuntimeError(f"dct_amd: unsupported dtype {t.dtype}")


# ------------------------------------------------------------------------------ conv family
def relu_bits_like(t: torch.Tensor):
    """Buffer for the ReLU-gate bits of a dense bf16 NHWC tensor (one bit per element), or None where bits do not apply."""
    if t.dtype != torch.bfloat16 or t.shape[3] % 8 or not t.is_contiguous():
        return None
    return torch.empty(t.shape[0], t.shape[1], t.shape[2], t.shape[3] // 8, dtype=torch.uint8, device=t.device)


def _bits_ok(bits, of):
    if bits is None:
        return None
    assert of is not None and bits.dtype == torch.uint8 and bits.is_contiguous() and of.is_contiguous()
    assert tuple(bits.shape) == (of.shape[0], of.shape[1], of.shape[2], of.shape[3] // 8), (bits.shape, of.shape)
    return bits


class StemFusionUnsupported(RuntimeError):
    """conv2d(..., stem=...) on a layer whose kernel cannot take the stem's weight gradient along (the caller runs the two launches instead)."""


class UnpoolOnLoadUnsupported(RuntimeError):
    """conv2d / conv2d_wgrad(..., unpool=...) on a layer whose kernel cannot expand the pooled gradient while it stages (nothing was
    launched: the caller un-pools into a buffer with `maxpool_bwd` and calls again without ``unpool``)."""


def _check_unpool(unpool, pooled):
    codes, H, W = unpool
    want = (pooled.shape[0], (H + 1) // 2, (W + 1) // 2, pooled.shape[3])
    assert tuple(pooled.shape) == want and pooled.is_contiguous(), "unpool: the input is the DENSE gradient at the pooled tensor"
    assert codes.dtype == torch.uint8 and codes.is_contiguous() and tuple(codes.shape) == want, "unpool: dense routing codes of the pooling"


def conv2d(x, w_packed, bias, y, *, R=3, S=3, stride=1, dil=1, pad_h=0, pad_w=0, relu=False, mask=None,
           mask_channels=0, mask_scale=1.0, accumulate=False, scatter2x2=False, mask_bits=None, relu_bits_out=None,
           pool_out=None, pool_codes=None, pool_only=False, stem=None, unpool=None):
    """y (NHWC view, written in place) = epilogue(conv(x, w_packed)); see include/dct.h dct_conv2d.

    ``unpool`` = (codes, H, W): x is the gradient at a max-pooled tensor and the convolution runs on its un-pooled [N,H,W,C] form, which
    the kernel expands from (x, codes) while it stages (dct_conv_desc.unpool_*); raises ``UnpoolOnLoadUnsupported`` (nothing launched)
    when the layer does not take a kernel that can.

    ``relu_bits_out`` (uint8 [N,H,W,C/8], dense y): the launch also leaves the ReLU-gate bits of y there (`relu_bits_like`);
    ``mask_bits``: such bits of ``mask`` -- the data gradient then reads 1/16 of the bytes where its epilogue can.
    ``pool_out`` (dense [N,(H+1)//2,(W+1)//2,C], y's dtype) / ``pool_codes`` (uint8, same shape): the 2x2 ceil-mode max pooling of y
    (and its routing codes) in the same call -- `maxpool_fwd(y, pool_out, codes=pool_codes)` bit for bit; ``pool_only``: the caller
    reads the pooled tensor alone -- y may be left unwritten.
    ``stem`` = (x, dw, db, accumulate): the UNet stem's weight / bias gradient from this data gradient's output tile (dct_conv_desc.stem_*);
    y is then NOT written.  Raises ``StemFusionUnsupported`` when the layer does not take the kernel that can."""
    if pool_out is not None:
        want = (y.shape[0], (y.shape[1] + 1) // 2, (y.shape[2] + 1) // 2, y.shape[3])
        assert tuple(pool_out.shape) == want and pool_out.is_contiguous() and pool_out.dtype == y.dtype, "pool_out: dense ceil-mode half of y"
        assert pool_codes is None or (pool_codes.dtype == torch.uint8 and pool_codes.is_contiguous() and tuple(pool_codes.shape) == want)
        assert not accumulate and not scatter2x2 and mask is None and mask_bits is None, "pool_out is a forward-pass feature"
    else:
        assert pool_codes is None and not pool_only
    if unpool is not None:
        _check_unpool(unpool, x)
    d = conv_desc(R, S, stride, dil, pad_h, pad_w, relu, scatter2x2, accumulate, mask_channels, mask_scale,
                  _bits_ok(mask_bits, mask), _bits_ok(relu_bits_out, y), pool_out, pool_codes, pool_only, stem, unpool)
    if stem is not None:
        sx, sdw, sdb, _ = stem
        assert sx.dtype == torch.float32 and sx.is_contiguous() and sx.numel() == y.shape[0] * (y.shape[1] + 2) * (y.shape[2] + 2)
        assert sdw.dtype == torch.float32 and sdw.numel() == 64 * 9 and sdb.dtype == torch.float32 and sdb.numel() == 64
    vx, vy = view(x), view(y)
    lib = _lib.load()
    dt = _dt(x)
    need = lib.dct_conv2d_workspace_bytes(C.byref(vx), C.byref(vy), C.byref(d), dt)
    ws = _ws(need, x.device)
    vm = view(mask) if mask is not None else None
    if stem is not None:
        rc = getattr(lib, "dct_conv2d")(C.byref(vx), ptr(w_packed), ptr(bias), C.byref(vm) if vm is not None else None,
                                        C.byref(vy), C.byref(d), dt, ptr(ws), ws.numel(), stream())
        if rc == _lib.ERR_UNSUPPORTED:
            raise StemFusionUnsupported()
        _lib.check(rc, "dct_conv2d")
        return y
    if unpool is not None:
        rc = getattr(lib, "dct_conv2d")(C.byref(vx), ptr(w_packed), ptr(bias), C.byref(vm) if vm is not None else None,
                                        C.byref(vy), C.byref(d), dt, ptr(ws), ws.numel(), stream())
        if rc == _lib.ERR_UNSUPPORTED:
            raise UnpoolOnLoadUnsupported()
        _lib.check(rc, "dct_conv2d")
        return y
    call("dct_conv2d", C.byref(vx), ptr(w_packed), ptr(bias), C.byref(vm) if vm is not None else None,
         C.byref(vy), C.byref(d), dt, ptr(ws), ws.numel(), stream())
    return y


def conv2d_wgrad(p, q, dw, *, R=3, S=3, stride=1, dil=1, pad_h=0, pad_w=0, accumulate=False, db=None, unpool=None):
    """dw[p.c][R][S][q.c] (fp32, dense) (+)= sum_m p[m] (x) q[shifted m]; with ``db`` (bf16 only) also
    db[p.c] (+)= sum_m p[m] in the same launch.  ``unpool`` = (codes, H, W): p is the gradient at a max-pooled tensor, expanded to its
    un-pooled [N,H,W,C] form while the kernel stages (dct_conv_desc.unpool_*); ``UnpoolOnLoadUnsupported`` when the layer's kernel cannot."""
    if unpool is not None:
        _check_unpool(unpool, p)
    d = conv_desc(R, S, stride, dil, pad_h, pad_w, accumulate=accumulate, unpool=unpool)
    vp, vq = view(p), view(q)
    lib = _lib.load()
    dt = _dt(p)
    need = lib.dct_conv2d_wgrad_workspace_bytes(C.byref(vp), C.byref(vq), C.byref(d), dt)
    if need == 0:
        if unpool is not None:
            raise UnpoolOnLoadUnsupported()
        raise RuntimeError("dct_amd: conv2d_wgrad unsupported shape")
    ws = _ws(need, p.device)
    name = "dct_conv2d_wgrad_bias" if db is not None else "dct_conv2d_wgrad"
    args = (C.byref(vp), C.byref(vq), ptr(dw)) + ((ptr(db),) if db is not None else ()) + (C.byref(d), dt, ptr(ws), ws.numel(), stream())
    if unpool is not None:
        rc = getattr(lib, name)(*args)
        if rc == _lib.ERR_UNSUPPORTED:
            raise UnpoolOnLoadUnsupported()
        _lib.check(rc, name)
        return dw
    call(name, *args)
    return dw


def bias_grad(dy, db, accumulate=False):
    vd = view(dy)
    need = _lib.load().dct_bias_grad_workspace_bytes(C.byref(vd))
    ws = _ws(need, dy.device)
    call("dct_bias_grad", C.byref(vd), ptr(db), int(accumulate), _dt(dy), ptr(ws), ws.numel(), stream())
    return db


def bias_grad_batched(dys, dbs, accumulate=False):
    """db_k (+)= column sums of dy_k for up to eight (dy, db) pairs of one dtype in ONE launch pair (dct_bias_grad_batched)."""
    n = len(dys)
    if n == 1:
        return bias_grad(dys[0], dbs[0], accumulate)
    assert 1 < n <= 8 and len(dbs) == n and all(d.dtype == dys[0].dtype for d in dys)
    views = (_lib.View * n)(*[view(d) for d in dys])
    ptrs = (C.c_void_p * n)(*[int(b.data_ptr()) for b in dbs])
    need = _lib.load().dct_bias_grad_batched_workspace_bytes(views, n)
    ws = _ws(need, dys[0].device)
    call("dct_bias_grad_batched", views, ptrs, n, int(accumulate), _dt(dys[0]), ptr(ws), ws.numel(), stream())


def pack_weight(src_f32, dst, P, T, Q, transpose=False, flip_taps=False):
    call("dct_pack_weight", ptr(src_f32), ptr(dst), P, T, Q, int(transpose), int(flip_taps), _dt(dst), stream())
    return dst


def pack_jobs_table(jobs, device):
    """Device table for pack_weights_batched.  jobs: list of (src fp32 or bf16 tensor, dst tensor, P, T, Q, mode, flip) with
    mode 1 -> dst[Q][T'][P], mode 2 -> dst[T'][Q][P] (dct_pack_weight's transpose codes).  Returns (table, n, tiles, edge):
    edge 64 when every job is 16-bit -> 16-bit with P, Q multiples of 64 (the vectorised kernel), else 32."""
    import struct
    wide = all(src.dtype == dst.dtype and src.element_size() == 2 and P % 64 == 0 and Q % 64 == 0 and
               src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0 for src, dst, P, T, Q, mode, flip in jobs)
    edge = 64 if wide else 32
    buf, tiles = bytearray(), 0
    for src, dst, P, T, Q, mode, flip in jobs:
        assert P % 32 == 0 and Q % 32 == 0 and mode in (1, 2)
        dq, dt = (T * P, P) if mode == 1 else (P, Q * P)
        assert src.dtype in (torch.float32, torch.bfloat16)
        buf += struct.pack("<QQiiiiqqii", src.data_ptr(), dst.data_ptr(), P, T, Q, int(bool(flip)), dq, dt, tiles,
                           int(src.dtype == torch.bfloat16))
        tiles += (P // edge) * (Q // edge) * T
    table = torch.frombuffer(buf, dtype=torch.uint8).clone().to(device)
    return table, len(jobs), tiles, edge


def pack_weights_batched(table, njobs, tiles, dtype, edge=32):
    if edge == 64:
        call("dct_pack_weights_batched64", ptr(table), int(njobs), int(tiles), stream())
    else:
        call("dct_pack_weights_batched", ptr(table), int(njobs), int(tiles), DTYPE_OF[dtype], stream())


def bn_running_table(layers, device):
    """Device record table for bn_running_update.  layers: list of (running_mean, running_var, num_batches_tracked or None, c,
    mean_off, var_off) -- offsets in elements into the flat statistics buffer of a forward pass."""
    import struct
    buf = bytearray()
    for rm, rv, nbt, c, moff, voff in layers:
        assert rm.dtype == torch.float32 and rv.dtype == torch.float32 and rm.is_contiguous() and rv.is_contiguous()
        assert nbt is None or nbt.dtype == torch.int64
        buf += struct.pack("<QQQiiii", rm.data_ptr(), rv.data_ptr(), nbt.data_ptr() if nbt is not None else 0, int(c), int(moff), int(voff), 0)
    return torch.frombuffer(buf, dtype=torch.uint8).clone().to(device)


def bn_running_update(table, n_layers, stats, momentum):
    call("dct_bn_running_update", ptr(table), int(n_layers), ptr(stats), float(momentum), stream())


def flat_sum(out, a, b, c=None):
    """out = a + b (+ c): flat fp32 buffers of equal length (multiple of 4 elements)."""
    call("dct_flat_sum", ptr(out), ptr(a), ptr(b), ptr(c), out.numel(), stream())
    return out


def flat_scale(x, scale):
    """x *= scale in place: a dense fp32 buffer (or slice of one) of any length."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    if x.numel():
        call("dct_flat_scale", ptr(x), float(scale), x.numel(), stream())
    return x


def conv_cin1_fwd(x, w, bias, y, *, R=3, S=3, stride=1, dil=1, pad_h=0, pad_w=0, relu=False, relu_bits_out=None):
    d = conv_desc(R, S, stride, dil, pad_h, pad_w, relu, relu_bits_out=_bits_ok(relu_bits_out, y))
    vx, vy = view(x), view(y)
    call("dct_conv_cin1_fwd", C.byref(vx), ptr(w), ptr(bias), C.byref(vy), C.byref(d), _dt(y), stream())
    return y


def conv_cin1_dgrad(dy, w, dx, *, R=3, S=3, stride=1, dil=1, pad_h=0, pad_w=0):
    d = conv_desc(R, S, stride, dil, pad_h, pad_w)
    vdy, vdx = view(dy), view(dx)
    call("dct_conv_cin1_dgrad", C.byref(vdy), ptr(w), C.byref(vdx), C.byref(d), _dt(dy), stream())
    return dx


def conv_cin1_wgrad(x, dy, dw, db, *, R=3, S=3, stride=1, dil=1, pad_h=0, pad_w=0, accumulate=False):
    d = conv_desc(R, S, stride, dil, pad_h, pad_w)
    vx, vdy = view(x), view(dy)
    need = _lib.load().dct_conv_cin1_wgrad_workspace_bytes(C.byref(vdy), C.byref(d))
    ws = _ws(need, x.device)
    call("dct_conv_cin1_wgrad", C.byref(vx), C.byref(vdy), ptr(dw), ptr(db), C.byref(d), int(accumulate), _dt(dy),
         ptr(ws), ws.numel(), stream())


def head_fwd(x, w, bias, y):
    vx, vy = view(x), view(y)
    call("dct_conv1x1_head_fwd", C.byref(vx), ptr(w), ptr(bias), C.byref(vy), _dt(x), stream())
    return y


def head_bwd(x, dy, w, dx, dw, db, relu_mask=True, accumulate=False):
    vx, vdy = view(x), view(dy)
    vdx = view(dx) if dx is not None else None
    need = _lib.load().dct_conv1x1_head_bwd_workspace_bytes(C.byref(vx), dy.shape[3])
    ws = _ws(need, x.device)
    call("dct_conv1x1_head_bwd", C.byref(vx), C.byref(vdy), ptr(w), C.byref(vdx) if vdx is not None else None,
         ptr(dw), ptr(db), int(relu_mask), int(accumulate), _dt(x), ptr(ws), ws.numel(), stream())


# ------------------------------------------------------------------------------ pointwise
def maxpool_fwd(x, y, codes=None):
    """``codes`` (uint8, dense, y's shape): also keep the routing decision (window position of the first maximum + its ReLU
    gate) for `maxpool_bwd(codes=...)`, which then does not re-read x."""
    vx, vy = view(x), view(y)
    if codes is not None:
        assert codes.dtype == torch.uint8 and codes.is_contiguous() and tuple(codes.shape) == tuple(y.shape)
        call("dct_maxpool2x2_fwd_codes", C.byref(vx), C.byref(vy), ptr(codes), _dt(x), stream())
    else:
        call("dct_maxpool2x2_fwd", C.byref(vx), C.byref(vy), _dt(x), stream())
    return y


def maxpool_bwd(x, dy, dx, relu_mask=False, scale=1.0, codes=None, skip=None):
    """``skip`` (with codes): the gradient at a bilinearly resized copy of the pooled tensor; its bilinear backward is added to dy on the way."""
    vdy, vdx = view(dy), view(dx)
    if codes is not None and skip is not None:
        assert codes.dtype == torch.uint8 and codes.is_contiguous() and tuple(codes.shape) == tuple(dy.shape)
        vs = view(skip)
        rc = _lib.load().dct_maxpool2x2_bwd_codes_skip(ptr(codes), C.byref(vdy), C.byref(vs), C.byref(vdx), int(relu_mask), float(scale),
                                                       _dt(dy), stream())
        if rc == _lib.ERR_UNSUPPORTED:
            # views the 16-byte kernel cannot take (channel counts / slice offsets off the vector width): the sum formed in memory,
            # as before the fusion -- bilinear backward (which has a scalar form) into a buffer, + dy, then the plain un-pooling
            total = bilinear_bwd(skip, torch.empty_like(dy))
            total += dy
            vt = view(total)
            call("dct_maxpool2x2_bwd_codes", ptr(codes), C.byref(vt), C.byref(vdx), int(relu_mask), float(scale), _dt(dy), stream())
            return dx
        _lib.check(rc, "dct_maxpool2x2_bwd_codes_skip")
        return dx
    assert skip is None
    if codes is not None:
        assert codes.dtype == torch.uint8 and codes.is_contiguous() and tuple(codes.shape) == tuple(dy.shape)
        call("dct_maxpool2x2_bwd_codes", ptr(codes), C.byref(vdy), C.byref(vdx), int(relu_mask), float(scale), _dt(dy), stream())
        return dx
    vx = view(x)
    call("dct_maxpool2x2_bwd", C.byref(vx), C.byref(vdy), C.byref(vdx), int(relu_mask), float(scale), _dt(x), stream())
    return dx


def bilinear_fwd(x, y):
    vx, vy = view(x), view(y)
    call("dct_bilinear_fwd", C.byref(vx), C.byref(vy), _dt(x), _dt(y), stream())
    return y


def bilinear_fwd_batched(xs, ys):
    """ys[k] = bilinear_fwd(xs[k]) for up to eight pairs of one dtype in ONE launch (dct_bilinear_fwd_batched); views off the 16-byte vector width
    fall back to one launch per tensor."""
    n = len(xs)
    assert n == len(ys) and 1 <= n <= 8 and all(_dt(x) == _dt(xs[0]) == _dt(y) for x, y in zip(xs, ys))
    vx = (_lib.View * n)(*[view(x) for x in xs])
    vy = (_lib.View * n)(*[view(y) for y in ys])
    rc = _lib.load().dct_bilinear_fwd_batched(vx, vy, n, _dt(xs[0]), stream())
    if rc == _lib.ERR_UNSUPPORTED:
        for x, y in zip(xs, ys):
            bilinear_fwd(x, y)
        return ys
    _lib.check(rc, "dct_bilinear_fwd_batched")
    return ys


def bilinear_bwd(dy, dx, accumulate=False):
    vdy, vdx = view(dy), view(dx)
    call("dct_bilinear_bwd", C.byref(vdy), C.byref(vdx), _dt(dy), _dt(dx), int(accumulate), stream())
    return dx


def dropout_fwd(x, y, p, seed, offset, mask_out=None, calls_dev=None, parity=0):
    """``calls_dev`` (int64[2] device tensor): the call counter lives on the device -- the launch reads word ``parity``, is call number that + 1
    (offset = number << 40; ``offset`` is ignored) and stores the number to the other word; the caller alternates ``parity`` from call to call.
    The graph-replayable form."""
    vx, vy = view(x), view(y)
    if calls_dev is not None:
        assert calls_dev.dtype == torch.int64 and calls_dev.numel() >= 2
        call("dct_dropout_fwd_dev", C.byref(vx), C.byref(vy), ptr(mask_out), float(p), int(seed), ptr(calls_dev), int(parity), _dt(x), stream())
        return y
    call("dct_dropout_fwd", C.byref(vx), C.byref(vy), ptr(mask_out), float(p), int(seed), int(offset), _dt(x), stream())
    return y


def dropout_maxpool_fwd(x, y, codes, p, seed, calls_dev, parity):
    """y, codes = maxpool_fwd(dropout_fwd(x, calls_dev=..., parity=...), codes=...) in one pass, without the dropped tensor
    (dct_dropout_maxpool2x2_fwd_codes; bit for bit the two launches)."""
    assert calls_dev.dtype == torch.int64 and calls_dev.numel() >= 2
    assert codes.dtype == torch.uint8 and codes.is_contiguous() and tuple(codes.shape) == tuple(y.shape)
    vx, vy = view(x), view(y)
    call("dct_dropout_maxpool2x2_fwd_codes", C.byref(vx), C.byref(vy), ptr(codes), float(p), int(seed), ptr(calls_dev), int(parity), _dt(x), stream())
    return y


def dropout_apply(x, y, mask_u8, p):
    vx, vy = view(x), view(y)
    call("dct_dropout_apply", C.byref(vx), C.byref(vy), ptr(mask_u8), float(p), _dt(x), stream())
    return y


def relu_bwd(g, a, y, scale=1.0):
    vg, va, vy = view(g), view(a), view(y)
    call("dct_relu_bwd", C.byref(vg), C.byref(va), C.byref(vy), float(scale), _dt(g), stream())
    return y


def cast(x, y):
    vx, vy = view(x), view(y)
    call("dct_cast", C.byref(vx), C.byref(vy), _dt(x), _dt(y), stream())
    return y


# ------------------------------------------------------------------------------ losses etc.
def _loss_ws(device):
    return torch.empty(_lib.load().dct_loss_workspace_bytes(0), dtype=torch.uint8, device=device)


def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(ts))(*[ptr(t) for t in ts])
    return arr


def ce_fwd(logits_pc, targets, C_, ignore_index=255):
    """logits_pc: fp32 [P, C] dense; targets int64 [P].  Returns device tensor [2] = (mean loss, count)."""
    out = torch.empty(2, dtype=torch.float32, device=logits_pc.device)
    ws = _loss_ws(logits_pc.device)
    call("dct_ce_fwd", ptr(logits_pc), ptr(targets), logits_pc.numel() // C_, C_, int(ignore_index), ptr(out),
         ptr(ws), ws.numel(), stream())
    return out


def ce_bwd(logits_pc, targets, C_, count, dlogits, gscale=None, gmul=1.0, ignore_index=255, accumulate=False):
    call("dct_ce_bwd", ptr(logits_pc), ptr(targets), logits_pc.numel() // C_, C_, int(ignore_index), ptr(count),
         ptr(gscale), float(gmul), ptr(dlogits), int(accumulate), stream())
    return dlogits


def ce_step(logits_pc, targets, C_, dlogits, gscale=None, gmul=1.0, ignore_index=255, accumulate=False):
    """ce_fwd + ce_bwd of the same logits in two launches instead of three (dct_ce_step; bit for bit the two calls).  Returns the device
    tensor [2] = (mean loss, count); ``dlogits`` is written (added to)."""
    out = torch.empty(2, dtype=torch.float32, device=logits_pc.device)
    ws = _loss_ws(logits_pc.device)
    call("dct_ce_step", ptr(logits_pc), ptr(targets), logits_pc.numel() // C_, C_, int(ignore_index), ptr(out), ptr(gscale), float(gmul),
         ptr(dlogits), int(accumulate), ptr(ws), ws.numel(), stream())
    return out


def softmax_fwd(logits_pc, C_):
    probs = torch.empty_like(logits_pc)
    call("dct_softmax_fwd", ptr(logits_pc), ptr(probs), logits_pc.numel() // C_, C_, stream())
    return probs


def softmax_bwd(probs, dprobs, C_, dlogits=None, accumulate=False):
    if dlogits is None:
        dlogits = torch.empty_like(probs)
    call("dct_softmax_bwd", ptr(probs), ptr(dprobs), ptr(dlogits), probs.numel() // C_, C_, int(accumulate), stream())
    return dlogits


def entropy_fwd(probs, C_):
    out = torch.empty(probs.numel() // C_, dtype=torch.float32, device=probs.device)
    call("dct_entropy_fwd", ptr(probs), ptr(out), probs.numel() // C_, C_, stream())
    return out


def entropy_bwd(probs, dmap, C_):
    out = torch.empty_like(probs)
    call("dct_entropy_bwd", ptr(probs), ptr(dmap), ptr(out), probs.numel() // C_, C_, stream())
    return out


def jsd_map_fwd(probs: List[torch.Tensor], C_):
    P = probs[0].numel() // C_
    out = torch.empty(P, dtype=torch.float32, device=probs[0].device)
    call("dct_jsd_map_fwd", _ptr_array(probs), len(probs), ptr(out), P, C_, stream())
    return out


def jsd_map_bwd(probs: List[torch.Tensor], dmap, C_):
    outs = [torch.empty_like(p) for p in probs]
    call("dct_jsd_map_bwd", _ptr_array(probs), len(probs), ptr(dmap), _ptr_array(outs), probs[0].numel() // C_, C_, stream())
    return outs


def kl_map_fwd(p, y, C_, eps=1e-10):
    out = torch.empty(p.numel() // C_, dtype=torch.float32, device=p.device)
    call("dct_kl_map_fwd", ptr(p), ptr(y), ptr(out), p.numel() // C_, C_, float(eps), stream())
    return out


def kl_map_bwd(p, y, dmap, C_, eps=1e-10):
    dp = torch.empty_like(p)
    call("dct_kl_map_bwd", ptr(p), ptr(y), ptr(dmap), ptr(dp), p.numel() // C_, C_, float(eps), stream())
    return dp


def jsd_logits_fwd(logits: List[torch.Tensor], C_):
    out = torch.empty(1, dtype=torch.float32, device=logits[0].device)
    ws = _loss_ws(out.device)
    call("dct_jsd_logits_fwd", _ptr_array(logits), len(logits), logits[0].numel() // C_, C_, ptr(out), ptr(ws), ws.numel(), stream())
    return out


def jsd_logits_bwd(logits: List[torch.Tensor], C_, dlogits: List[torch.Tensor], gscale=None, gmul=1.0, accumulate=False):
    call("dct_jsd_logits_bwd", _ptr_array(logits), len(logits), logits[0].numel() // C_, C_, ptr(gscale), float(gmul),
         _ptr_array(dlogits), int(accumulate), stream())
    return dlogits


def jsd_logits_step(logits: List[torch.Tensor], C_, dlogits: List[torch.Tensor] = None, want_probs: bool = True, gscale=None, gmul=1.0,
                    accumulate=False):
    """-> (mean JSD [1], [softmax(logits_s)] or None): `jsd_logits_fwd`, the S `softmax_fwd` maps and `jsd_logits_bwd` into ``dlogits`` (None:
    no gradients) as ONE pass over the logits + the finalize -- bit for bit what the separate launches give (dct_jsd_logits_step)."""
    dev = logits[0].device
    out = torch.empty(1, dtype=torch.float32, device=dev)
    ws = _loss_ws(dev)
    probs = [torch.empty_like(lp) for lp in logits] if want_probs else None
    call("dct_jsd_logits_step", _ptr_array(logits), len(logits), logits[0].numel() // C_, C_, ptr(out),
         _ptr_array(probs) if probs is not None else None, ptr(gscale), float(gmul),
         _ptr_array(dlogits) if dlogits is not None else None, int(accumulate), ptr(ws), ws.numel(), stream())
    return out, probs


def kl_logits_fwd(p_logits, y_logits, C_, eps=1e-10):
    out = torch.empty(1, dtype=torch.float32, device=p_logits.device)
    ws = _loss_ws(out.device)
    call("dct_kl_logits_fwd", ptr(p_logits), ptr(y_logits), p_logits.numel() // C_, C_, float(eps), ptr(out), ptr(ws), ws.numel(), stream())
    return out


def kl_logits_bwd(p_logits, y_logits, C_, dp, gscale=None, gmul=1.0, eps=1e-10, accumulate=False):
    call("dct_kl_logits_bwd", ptr(p_logits), ptr(y_logits), p_logits.numel() // C_, C_, float(eps), ptr(gscale), float(gmul),
         ptr(dp), int(accumulate), stream())
    return dp


def argmax(x_pc, C_):
    out = torch.empty(x_pc.numel() // C_, dtype=torch.int64, device=x_pc.device)
    call("dct_argmax", ptr(x_pc), ptr(out), x_pc.numel() // C_, C_, stream())
    return out


def fgsm_step(x, g, eps, want_noise=True):
    xa = torch.empty_like(x)
    noise = torch.empty_like(x) if want_noise else None
    call("dct_fgsm_step", ptr(x), ptr(g), float(eps), ptr(xa), ptr(noise), x.numel(), stream())
    return xa, noise


def adam_flat(p, g, m, v, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, bf16_shadow=None, grad_scale=1.0):
    call("dct_adam_flat", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(step_size), float(bc2_sqrt), float(beta1),
         float(beta2), float(eps), float(weight_decay), float(grad_scale), ptr(bf16_shadow), stream())


def adam_flat_dev(p, g, m, v, state, table, beta1, beta2, eps, weight_decay, bf16_shadow=None, grad_scale=1.0):
    """Adam with {step count, lr, table base, table length} in the float64[4] device tensor ``state`` and the
    host-computed {step_size, bc2_sqrt} pairs in the float32 device tensor ``table`` (dct_adam_flat_dev)."""
    call("dct_adam_flat_dev", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(state), ptr(table), float(beta1), float(beta2),
         float(eps), float(weight_decay), float(grad_scale), ptr(bf16_shadow), stream())
    return p


def dice_counts(logits_bpc, gt_bp, B, C_):
    """logits [B, pix, C] fp32 dense, gt [B, pix] int64 -> (inter, psum, gsum) int32 [B, C]."""
    dev = logits_bpc.device
    cnt = torch.zeros(3, B, C_, dtype=torch.int32, device=dev)
    ppi = logits_bpc.numel() // (B * C_)
    call("dct_dice_counts", ptr(logits_bpc), ptr(gt_bp), B, ppi, C_, ptr(cnt[0]), ptr(cnt[1]), ptr(cnt[2]), stream())
    return cnt[0], cnt[1], cnt[2]


def dice_update(inter, psum, gsum, method3d, axes_mask, smooth, acc):
    """Dice rows of one DiceMeter.add + the meter's running moments (acc: float64 [2, C+1]) in one launch."""
    B, C_ = inter.shape
    rows = 1 if method3d else B
    dice = torch.empty(rows, C_, dtype=torch.float32, device=inter.device)
    call("dct_dice_update", ptr(inter), ptr(psum), ptr(gsum), B, C_, int(method3d), int(axes_mask), float(smooth), ptr(dice), ptr(acc),
         stream())
    return dice


# ------------------------------------------------------------------------------ Enet family
class Tf(object):
    """Producer transform act(scale*x + shift) applied by consumers on load (include/dct.h dct_enet_tf)."""
    NONE, AFFINE, PRELU, RELU = 0, 1, 2, 3

    def __init__(self, scale, shift, slope=None, mode=1):
        self.scale, self.shift, self.slope, self.mode = scale, shift, slope, mode

    def c(self):
        return EnetTf(ptr(self.scale), ptr(self.shift), ptr(self.slope), int(self.mode))


def _tfp(tf):
    if tf is None:
        return None, None
    s = tf.c()
    return s, C.byref(s)


_red_ws = {}


def _enet_ws(device, nbytes):
    """Reduction scratch, cached per (device, stream): kernels of two models queued on different streams must not
    share it.  While a HIP graph is being captured the buffer comes from the graph's own pool instead."""
    if _group() is not None:        # members of a pass group run at the same time: scratch of their own, kept until the launches
        return keep(torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device))
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _red_ws.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _red_ws[key] = t
    return t


def _mixed(*ts):
    """(dtype code T, f32 mask) of a call whose view arguments are ``ts`` (None = absent): T is bf16 / f16 when any
    view is, and bit k of the mask marks view k as fp32 storage."""
    low = [t.dtype for t in ts if t is not None and t.dtype in (torch.bfloat16, torch.float16)]
    if len(set(low)) > 1:
        raise RuntimeError("dct_amd: bf16 and f16 views in one Enet call")
    mask = 0
    for k, t in enumerate(ts):
        if t is not None and t.dtype == torch.float32:
            mask |= 1 << k
        elif t is not None and t.dtype not in (torch.bfloat16, torch.float16):
            raise RuntimeError(f"dct_amd: unsupported dtype {t.dtype}")
    return (DTYPE_OF[low[0]] if low else F32), mask


def enet_conv(x, w, bias, tf, y, *, R, S, stride=1, dil=1, pad_h=0, pad_w=0, transposed=False, ws=(0, 0, 0),
              resid_grad=None, resid_mask=None, accumulate=False, compute=None):
    """``compute``: the net's compute dtype.  A forward conv reads a raw fp32 tensor and writes one, so no view says which mode
    the call belongs to; in bf16 / f16 mode the contraction's operands are rounded to that type (MFMA form, csrc/enet.hip)."""
    d = conv_desc(R, S, stride, dil, pad_h, pad_w, accumulate=accumulate)
    vx, vy = view(x), view(y)
    keep, tfp = _tfp(tf)
    vrg = view(resid_grad) if resid_grad is not None else None
    vrm = view(resid_mask) if resid_mask is not None else None
    dt, fm = _mixed(x, y, resid_grad, resid_mask)
    if dt == F32 and compute in (torch.bfloat16, torch.float16):
        dt = DTYPE_OF[compute]
    call("dct_enet_conv", C.byref(vx), ptr(w), ptr(bias), tfp, C.byref(vy), C.byref(d), int(transposed),
         int(ws[0]), int(ws[1]), int(ws[2]), C.byref(vrg) if vrg is not None else None,
         C.byref(vrm) if vrm is not None else None, fm, dt, stream())
    return y


def enet_conv_stats(x, w, bias, tf, y, stats, *, R, S, stride=1, dil=1, pad_h=0, pad_w=0, transposed=False, ws=(0, 0, 0), compute=None):
    """enet_conv with the consumer BatchNorm's partial sums written by the convolution's epilogue into ``stats`` (float64,
    >= tiles * C * 3 elements, tiles = ceil(pixels / 32)).  -> number of partial rows written, 0 when the call did not take the
    MFMA form (the statistics then need the usual reduction)."""
    d = conv_desc(R, S, stride, dil, pad_h, pad_w)
    vx, vy = view(x), view(y)
    keep, tfp = _tfp(tf)
    dt, fm = _mixed(x, y)
    if dt == F32 and compute in (torch.bfloat16, torch.float16):
        dt = DTYPE_OF[compute]
    rows = C.c_int(0)
    cap = stats.numel() // (3 * y.shape[3])
    call("dct_enet_conv_stats", C.byref(vx), ptr(w), ptr(bias), tfp, C.byref(vy), C.byref(d), int(transposed),
         int(ws[0]), int(ws[1]), int(ws[2]), fm, dt, ptr(stats), int(cap), C.byref(rows), stream())
    return int(rows.value)


def enet_bn_fwd_stats(raw, gamma, beta, eps, momentum, running_mean, running_var, training, scale, shift, mean, invstd,
                      dtype_hint=None, save_var=None, partial=None, partial_rows=0):
    vr = view(raw)
    dt, fm = _mixed(raw)
    if partial is not None and partial_rows > 0:          # rows written by enet_conv_stats: fold only
        call("dct_enet_bn_fwd_stats_rows", C.byref(vr), ptr(gamma), ptr(beta), float(eps), float(momentum), ptr(running_mean),
             ptr(running_var), int(training), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(save_var), fm, dt, ptr(partial),
             partial.numel() * partial.element_size(), int(partial_rows), stream())
        return
    ws = _enet_ws(raw.device, _lib.load().dct_enet_reduce_workspace_bytes(raw.shape[3]))
    call("dct_enet_bn_fwd_stats", C.byref(vr), ptr(gamma), ptr(beta), float(eps), float(momentum), ptr(running_mean),
         ptr(running_var), int(training), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(save_var), fm, dt, ptr(ws), ws.numel(),
         stream())


def enet_conv_bnbwd_stats(x, w, y, stats, rec_raw, rec_tf, rec_mean, rec_invstd, *, R, S, stride=1, dil=1, pad_h=0, pad_w=0,
                          transposed=False, ws=(0, 0, 0), compute=None):
    """Data-gradient convolution y = dgrad(x) (enet_conv without bias / transform / residual) whose epilogue also writes the
    BatchNorm-backward partial sums of the layer that produced y's tensor (raw output ``rec_raw``, consumer transform ``rec_tf``,
    saved statistics) into ``stats`` (float64, >= tiles * C * 3).  -> partial rows written (0: the usual reduction is needed)."""
    d = conv_desc(R, S, stride, dil, pad_h, pad_w)
    vx, vy, vr = view(x), view(y), view(rec_raw)
    dt, fm = _mixed(x, y)
    if dt == F32 and compute in (torch.bfloat16, torch.float16):
        dt = DTYPE_OF[compute]
    act = rec_tf.mode if rec_tf.mode in (2, 3) else 0
    rows = C.c_int(0)
    cap = stats.numel() // (3 * y.shape[3])
    call("dct_enet_conv_bnbwd_stats", C.byref(vx), ptr(w), C.byref(vy), C.byref(d), int(transposed), int(ws[0]), int(ws[1]), int(ws[2]),
         fm, dt, C.byref(vr), ptr(rec_tf.scale), ptr(rec_tf.shift), ptr(rec_tf.slope), int(act), ptr(rec_mean), ptr(rec_invstd),
         ptr(stats), int(cap), C.byref(rows), stream())
    return int(rows.value)


def enet_bn_bwd(raw, g, g_mask, tf, mean, invstd, dgamma, dbeta, dslope, c1c2, draw, training=True, partial=None, partial_rows=0):
    vr, vg, vd = view(raw), view(g), view(draw)
    vm = view(g_mask) if g_mask is not None else None
    act = tf.mode if tf.mode in (2, 3) else 0
    dt, fm = _mixed(raw, g, g_mask, draw)
    if partial is not None and partial_rows != 0:         # rows written by enet_conv_bnbwd_stats: fold + apply only
        call("dct_enet_bn_bwd_rows", C.byref(vr), C.byref(vg), None,
             ptr(tf.scale), ptr(tf.shift), ptr(tf.slope), int(act), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), ptr(dslope),
             ptr(c1c2), int(training), C.byref(vd), fm, dt, ptr(partial), partial.numel() * partial.element_size(), int(partial_rows),
             stream())
        return draw
    ws = _enet_ws(raw.device, _lib.load().dct_enet_reduce_workspace_bytes(raw.shape[3]))
    call("dct_enet_bn_bwd", C.byref(vr), C.byref(vg), C.byref(vm) if vm is not None else None,
         ptr(tf.scale), ptr(tf.shift), ptr(tf.slope), int(act), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), ptr(dslope),
         ptr(c1c2), int(training), C.byref(vd), fm, dt, ptr(ws), ws.numel(), stream())
    return draw


def enet_bn_bwd_sums(raw, g, g_mask, tf, mean, invstd, dgamma, dbeta, dslope, c1c2, training=True, partial=None, partial_rows=0):
    """The sums half of `enet_bn_bwd`: parameter gradients (+=) and the apply pass's two means (c1c2)."""
    vr, vg = view(raw), view(g)
    vm = view(g_mask) if g_mask is not None else None
    act = tf.mode if tf.mode in (2, 3) else 0
    dt, fm = _mixed(raw, g, g_mask)
    if partial is not None and partial_rows > 0:
        ws, nbytes = partial, partial.numel() * partial.element_size()
    else:
        ws = _enet_ws(raw.device, _lib.load().dct_enet_reduce_workspace_bytes(raw.shape[3]))
        nbytes, partial_rows = ws.numel(), 0
    call("dct_enet_bn_bwd_sums", C.byref(vr), C.byref(vg), C.byref(vm) if vm is not None else None,
         ptr(tf.scale), ptr(tf.shift), ptr(tf.slope), int(act), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), ptr(dslope),
         ptr(c1c2), int(training), fm, dt, ptr(ws), nbytes, int(partial_rows), stream())


def enet_bn_bwd_apply(raw, g, g_mask, tf, mean, invstd, c1c2, draw, leaf=False):
    """The apply half: draw = BatchNorm-backward of g (elementwise, from raw / g / c1c2).  ``leaf``: only weight gradients read the
    result -- under a `LeafSide` the launch is held back with them."""
    vr, vg, vd = view(raw), view(g), view(draw)
    vm = view(g_mask) if g_mask is not None else None
    act = tf.mode if tf.mode in (2, 3) else 0
    dt, fm = _mixed(raw, g, g_mask, draw)
    call("dct_enet_bn_bwd_apply", C.byref(vr), C.byref(vg), C.byref(vm) if vm is not None else None,
         ptr(tf.scale), ptr(tf.shift), ptr(tf.slope), int(act), ptr(mean), ptr(invstd), ptr(c1c2), C.byref(vd), fm, dt, int(leaf), stream())
    return draw


def enet_conv_bwd_in(raw, w, tf, g, g_mask, mean, invstd, c1c2, y, *, R, S, stride=1, dil=1, pad_h=0, pad_w=0, transposed=False,
                     ws=(0, 0, 0), resid_grad=None, resid_mask=None, accumulate=False, compute=None, bn=None):
    """Data-gradient convolution whose input is the BatchNorm-backward result of the layer (raw, tf, saved mean / invstd) for the
    upstream gradient g [masked by g_mask], computed ON LOAD (include/dct.h dct_enet_conv_bwd_in).  ``bn`` = (stats, rec_raw, rec_tf,
    rec_mean, rec_invstd): also write the output side's BatchNorm-backward partial rows (as enet_conv_bnbwd_stats).
    -> None when the library has no on-load form for this call (the caller materialises the tensor), else the rows written (0 without
    ``bn``)."""
    d = conv_desc(R, S, stride, dil, pad_h, pad_w, accumulate=accumulate)
    vx, vy, vg = view(raw), view(y), view(g)
    vm = view(g_mask) if g_mask is not None else None
    keep_tf, tfp = _tfp(tf)
    vrg = view(resid_grad) if resid_grad is not None else None
    vrm = view(resid_mask) if resid_mask is not None else None
    dt, fm = _mixed(raw, y, resid_grad, resid_mask)
    if dt == F32 and compute in (torch.bfloat16, torch.float16):
        dt = DTYPE_OF[compute]
    b = _lib.EnetBwdIn(C.pointer(vg), C.pointer(vm) if vm is not None else None, ptr(mean), ptr(invstd), ptr(c1c2))
    rows = C.c_int(0)
    if bn is not None:
        stats, rec_raw, rec_tf, rec_mean, rec_invstd = bn
        vr = view(rec_raw)
        act = rec_tf.mode if rec_tf.mode in (2, 3) else 0
        extra = (C.byref(vr), ptr(rec_tf.scale), ptr(rec_tf.shift), ptr(rec_tf.slope), int(act), ptr(rec_mean), ptr(rec_invstd),
                 ptr(stats), int(stats.numel() // (3 * y.shape[3])), C.byref(rows))
    else:
        extra = (None, None, None, None, 0, None, None, None, 0, None)
    rc = getattr(_lib.load(), "dct_enet_conv_bwd_in")(
        C.byref(vx), ptr(w), tfp, C.byref(b), C.byref(vy), C.byref(d), int(transposed), int(ws[0]), int(ws[1]), int(ws[2]),
        C.byref(vrg) if vrg is not None else None, C.byref(vrm) if vrm is not None else None, fm, dt, *extra, stream())
    if rc == -2:            # DCT_ERR_UNSUPPORTED: no MFMA form for this call
        return None
    _lib.check(rc, "dct_enet_conv_bwd_in")
    return int(rows.value)


def bn_fwd(raw, gamma, beta, eps, momentum, running_mean, running_var, training, scale, shift, mean, invstd, y=None, relu=True):
    """Wide-channel BatchNorm2d (+ReLU) forward of unet_bn: statistics -> scale/shift (+ running statistics) -> y."""
    vr = view(raw)
    vy = view(y) if y is not None else None
    ws = _enet_ws(raw.device, _lib.load().dct_bn_workspace_bytes(raw.shape[3]))
    call("dct_bn_fwd", C.byref(vr), ptr(gamma), ptr(beta), float(eps), float(momentum), ptr(running_mean), ptr(running_var),
         int(training), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), C.byref(vy) if vy is not None else None, int(relu),
         _dt(raw), ptr(ws), ws.numel(), stream())
    return y


def bn_bwd(raw, g, scale, shift, mean, invstd, dgamma, dbeta, c1c2, draw, training=True, relu=True, accumulate=True):
    vr, vg, vd = view(raw), view(g), view(draw)
    ws = _enet_ws(raw.device, _lib.load().dct_bn_workspace_bytes(raw.shape[3]))
    call("dct_bn_bwd", C.byref(vr), C.byref(vg), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta),
         int(accumulate), ptr(c1c2), int(training), int(relu), C.byref(vd), _dt(raw), ptr(ws), ws.numel(), stream())
    return draw


def enet_channel_sum(x, out):
    vx = view(x)
    ws = _enet_ws(x.device, _lib.load().dct_enet_reduce_workspace_bytes(x.shape[3]))
    dt, fm = _mixed(x)
    call("dct_enet_channel_sum", C.byref(vx), ptr(out), fm, dt, ptr(ws), ws.numel(), stream())


def enet_tail_fwd(raw, tf, main_in, rawm, tfm, idx, idx_channels, mode, out):
    vr, vo = view(raw), view(out)
    vmain = view(main_in) if main_in is not None else None
    vrm = view(rawm) if rawm is not None else None
    k1, tfp = _tfp(tf)
    k2, tfmp = _tfp(tfm)
    dt, fm = _mixed(raw, main_in, rawm, out)
    call("dct_enet_tail_fwd", C.byref(vr), tfp, C.byref(vmain) if vmain is not None else None,
         C.byref(vrm) if vrm is not None else None, tfmp, ptr(idx), int(idx_channels), int(mode), C.byref(vo), fm, dt, stream())
    return out


def enet_tail_bwd(dout, out_mask, idx, idx_channels, mode, dst, accumulate=False):
    vd, vm, vdst = view(dout), view(out_mask), view(dst)
    dt, fm = _mixed(dout, out_mask, dst)
    call("dct_enet_tail_bwd", C.byref(vd), C.byref(vm), ptr(idx), int(idx_channels), int(mode), int(accumulate), C.byref(vdst),
         fm, dt, stream())
    return dst


def enet_wgrad(a, tfa, b, tfb, dw, *, R, S, stride=1, dil=1, pad_h=0, pad_w=0):
    d = conv_desc(R, S, stride, dil, pad_h, pad_w)
    va, vb = view(a), view(b)
    need = _lib.load().dct_enet_wgrad_workspace_bytes(C.byref(va), C.byref(vb), C.byref(d))
    ws = _enet_ws(a.device, need)
    k1, tfap = _tfp(tfa)
    k2, tfbp = _tfp(tfb)
    dt, fm = _mixed(a, b)
    call("dct_enet_wgrad", C.byref(va), tfap, C.byref(vb), tfbp, ptr(dw), C.byref(d), fm, dt, ptr(ws), ws.numel(), stream())
    return dw

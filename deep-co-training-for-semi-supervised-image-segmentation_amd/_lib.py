"""ctypes binding of libdct_hip.so (C ABI: include/dct.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``.  It fails
loudly: a missing .so raises at first use, every non-zero status becomes ``RuntimeError``,
and tensors that are not on a HIP device are rejected (no CPU fallback exists).
"""
from __future__ import annotations

import ctypes as C
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCT_LIB_PATH") or os.path.join(_HERE, "libdct_hip.so")      # (override: A/B of two builds in one run)

F32, BF16, F16 = 0, 1, 2
DTYPE_OF = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
TORCH_OF = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
PROF_CLASSES = ("igemm", "wgrad", "pointwise", "loss", "adam", "other")


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32),
                ("sn", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("dil", C.c_int32),
                ("pad_h", C.c_int32), ("pad_w", C.c_int32), ("relu", C.c_int32), ("scatter2x2", C.c_int32),
                ("accumulate", C.c_int32), ("mask_channels", C.c_int32), ("mask_scale", C.c_float),
                ("mask_bits", C.c_void_p), ("relu_bits_out", C.c_void_p), ("pool_out", C.c_void_p), ("pool_codes", C.c_void_p),
                ("pool_only", C.c_int32), ("stem_x", C.c_void_p), ("stem_dw", C.c_void_p), ("stem_db", C.c_void_p),
                ("stem_accumulate", C.c_int32), ("unpool_codes", C.c_void_p), ("unpool_h", C.c_int32), ("unpool_w", C.c_int32)]


class EnetTf(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("slope", C.c_void_p), ("mode", C.c_int32)]


class EnetBwdIn(C.Structure):
    _fields_ = [("g", C.POINTER(View)), ("g_mask", C.POINTER(View)), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("c1c2", C.c_void_p)]


def conv_desc(R=3, S=3, stride=1, dil=1, pad_h=0, pad_w=0, relu=0, scatter2x2=0, accumulate=0,
              mask_channels=0, mask_scale=1.0, mask_bits=None, relu_bits_out=None, pool_out=None, pool_codes=None,
              pool_only=False, stem=None, unpool=None) -> ConvDesc:
    """``stem``: (x fp32 dense [N,H+2,W+2,1], dw [64,9], db [64], accumulate) -- dct_conv_desc.stem_*;
    ``unpool``: (codes uint8 dense [N,(H+1)//2,(W+1)//2,C], H, W) -- dct_conv_desc.unpool_*"""
    sx, sdw, sdb, sacc = stem if stem is not None else (None, None, None, False)
    ucodes, uh, uw = unpool if unpool is not None else (None, 0, 0)
    return ConvDesc(R, S, stride, dil, pad_h, pad_w, int(relu), int(scatter2x2), int(accumulate),
                    int(mask_channels), float(mask_scale),
                    mask_bits.data_ptr() if mask_bits is not None else None,
                    relu_bits_out.data_ptr() if relu_bits_out is not None else None,
                    pool_out.data_ptr() if pool_out is not None else None,
                    pool_codes.data_ptr() if pool_codes is not None else None, int(bool(pool_only)),
                    sx.data_ptr() if sx is not None else None, sdw.data_ptr() if sdw is not None else None,
                    sdb.data_ptr() if sdb is not None else None, int(bool(sacc)),
                    ucodes.data_ptr() if ucodes is not None else None, int(uh), int(uw))


def view(t: torch.Tensor) -> View:
    """View of a physical-NHWC tensor ``t`` of shape [N,H,W,C] (any n/h/w strides, channel stride 1)."""
    if t.dim() != 4 or (t.shape[3] > 1 and t.stride(3) != 1):
        raise ValueError(f"dct view needs [N,H,W,C] with unit channel stride, got {tuple(t.shape)} / {t.stride()}")
    if not t.is_cuda:
        raise RuntimeError("dct_amd: tensors must live on the HIP device (there is no CPU fallback)")
    return View(t.data_ptr(), t.shape[0], t.shape[1], t.shape[2], t.shape[3], t.stride(0), t.stride(1), t.stride(2))


_P = C.c_void_p
_VP = C.POINTER(View)
_DP = C.POINTER(ConvDesc)
_i, _i64, _f, _sz, _u64 = C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint64
_TP = C.POINTER(EnetTf)

# name -> (restype, argtypes); mirrors include/dct.h one to one
SIGNATURES = {
    "dct_version": (_i, []),
    "dct_status_string": (C.c_char_p, [_i]),
    "dct_conv2d_workspace_bytes": (_sz, [_VP, _VP, _DP, _i]),
    "dct_conv2d": (_i, [_VP, _P, _P, _VP, _VP, _DP, _i, _P, _sz, _P]),
    "dct_conv2d_wgrad_workspace_bytes": (_sz, [_VP, _VP, _DP, _i]),
    "dct_conv2d_wgrad": (_i, [_VP, _VP, _P, _DP, _i, _P, _sz, _P]),
    "dct_conv2d_wgrad_bias": (_i, [_VP, _VP, _P, _P, _DP, _i, _P, _sz, _P]),
    "dct_bias_grad": (_i, [_VP, _P, _i, _i, _P, _sz, _P]),
    "dct_bias_grad_workspace_bytes": (_sz, [_VP]),
    "dct_bias_grad_batched": (_i, [_P, _P, _i, _i, _i, _P, _sz, _P]),
    "dct_bias_grad_batched_workspace_bytes": (_sz, [_P, _i]),
    "dct_pack_weight": (_i, [_P, _P, _i, _i, _i, _i, _i, _i, _P]),
    "dct_pack_weights_batched": (_i, [_P, _i, _i, _i, _P]),
    "dct_pack_weights_batched64": (_i, [_P, _i, _i, _P]),
    "dct_bn_running_update": (_i, [_P, _i, _P, C.c_float, _P]),
    "dct_flat_sum": (_i, [_P, _P, _P, _P, C.c_longlong, _P]),
    "dct_flat_scale": (_i, [_P, C.c_float, C.c_longlong, _P]),
    "dct_conv_cin1_fwd": (_i, [_VP, _P, _P, _VP, _DP, _i, _P]),
    "dct_conv_cin1_dgrad": (_i, [_VP, _P, _VP, _DP, _i, _P]),
    "dct_conv_cin1_wgrad_workspace_bytes": (_sz, [_VP, _DP]),
    "dct_conv_cin1_wgrad": (_i, [_VP, _VP, _P, _P, _DP, _i, _i, _P, _sz, _P]),
    "dct_conv1x1_head_fwd": (_i, [_VP, _P, _P, _VP, _i, _P]),
    "dct_conv1x1_head_bwd_workspace_bytes": (_sz, [_VP, _i]),
    "dct_conv1x1_head_bwd": (_i, [_VP, _VP, _P, _VP, _P, _P, _i, _i, _i, _P, _sz, _P]),
    "dct_maxpool2x2_fwd": (_i, [_VP, _VP, _i, _P]),
    "dct_maxpool2x2_bwd": (_i, [_VP, _VP, _VP, _i, _f, _i, _P]),
    "dct_maxpool2x2_fwd_codes": (_i, [_VP, _VP, _P, _i, _P]),
    "dct_maxpool2x2_bwd_codes": (_i, [_P, _VP, _VP, _i, _f, _i, _P]),
    "dct_maxpool2x2_bwd_codes_skip": (_i, [_P, _VP, _VP, _VP, _i, _f, _i, _P]),
    "dct_bilinear_fwd": (_i, [_VP, _VP, _i, _i, _P]),
    "dct_bilinear_fwd_batched": (_i, [_P, _P, _i, _i, _P]),
    "dct_bilinear_bwd": (_i, [_VP, _VP, _i, _i, _i, _P]),
    "dct_dropout_fwd": (_i, [_VP, _VP, _P, _f, _u64, _u64, _i, _P]),
    "dct_dropout_fwd_dev": (_i, [_VP, _VP, _P, _f, _u64, _P, _i, _i, _P]),
    "dct_dropout_maxpool2x2_fwd_codes": (_i, [_VP, _VP, _P, _f, _u64, _P, _i, _i, _P]),
    "dct_dropout_apply": (_i, [_VP, _VP, _P, _f, _i, _P]),
    "dct_relu_bwd": (_i, [_VP, _VP, _VP, _f, _i, _P]),
    "dct_cast": (_i, [_VP, _VP, _i, _i, _P]),
    "dct_loss_workspace_bytes": (_sz, [_i64]),
    "dct_ce_fwd": (_i, [_P, _P, _i64, _i, _i, _P, _P, _sz, _P]),
    "dct_ce_bwd": (_i, [_P, _P, _i64, _i, _i, _P, _P, _f, _P, _i, _P]),
    "dct_ce_step": (_i, [_P, _P, _i64, _i, _i, _P, _P, _f, _P, _i, _P, _sz, _P]),
    "dct_softmax_fwd": (_i, [_P, _P, _i64, _i, _P]),
    "dct_softmax_bwd": (_i, [_P, _P, _P, _i64, _i, _i, _P]),
    "dct_entropy_fwd": (_i, [_P, _P, _i64, _i, _P]),
    "dct_entropy_bwd": (_i, [_P, _P, _P, _i64, _i, _P]),
    "dct_jsd_map_fwd": (_i, [_P, _i, _P, _i64, _i, _P]),
    "dct_jsd_map_bwd": (_i, [_P, _i, _P, _P, _i64, _i, _P]),
    "dct_kl_map_fwd": (_i, [_P, _P, _P, _i64, _i, _f, _P]),
    "dct_kl_map_bwd": (_i, [_P, _P, _P, _P, _i64, _i, _f, _P]),
    "dct_jsd_logits_fwd": (_i, [_P, _i, _i64, _i, _P, _P, _sz, _P]),
    "dct_jsd_logits_bwd": (_i, [_P, _i, _i64, _i, _P, _f, _P, _i, _P]),
    "dct_jsd_logits_step": (_i, [_P, _i, _i64, _i, _P, _P, _P, _f, _P, _i, _P, _sz, _P]),
    "dct_kl_logits_fwd": (_i, [_P, _P, _i64, _i, _f, _P, _P, _sz, _P]),
    "dct_kl_logits_bwd": (_i, [_P, _P, _i64, _i, _f, _P, _f, _P, _i, _P]),
    "dct_argmax": (_i, [_P, _P, _i64, _i, _P]),
    "dct_fgsm_step": (_i, [_P, _P, _f, _P, _P, _i64, _P]),
    "dct_adam_flat": (_i, [_P, _P, _P, _P, _i64, _f, _f, C.c_double, C.c_double, _f, _f, _f, _P, _P]),
    "dct_adam_flat_dev": (_i, [_P, _P, _P, _P, _i64, _P, _P, C.c_double, C.c_double, _f, _f, _f, _P, _P]),
    "dct_enet_conv": (_i, [_VP, _P, _P, _TP, _VP, _DP, _i, _i, _i, _i, _VP, _VP, _i, _i, _P]),
    "dct_enet_reduce_workspace_bytes": (_sz, [_i]),
    "dct_enet_bn_fwd_stats": (_i, [_VP, _P, _P, _f, _f, _P, _P, _i, _P, _P, _P, _P, _P, _i, _i, _P, _sz, _P]),
    "dct_enet_bn_fwd_stats_rows": (_i, [_VP, _P, _P, _f, _f, _P, _P, _i, _P, _P, _P, _P, _P, _i, _i, _P, _sz, _i, _P]),
    "dct_enet_conv_stats": (_i, [_VP, _P, _P, _TP, _VP, _DP, _i, _i, _i, _i, _i, _i, _P, _i, _P, _P]),
    "dct_enet_bn_bwd": (_i, [_VP, _VP, _VP, _P, _P, _P, _i, _P, _P, _P, _P, _P, _P, _i, _VP, _i, _i, _P, _sz, _P]),
    "dct_enet_bn_bwd_rows": (_i, [_VP, _VP, _VP, _P, _P, _P, _i, _P, _P, _P, _P, _P, _P, _i, _VP, _i, _i, _P, _sz, _i, _P]),
    "dct_enet_conv_bnbwd_stats": (_i, [_VP, _P, _VP, _DP, _i, _i, _i, _i, _i, _i, _VP, _P, _P, _P, _i, _P, _P, _P, _i, _P, _P]),
    "dct_enet_conv_bwd_in": (_i, [_VP, _P, _TP, _P, _VP, _DP, _i, _i, _i, _i, _VP, _VP, _i, _i, _VP, _P, _P, _P, _i, _P, _P, _P, _i, _P, _P]),
    "dct_enet_bn_bwd_sums": (_i, [_VP, _VP, _VP, _P, _P, _P, _i, _P, _P, _P, _P, _P, _P, _i, _i, _i, _P, _sz, _i, _P]),
    "dct_enet_bn_bwd_apply": (_i, [_VP, _VP, _VP, _P, _P, _P, _i, _P, _P, _P, _VP, _i, _i, _i, _P]),
    "dct_enet_channel_sum": (_i, [_VP, _P, _i, _i, _P, _sz, _P]),
    "dct_enet_tail_fwd": (_i, [_VP, _TP, _VP, _VP, _TP, _P, _i, _i, _VP, _i, _i, _P]),
    "dct_enet_tail_bwd": (_i, [_VP, _VP, _P, _i, _i, _i, _VP, _i, _i, _P]),
    "dct_enet_wgrad_workspace_bytes": (_sz, [_VP, _VP, _DP]),
    "dct_enet_wgrad": (_i, [_VP, _TP, _VP, _TP, _P, _DP, _i, _i, _P, _sz, _P]),
    "dct_group_begin": (_i, [_i]),
    "dct_group_member": (_i, [_i]),
    "dct_group_max": (_i, []),
    "dct_group_abort": (_i, []),
    "dct_group_end": (_i, [_P, _P, _P]),
    "dct_leaves_begin": (_i, []),
    "dct_leaves_flush": (_i, [_P, _P]),
    "dct_leaves_end": (_i, [_P, _P]),
    "dct_bn_workspace_bytes": (_sz, [_i]),
    "dct_bn_fwd": (_i, [_VP, _P, _P, _f, _f, _P, _P, _i, _P, _P, _P, _P, _VP, _i, _i, _P, _sz, _P]),
    "dct_bn_bwd": (_i, [_VP, _VP, _P, _P, _P, _P, _P, _P, _i, _P, _i, _i, _VP, _i, _P, _sz, _P]),
    "dct_dice_counts": (_i, [_P, _P, _i, _i64, _i, _P, _P, _P, _P]),
    "dct_dice_update": (_i, [_P, _P, _P, _i, _i, _i, C.c_uint32, _f, _P, _P, _P]),
    "dct_tune_set": (_i, [_i, _i]),
    "dct_prof_enable": (_i, [_i]),
    "dct_prof_read": (_i, [_P, _P, _i]),
    "dct_clock_probe": (_i, [_P, C.c_uint64, _P]),
    "dct_stamp": (_i, [_P, C.c_uint32, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load libdct_hip.so (once).  torch is imported first so that the HIP runtime the
    library binds to is the one PyTorch already loaded (same streams, same allocations)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"dct_amd: {LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C <package>/csrc`).  There is no fallback path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


ERR_UNSUPPORTED = -2      # DCT_ERR_UNSUPPORTED


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().dct_status_string(status).decode()
        raise RuntimeError(f"dct_amd: {what or 'call'} failed: {msg} (status {status})")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    if t is None:
        return 0
    if not t.is_cuda:
        raise RuntimeError("dct_amd: tensors must live on the HIP device (there is no CPU fallback)")
    return t.data_ptr()


# While a pass group is open (hip_ops.PassGroup) only the entry points that RECORD their launches may be called: anything
# else would launch at once, ahead of the recorded launches it depends on.
_group_tls = threading.local()     # per thread, as the library's own recording state (csrc/enet.hip: thread_local g_grp)
_RECORDING = None


def set_group_open(flag: bool) -> None:
    _group_tls.open = bool(flag)


def call(name: str, *args) -> None:
    if getattr(_group_tls, "open", False) and not name.startswith(("dct_enet_", "dct_group_", "dct_leaves_")):
        raise RuntimeError(f"dct_amd: {name} inside an open pass group (only the Enet entry points record their launches)")
    check(getattr(load(), name)(*args), name)


def prof_enable(on: bool) -> None:
    load().dct_prof_enable(1 if on else 0)


def prof_read(reset: bool = True) -> dict:
    ms = (C.c_double * len(PROF_CLASSES))()
    cnt = (C.c_int64 * len(PROF_CLASSES))()
    check(load().dct_prof_read(ms, cnt, 1 if reset else 0), "dct_prof_read")
    return {k: {"ms": ms[i], "launches": cnt[i]} for i, k in enumerate(PROF_CLASSES)}

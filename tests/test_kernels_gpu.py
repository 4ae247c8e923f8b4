"""Per-kernel parity: every C-ABI entry point (through dct_amd.hip_ops) against a plain
PyTorch fp32 CPU computation of the same op, and against the oracle for the losses.
fp32 mode is the parity path (tight tolerance); bf16 mode is compared against the same
math on bf16-rounded inputs (tolerance = bf16 output rounding + accumulation order)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16]
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from dct_amd import hip_ops
    return hip_ops


def q(t, dtype):
    """round to the kernel's storage dtype, back to fp32 (the reference sees what the kernel sees)"""
    return t.to(dtype).float()


def to_dev(t_nchw, dtype):
    return t_nchw.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def to_cpu(t_nhwc):
    return t_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, ref, dtype, what, rtol32=2e-4, rtol16=2e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = ref.abs().max().item() + 1e-30
    rtol = rtol32 if dtype == torch.float32 else rtol16
    err = (got - ref).abs()
    tol = rtol * scale + rtol * ref.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        idx = np.unravel_index(int(err.argmax()), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} off; max err {err.max().item():.3e} at {idx} "
                             f"(got {got[idx].item():.6f} ref {ref[idx].item():.6f}); ref scale {scale:.3e}; "
                             f"finite={bool(torch.isfinite(got).all())}")


def kmajor(w_oihw, dtype):
    return w_oihw.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


# ------------------------------------------------------------------------------------ conv2d
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Cin,H,W,Cout,k,pad,stride,dil", [
    (2, 64, 20, 18, 128, 3, 0, 1, 1),     # valid 3x3, BN=128
    (2, 64, 20, 18, 64, 3, 0, 1, 1),      # BN=64 variant
    (1, 128, 13, 11, 64, 3, 2, 1, 1),     # full (dgrad-form) padding
    (1, 512, 8, 8, 256, 3, 0, 1, 1),      # tiny M, long K -> split-K path
    (2, 64, 16, 16, 64, 1, 0, 1, 1),      # 1x1
    (2, 128, 20, 18, 64, 2, 0, 2, 1),     # 2x2 stride 2 (convT dgrad form)
    (1, 64, 24, 24, 64, 3, 2, 1, 2),      # dilated
    (3, 64, 131, 7, 128, 3, 1, 1, 1),     # M not a tile multiple, pad 1
])
def test_conv2d_fwd(ops, dtype, B, Cin, H, W, Cout, k, pad, stride, dil):
    g = torch.Generator().manual_seed(1)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil))
    y = torch.empty(B, ref.shape[2], ref.shape[3], Cout, dtype=dtype, device=DEV)
    ops.conv2d(to_dev(x, dtype), kmajor(w, dtype), b.to(DEV), y, R=k, S=k, stride=stride, dil=dil, pad_h=pad, pad_w=pad, relu=True)
    close(to_cpu(y), ref, dtype, "conv2d fwd")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_epilogue_mask_accumulate_views(ops, dtype):
    g = torch.Generator().manual_seed(2)
    B, Cin, H, W, Cout = 2, 64, 12, 10, 128
    xbig = q(torch.randn(B, Cin + 64, H, W, generator=g), dtype)       # x is a channel slice of a wider buffer
    x = xbig[:, 64:]
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / 24, dtype)
    maskt = q(torch.randn(B, Cout, H - 2, W - 2, generator=g), dtype)
    old = q(torch.randn(B, Cout + 64, H - 2, W - 2, generator=g), dtype)  # y is a channel slice too
    conv = F.conv2d(x, w)
    masked = conv.clone()
    masked[:, :64] = torch.where(maskt[:, :64] > 0, conv[:, :64] * 2.0, torch.zeros(()))
    ref = old.clone()
    ref[:, :Cout] += masked
    xd = to_dev(xbig, dtype)
    yd = to_dev(old, dtype)
    ops.conv2d(xd[..., 64:], kmajor(w, dtype), None, yd[..., :Cout], mask=to_dev(maskt, dtype), mask_channels=64,
               mask_scale=2.0, accumulate=True)
    close(to_cpu(yd), ref, dtype, "conv2d mask/accumulate/views")


# shared-halo 3x3 kernels (igemm3m_kernel: 8 x 16 patches on large images; igemm3p_kernel: packed rows on small ones): taken for bf16
# 3x3 stride-1 layers whose tiles cover the image well and fill the device; checked against F.conv2d and against the per-tap kernel
# (dct_tune_set(DCT_TUNE_IGEMM_HALO, 0)).
@pytest.mark.parametrize("B,Cin,H,W,Cout,pad", [
    (4, 64, 130, 130, 128, 0),      # exact patch cover, Cout tile 128, one channel slice (single halo stage)
    (4, 64, 101, 117, 64, 0),       # ragged patches at the right / bottom edge, Cout tile 64 (four blocks per CU)
    (4, 64, 100, 116, 64, 2),       # the same tile in data-gradient form
    (5, 128, 122, 90, 128, 2),      # data-gradient form: pad 2, halo outside the image reads the zero page; two slices
    (3, 192, 96, 96, 256, 0),       # three channel slices, two Cout tiles
    (8, 512, 27, 27, 128, 0),       # small image: packed-rows kernel (5 rows of 25 per tile), two-way split over channel slices
    (16, 256, 18, 16, 256, 2),      # packed rows, data-gradient form (pad 2), ragged last tile, unsplit
    (16, 1024, 13, 13, 128, 0),     # packed rows: one 11 x 11 tile per image, four-way split
])
def test_conv2d_3x3_shared_halo(ops, B, Cin, H, W, Cout, pad):
    from dct_amd import _lib
    _lib.load().dct_tune_set(10, 1)          # small images: the packed-rows kernel is what these cases test
    try:
        _shared_halo_case(ops, B, Cin, H, W, Cout, pad)
    finally:
        _lib.load().dct_tune_set(10, _PACKED_DEFAULT)


@pytest.mark.parametrize("B,Cin,H,W,Cout,pad", [(4, 64, 130, 130, 128, 0), (4, 64, 101, 117, 64, 0), (5, 128, 122, 90, 128, 2),
                                                (3, 192, 96, 96, 256, 0)])
def test_conv2d_3x3_shared_halo_is_reproducible(ops, B, Cin, H, W, Cout, pad):
    """Repeat launches of the shared-halo kernel (LDS-DMA stages behind barriers) -- with and without the mask / accumulate
    epilogue -- must reproduce the first one bit for bit: a missed wait shows as run-to-run differences."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(12)
    x = to_dev(q(torch.randn(B, Cin, H, W, generator=g), dtype), dtype)
    w = kmajor(q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype), dtype)
    b = torch.randn(Cout, generator=g).to(DEV)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    mask = to_dev(q(torch.randn(B, Cout, Ho, Wo, generator=g), dtype), dtype)
    outs = []
    for _ in range(20):
        y = torch.empty(B, Ho, Wo, Cout, dtype=dtype, device=DEV)
        ops.conv2d(x, w, b, y, pad_h=pad, pad_w=pad, relu=True)
        z = torch.ones(B, Ho, Wo, Cout, dtype=dtype, device=DEV)
        ops.conv2d(x, w, None, z, pad_h=pad, pad_w=pad, mask=mask, mask_scale=2.0, accumulate=True)
        outs.append((y, z))
    torch.cuda.synchronize()
    for y, z in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(z, outs[0][1])


_PACKED_DEFAULT = 1
_LEAN_DEFAULT = 31   # DCT_TUNE_LEAN default (csrc/wgrad.hip g_tune_lean)
# repeat launches of the race screens: 200 when a kernel's synchronisation was touched (DCT_LONG_TESTS=1), 50 in the default run,
# which has to stay inside the driver's budget as tests are added
_RACE_LAUNCHES = 200 if os.environ.get("DCT_LONG_TESTS") else 50


def _shared_halo_case(ops, B, Cin, H, W, Cout, pad):
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype)
    b = torch.randn(Cout, generator=g)
    conv = F.conv2d(x, w, b, padding=pad)
    Ho, Wo = conv.shape[2], conv.shape[3]
    xd, wd, bd = to_dev(x, dtype), kmajor(w, dtype), b.to(DEV)
    y = torch.empty(B, Ho, Wo, Cout, dtype=dtype, device=DEV)
    ops.conv2d(xd, wd, bd, y, pad_h=pad, pad_w=pad, relu=True)
    close(to_cpu(y), F.relu(conv), dtype, "halo conv fwd")
    lib = _lib.load()
    lib.dct_tune_set(7, 0)
    try:
        y2 = torch.empty_like(y)
        ops.conv2d(xd, wd, bd, y2, pad_h=pad, pad_w=pad, relu=True)
    finally:
        lib.dct_tune_set(7, 1)
    # same products, different K order (slice-major vs tap-major): fp32 accumulation differs in the last bits only
    assert (y.float() - y2.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, y2.float().abs().max().item())
    # mask (ReLU gate of the destination) + channel-slice views + accumulate
    maskt = q(torch.randn(B, Cout, Ho, Wo, generator=g), dtype)
    old = q(torch.randn(B, Cout + 64, Ho, Wo, generator=g), dtype)
    plain = F.conv2d(x, w, padding=pad)
    masked = plain.clone()
    masked[:, :64] = torch.where(maskt[:, :64] > 0, plain[:, :64] * 2.0, torch.zeros(()))
    ref = old.clone()
    ref[:, :Cout] += masked
    yd = to_dev(old, dtype)
    ops.conv2d(xd, wd, None, yd[..., :Cout], pad_h=pad, pad_w=pad, mask=to_dev(maskt, dtype), mask_channels=64,
               mask_scale=2.0, accumulate=True)
    close(to_cpu(yd), ref, dtype, "halo conv mask/accumulate/views")


# Lean loop forms (DCT_TUNE_LEAN, knob 38): the same kernels with a shorter instruction stream -- staging by buffer_load ... lds with
# constant lane offsets (lanes that must stage zeros hold an offset the descriptor rejects), scalar step offsets, immediate read
# offsets.  Every form must reproduce the plain form bit for bit.  (Two further forms of the packed-rows loop were built on top of the
# lean one, verified the same way -- 150 repeat launches each -- and removed because they lost: a ring of three weight stages behind
# counted vmcnt waits and raw barriers, and all twelve fragment reads of a K-step issued up front: DESIGN.md 10.)
@pytest.mark.parametrize("B,Cin,H,W,Cout,pad", [
    (8, 512, 27, 27, 128, 0),       # 5 rows of 25 per tile, two-way split over channel slices (fp32 slabs)
    (16, 256, 18, 16, 256, 2),      # data-gradient form (pad 2: halo lanes outside the image stage zeros), ragged last tile
    (16, 1024, 13, 13, 128, 0),     # one 11 x 11 tile per image, four-way split: 36 K-steps per block
    (16, 512, 16, 16, 512, 0),      # 14 x 14 outputs, two tiles per image, unsplit: 72 K-steps, eight channel slices
    (3, 64, 20, 30, 128, 0),        # a single channel slice: nine K-steps, the ring never wraps a slice
])
def test_conv2d_packed_rows_lean_form_is_bit_identical(ops, B, Cin, H, W, Cout, pad):
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(21)
    x = to_dev(q(torch.randn(B, Cin, H, W, generator=g), dtype), dtype)
    w = kmajor(q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype), dtype)
    b = torch.randn(Cout, generator=g).to(DEV)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    lib = _lib.load()
    lib.dct_tune_set(10, 1)
    lib.dct_tune_set(24, 30)         # every case on the packed-rows kernel, whatever its tile fill
    outs = {}
    try:
        for form in (0, 2):          # plain, lean
            assert lib.dct_tune_set(38, form) == 0
            y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=dtype, device=DEV)
            ops.conv2d(x, w, b, y, pad_h=pad, pad_w=pad, relu=True)
            outs[form] = y
        first = outs[2]
        for _ in range(20):          # repeat launches reproduce the first (LDS-DMA stages behind barriers)
            y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=dtype, device=DEV)
            ops.conv2d(x, w, b, y, pad_h=pad, pad_w=pad, relu=True)
            assert torch.equal(y.view(torch.int16), first.view(torch.int16))
    finally:
        lib.dct_tune_set(38, _LEAN_DEFAULT)
        lib.dct_tune_set(24, 50)
        lib.dct_tune_set(10, _PACKED_DEFAULT)
    assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16)), "lean form differs from the plain form"
    assert not torch.isnan(first.float()).any()


@pytest.mark.parametrize("B,Cin,H,W,Cout,pad", [
    (4, 64, 130, 132, 64, 0),       # filter-row kernel, wide images: runs of 64 pixels, ragged row tail (130 = 2 x 64 + 2: one sub-step)
    (8, 256, 27, 27, 128, 0),       # filter-row kernel, narrow images: packed rows at pitch 27, last step of an image partial
    (16, 1024, 11, 11, 128, 0),     # per-tap kernel (step fill under the filter-row threshold): one wave per K-step decodes the pixels
    (3, 128, 70, 101, 128, 1),      # padding: the filter-row kernel keeps its plain form, the per-tap kernel's bounds go through the table
    # round 5: the lean filter-row loop skips the 16-row sub-steps of a K-step that hold no dy pixel (they were staged as zeros)
    (4, 64, 88, 88, 64, 0),         # wide rows of 86 = 64 + 22: the tail step multiplies two sub-steps of four (the UNet's enc1a)
    (4, 128, 48, 48, 128, 0),       # narrow rows of 46 at pitch 48, one per step: three sub-steps (enc2a)
    (3, 128, 30, 20, 64, 0),        # three packed rows of 18 per step (58 LDS rows: four sub-steps), an image's last step holds ONE row: two
    (2, 64, 50, 35, 64, 0),         # rows of 33 at pitch 35: three sub-steps, the third with one live row
])
def test_conv2d_wgrad_lean_forms_are_bit_identical(ops, B, Cin, H, W, Cout, pad):
    """Weight gradients with DCT_TUNE_LEAN = 0 (plain loops) and 15 (lean loops): dW bit for bit; the bias gradient to fp32 rounding
    (the lean filter-row kernel sums it with v_dot2c_f32_bf16, two pixels per instruction).  The filter-row kernel is forced for
    every 3x3 shape (knob 9: its fill threshold), so that both forms run the same kernel whatever the planner would choose."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(23)
    x = to_dev(q(torch.randn(B, Cin, H, W, generator=g), dtype), dtype)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    dy = to_dev(q(torch.randn(B, Cout, Ho, Wo, generator=g), dtype), dtype)
    lib = _lib.load()
    outs = {}
    force_rows = (B, Cin, H, W) in ((4, 64, 130, 132), (4, 64, 88, 88), (4, 128, 48, 48), (3, 128, 30, 20), (2, 64, 50, 35))
    try:
        if force_rows:
            assert lib.dct_tune_set(9, 1) == 0
        for form in (0, 31, 31):
            assert lib.dct_tune_set(38, form) == 0
            dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=DEV)
            db = torch.full((Cout,), float("nan"), device=DEV)
            ops.conv2d_wgrad(dy, x, dw, pad_h=pad, pad_w=pad, accumulate=False, db=db)
            torch.cuda.synchronize()
            outs.setdefault(form, []).append((dw, db))
    finally:
        lib.dct_tune_set(38, _LEAN_DEFAULT)
        lib.dct_tune_set(9, 70)
    (dw0, db0), (dw1, db1), (dw2, db2) = outs[0][0], outs[31][0], outs[31][1]
    assert torch.equal(dw0, dw1), "lean weight gradient differs from the plain form"
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2), "lean form does not reproduce itself"
    assert (db0 - db1).abs().max().item() <= 2e-6 * max(1.0, db0.abs().max().item()) * math.sqrt(B * Ho * Wo)


@pytest.mark.parametrize("dtype,B,Cin,H,W,Cout,halo_min_blocks", [
    (torch.bfloat16, 4, 64, 100, 132, 64, 400),     # shared-halo kernel, 64-channel tile: pooled in the epilogue (even extents)
    (torch.bfloat16, 4, 128, 83, 147, 128, 400),    # ... 128-channel tile, odd extents: ceil-mode windows of one row / one column / one pixel
    (torch.bfloat16, 5, 128, 66, 98, 256, 400),     # ... two channel tiles
    (torch.bfloat16, 1, 64, 29, 23, 64, 1),         # ... a single partial patch row / column per image
    (torch.bfloat16, 8, 256, 27, 27, 128, 400),     # packed-rows kernel: the pooling kernel runs behind it
    (torch.bfloat16, 2, 64, 20, 20, 64, 400),       # per-tap kernel: same
    (torch.float32, 2, 32, 21, 18, 64, 400),        # fp32: same
])
def test_conv2d_with_pooling_is_bit_identical_to_conv_then_pool(ops, dtype, B, Cin, H, W, Cout, halo_min_blocks):
    """dct_conv_desc.pool_out / pool_codes: the 2x2 ceil-mode max pooling (and its routing codes) of a convolution's output from the
    same call -- out of the staged tile in the shared-halo kernel, as a launch behind the conv elsewhere -- against
    dct_conv2d followed by dct_maxpool2x2_fwd_codes: y, pooled values, codes and ReLU-gate bits bit for bit."""
    from dct_amd import _lib
    g = torch.Generator().manual_seed(53)
    x = to_dev(q(torch.randn(B, Cin, H, W, generator=g), dtype), dtype)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    bias = torch.randn(Cout, generator=g).to(DEV)
    wp = kmajor(q(w, dtype), dtype)
    Ho, Wo = H - 2, W - 2
    Hp, Wp = (Ho + 1) // 2, (Wo + 1) // 2
    lib = _lib.load()
    assert lib.dct_tune_set(19, halo_min_blocks) == 0
    try:
        outs = []
        for fused in (False, True):
            y = torch.full((B, Ho, Wo, Cout), float("nan"), device=DEV, dtype=dtype)
            pooled = torch.full((B, Hp, Wp, Cout), float("nan"), device=DEV, dtype=dtype)
            codes = torch.full((B, Hp, Wp, Cout), 255, device=DEV, dtype=torch.uint8)
            bits = ops.relu_bits_like(y) if dtype == torch.bfloat16 else None
            if fused:
                ops.conv2d(x, wp, bias, y, relu=True, relu_bits_out=bits, pool_out=pooled, pool_codes=codes)
            else:
                ops.conv2d(x, wp, bias, y, relu=True, relu_bits_out=bits)
                ops.maxpool_fwd(y, pooled, codes=codes)
            torch.cuda.synchronize()
            outs.append((y, pooled, codes, bits))
        # without codes (an inference pass)
        y2 = torch.empty_like(outs[0][0])
        pooled2 = torch.full_like(outs[0][1], float("nan"))
        ops.conv2d(x, wp, bias, y2, relu=True, pool_out=pooled2)
        # pool_only: y's contents are unspecified afterwards, the pooled tensor and the codes are the same
        y3 = torch.empty_like(outs[0][0])
        pooled3 = torch.full_like(outs[0][1], float("nan"))
        codes3 = torch.full_like(outs[0][2], 255)
        ops.conv2d(x, wp, bias, y3, relu=True, pool_out=pooled3, pool_codes=codes3, pool_only=True)
        torch.cuda.synchronize()
    finally:
        lib.dct_tune_set(19, 400)
    (y0, p0, c0, b0), (y1, p1, c1, b1) = outs
    assert torch.isfinite(p0.float()).all() and (c0 < 16).all()
    assert torch.equal(y0, y1) and torch.equal(p0, p1) and torch.equal(c0, c1)
    assert b0 is None or torch.equal(b0, b1)
    assert torch.equal(y0, y2) and torch.equal(p0, pooled2)
    assert torch.equal(p0, pooled3) and torch.equal(c0, codes3)
    ref = torch.nn.functional.max_pool2d(y0.float().permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)
    assert torch.equal(p0.float(), ref)


@pytest.mark.parametrize("B,Ha,Wa,acc", [(4, 100, 132, False), (5, 83, 147, True), (2, 120, 110, False)])
def test_conv2d_dgrad_with_stem_weight_gradient(ops, B, Ha, Wa, acc):
    """dct_conv_desc.stem_x: the UNet stem's weight / bias gradient taken from the output tile of the 64 -> 64 channel data gradient that
    produces its dy (masked by ReLU-gate bits, rounded to bf16; x windows as bf16 high + low parts on the matrix pipe) against the two
    launches it replaces (dct_conv2d writing dy, dct_conv_cin1_wgrad reading it back): same gradients to fp32 rounding, with and
    without accumulation, ragged patches, a single-patch image."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(61)
    C = 64
    dd = to_dev(q(torch.randn(B, C, Ha - 2, Wa - 2, generator=g), dtype), dtype)            # gradient at the second conv's output
    w = q(torch.randn(C, C, 3, 3, generator=g) / 24, dtype)                                  # second conv's weights [cout][cin][3][3]
    wd = kmajor(w.flip(2, 3).permute(1, 0, 2, 3).contiguous(), dtype)                        # data-gradient pack: [cin][tap'][cout]
    a1 = to_dev(q(torch.randn(B, C, Ha, Wa, generator=g), dtype), dtype)                     # the stem's output (its sign is the gate)
    a1bits = _pack_bits(a1)
    ximg = torch.rand(B, Ha + 2, Wa + 2, 1, generator=g).to(DEV)
    seed_w, seed_b = torch.randn(C, 9, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    lib = _lib.load()
    assert lib.dct_tune_set(19, 1) == 0
    try:
        da = torch.full((B, Ha, Wa, C), float("nan"), device=DEV, dtype=dtype)
        ops.conv2d(dd, wd, None, da, pad_h=2, pad_w=2, mask=a1, mask_bits=a1bits)
        dw0, db0 = seed_w.clone(), seed_b.clone()
        ops.conv_cin1_wgrad(ximg, da, dw0, db0, accumulate=acc)
        dw1, db1 = seed_w.clone(), seed_b.clone()
        scratch = torch.empty_like(da)
        ops.conv2d(dd, wd, None, scratch, pad_h=2, pad_w=2, mask=a1, mask_bits=a1bits, stem=(ximg, dw1, db1, acc))
        torch.cuda.synchronize()
    finally:
        lib.dct_tune_set(19, 400)
    # fp64 reference from the stored (masked, rounded) data gradient
    daf = da.double().permute(0, 3, 1, 2).cpu()
    xr = ximg.double().permute(0, 3, 1, 2).cpu()
    cols = F.unfold(xr, 3).view(B, 9, Ha * Wa)
    ref_w = torch.einsum("bcp,btp->ct", daf.reshape(B, C, Ha * Wa), cols) + (seed_w.double().cpu() if acc else 0)
    ref_b = daf.sum((0, 2, 3)) + (seed_b.double().cpu() if acc else 0)
    sw, sb = ref_w.abs().max().item(), ref_b.abs().max().item()
    assert (dw1.double().cpu() - ref_w).abs().max().item() <= 1e-4 * sw, "fused stem dW vs fp64"
    assert (db1.double().cpu() - ref_b).abs().max().item() <= 1e-4 * sb, "fused stem db vs fp64"
    assert (dw0.double().cpu() - ref_w).abs().max().item() <= 1e-4 * sw, "two-launch stem dW vs fp64"
    assert (dw1 - dw0).abs().max().item() <= 5e-5 * sw and (db1 - db0).abs().max().item() <= 5e-5 * sb


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,C,H,W,Hs,Ws", [(2, 64, 37, 50, 28, 41), (1, 128, 12, 9, 20, 16), (3, 64, 21, 21, 11, 11)])
def test_maxpool_bwd_with_skip_gather(ops, dtype, B, C, H, W, Hs, Ws):
    """dct_maxpool2x2_bwd_codes_skip: un-pooling of (dy + bilinear backward of the gradient at a resized copy of the pooled tensor) against
    dct_bilinear_bwd (accumulating into dy) followed by dct_maxpool2x2_bwd_codes -- down- and up-sampled skips, odd extents, the ReLU gate."""
    g = torch.Generator().manual_seed(91)
    x = to_dev(q(torch.randn(B, C, H, W, generator=g), dtype), dtype)
    Hp, Wp = (H + 1) // 2, (W + 1) // 2
    pooled = torch.empty(B, Hp, Wp, C, device=DEV, dtype=dtype)
    codes = torch.empty(B, Hp, Wp, C, device=DEV, dtype=torch.uint8)
    ops.maxpool_fwd(x, pooled, codes=codes)
    dy = to_dev(q(torch.randn(B, C, Hp, Wp, generator=g), dtype), dtype)
    skip = to_dev(q(torch.randn(B, C, Hs, Ws, generator=g), dtype), dtype)
    want_dy = dy.clone()
    ops.bilinear_bwd(skip, want_dy, accumulate=True)
    want = torch.full((B, H, W, C), float("nan"), device=DEV, dtype=dtype)
    ops.maxpool_bwd(x, want_dy, want, relu_mask=True, scale=2.0, codes=codes)
    got = torch.full((B, H, W, C), float("nan"), device=DEV, dtype=dtype)
    ops.maxpool_bwd(x, dy, got, relu_mask=True, scale=2.0, codes=codes, skip=skip)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got == 0, want == 0) or dtype == torch.bfloat16        # the routing is the same (bf16: a sum may round to zero in one form only)
    close(to_cpu(got), to_cpu(want), dtype, "unpool with skip gather", rtol32=2e-6, rtol16=8e-3)


@pytest.mark.parametrize("B,Ha,Wa", [(4, 100, 132), (5, 83, 147)])
def test_conv2d_fused_epilogues_race_screen(ops, B, Ha, Wa):
    """The two round-4 epilogue branches of the shared-halo kernel re-use LDS the K loop has just left (the pooled items read the staged
    tile before the row stores; the stem branch writes the x-patch parts and the four waves' partial products into dead stage space,
    reads the tile back through transposing reads, and adds behind two more barriers).  A missed barrier shows as a launch that
    differs: 200 launches of each must reproduce the first bit for bit (pooled values, codes, stem dW / db)."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(71)
    C = 64
    x = to_dev(q(torch.randn(B, C, Ha + 2, Wa + 2, generator=g), dtype), dtype)
    w = kmajor(q(torch.randn(C, C, 3, 3, generator=g) / 24, dtype), dtype)
    bias = torch.randn(C, generator=g).to(DEV)
    dd = to_dev(q(torch.randn(B, C, Ha - 2, Wa - 2, generator=g), dtype), dtype)
    a1 = to_dev(q(torch.randn(B, C, Ha, Wa, generator=g), dtype), dtype)
    a1bits = _pack_bits(a1)
    ximg = torch.rand(B, Ha + 2, Wa + 2, 1, generator=g).to(DEV)
    lib = _lib.load()
    assert lib.dct_tune_set(19, 1) == 0
    try:
        first = None
        for it in range(_RACE_LAUNCHES):
            y = torch.empty(B, Ha, Wa, C, device=DEV, dtype=dtype)
            pooled = torch.full((B, (Ha + 1) // 2, (Wa + 1) // 2, C), float("nan"), device=DEV, dtype=dtype)
            codes = torch.full(pooled.shape, 255, device=DEV, dtype=torch.uint8)
            ops.conv2d(x, w, bias, y, relu=True, pool_out=pooled, pool_codes=codes, pool_only=True)
            dw = torch.full((C, 9), float("nan"), device=DEV)
            db = torch.full((C,), float("nan"), device=DEV)
            scratch = torch.empty(B, Ha, Wa, C, device=DEV, dtype=dtype)
            ops.conv2d(dd, w, None, scratch, pad_h=2, pad_w=2, mask=a1, mask_bits=a1bits, stem=(ximg, dw, db, False))
            torch.cuda.synchronize()
            cur = (pooled, codes, dw, db)
            if first is None:
                first = cur
                assert torch.isfinite(pooled.float()).all() and torch.isfinite(dw).all() and torch.isfinite(db).all()
            else:
                for k, (a, b) in enumerate(zip(first, cur)):
                    assert torch.equal(a, b), f"launch {it}: output {k} differs from the first launch"
    finally:
        lib.dct_tune_set(19, 400)


@pytest.mark.parametrize("B,Cin,H,W,Cout,pad", [
    (16, 1024, 11, 11, 128, 0),     # per-tap kernel: the rotating decoder wave and its two-slot offset table, 31 K-steps per block
    (6, 128, 9, 9, 128, 0),         # per-tap kernel, ONE to two K-steps per chunk: the table's prologue and tail
    (8, 256, 27, 27, 128, 0),       # filter-row kernel, narrow images
    (4, 64, 130, 132, 64, 0),       # filter-row kernel, wide images with a ragged row tail
])
def test_conv2d_wgrad_lean_forms_race_screen(ops, B, Cin, H, W, Cout, pad):
    """The lean weight-gradient kernels stage through buffer loads behind the same barriers as before, and the per-tap one adds a new
    hand-off: one wave per K-step writes the step-after-next's pixel offsets into an LDS table slot that every wave read one step
    earlier.  A missed wait there shows as a launch that differs: 200 launches must reproduce the first bit for bit."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(31)
    x = to_dev(q(torch.randn(B, Cin, H, W, generator=g), dtype), dtype)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    dy = to_dev(q(torch.randn(B, Cout, Ho, Wo, generator=g), dtype), dtype)
    first = None
    for it in range(_RACE_LAUNCHES):
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=DEV)
        db = torch.full((Cout,), float("nan"), device=DEV)
        ops.conv2d_wgrad(dy, x, dw, pad_h=pad, pad_w=pad, accumulate=False, db=db)
        if first is None:
            first = (dw, db)
            assert not torch.isnan(dw).any() and not torch.isnan(db).any()
        else:
            assert torch.equal(dw, first[0]) and torch.equal(db, first[1]), f"launch {it} differs from the first"


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,pad,stride,scatter", [
    (16, 1024, 11, 11, 256, 3, 0, 1, False),    # per-tap kernel, taps inside the image
    (16, 256, 11, 11, 256, 3, 2, 1, False),     # data-gradient form: taps outside the image through the per-row tap mask
    (4, 128, 40, 44, 64, 3, 1, 1, False),       # 256 x 64 tile (Cout = 64), bounds
    (16, 512, 9, 9, 256, 1, 0, 1, True),        # transposed 2x2 s2 convolution forward (scatter epilogue)
])
def test_conv2d_per_tap_lean_form_is_bit_identical(ops, B, Cin, H, W, Cout, k, pad, stride, scatter):
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(29)
    x = to_dev(q(torch.randn(B, Cin, H, W, generator=g), dtype), dtype)
    lib = _lib.load()
    lib.dct_tune_set(7, 0)           # per-tap kernel everywhere
    outs = []
    try:
        if scatter:
            w = q(torch.randn(4 * Cout, Cin, generator=g) / math.sqrt(Cin), dtype).to(DEV).to(dtype).contiguous()
            b = torch.randn(Cout, generator=g).to(DEV)
            for form in (0, 15):
                lib.dct_tune_set(38, form)
                y = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), dtype=dtype, device=DEV)
                ops.conv2d(x, w, b, y, R=1, S=1, relu=True, scatter2x2=True)
                outs.append(y)
        else:
            w = kmajor(q(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype), dtype)
            b = torch.randn(Cout, generator=g).to(DEV)
            Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
            for form in (0, 15):
                lib.dct_tune_set(38, form)
                y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=dtype, device=DEV)
                ops.conv2d(x, w, b, y, R=k, S=k, stride=stride, pad_h=pad, pad_w=pad, relu=True)
                outs.append(y)
    finally:
        lib.dct_tune_set(38, _LEAN_DEFAULT)
        lib.dct_tune_set(7, 1)
    assert not torch.isnan(outs[0].float()).any()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), "lean per-tap kernel differs from the plain form"


@pytest.mark.parametrize("B,Cin,H,W,Cout,pad,halo", [
    (16, 1024, 11, 11, 1024, 0, 1),   # the centre's second convolution: per-tap kernel, 88 tiles x 5 splits (fp32 slabs + fold)
    (16, 1024, 9, 9, 512, 2, 1),      # its data-gradient form: packed-rows kernel split over channel slices
    (16, 512, 16, 16, 512, 0, 1),     # packed rows, unsplit, staged epilogue
    (5, 128, 30, 46, 128, 0, 0),      # per-tap kernel on a mid-size image, a block count that is not a multiple of eight
    (3, 64, 40, 44, 64, 1, 0),        # 256 x 64 per-tap tile with bounds
])
def test_conv2d_xcd_block_order_is_bit_identical(ops, B, Cin, H, W, Cout, pad, halo):
    """DCT_TUNE_IGEMM_XCD: the per-tap and packed-rows kernels launched as a 1-D grid whose blocks are re-dealt XCD by XCD (2: on
    every layer) give the natural grid's output bit for bit -- same tiles, same K order; only which block computes which tile moves.
    The padding blocks of the 1-D launch (total not a multiple of eight) must leave nothing behind."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(41)
    x = to_dev(q(torch.randn(B, Cin, H, W, generator=g), dtype), dtype)
    w = kmajor(q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype), dtype)
    b = torch.randn(Cout, generator=g).to(DEV)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    lib = _lib.load()
    outs = []
    try:
        lib.dct_tune_set(7, halo)
        for mode in (0, 2, 2):
            assert lib.dct_tune_set(39, mode) == 0
            y = torch.full((B, Ho + 1, Wo, Cout), float("nan"), dtype=dtype, device=DEV)     # one guard row behind every image
            ops.conv2d(x, w, b, y[:, :Ho], pad_h=pad, pad_w=pad, relu=True)
            outs.append(y)
    finally:
        lib.dct_tune_set(39, 1)
        lib.dct_tune_set(7, 1)
    assert not torch.isnan(outs[0][:, :Ho].float()).any() and torch.isnan(outs[1][:, Ho].float()).all()
    assert torch.equal(outs[0][:, :Ho].view(torch.int16), outs[1][:, :Ho].view(torch.int16)), "XCD-dealt launch differs from the natural grid"
    assert torch.equal(outs[1][:, :Ho].view(torch.int16), outs[2][:, :Ho].view(torch.int16))


def _pooled_gradient_case(B, C, H, W, seed):
    """A ReLU'd activation d [B,H,W,C], its 2x2 ceil-mode pooling codes, and a random gradient at the pooled tensor (bf16, dense)."""
    from dct_amd import hip_ops as K
    g = torch.Generator().manual_seed(seed)
    d = torch.relu(torch.randn(B, H, W, C, generator=g)).to(DEV).to(torch.bfloat16)
    Hp, Wp = (H + 1) // 2, (W + 1) // 2
    pooled = torch.empty(B, Hp, Wp, C, dtype=torch.bfloat16, device=DEV)
    codes = torch.empty(B, Hp, Wp, C, dtype=torch.uint8, device=DEV)
    K.maxpool_fwd(d, pooled, codes=codes)
    dp = torch.randn(B, Hp, Wp, C, generator=g).to(DEV).to(torch.bfloat16)
    dd = K.maxpool_bwd(None, dp, torch.empty(B, H, W, C, dtype=torch.bfloat16, device=DEV), relu_mask=True, scale=1.0, codes=codes)
    return dp, codes, dd


@pytest.mark.parametrize("B,C,H,W,Cout", [
    (4, 64, 60, 76, 64),        # the first level's form: 64 -> 64 channels, one channel slice, four waves (two windows per thread)
    (3, 128, 58, 44, 128),      # 128-channel tiles, two channel slices: the second slice's windows are fetched during taps 3..8
    (2, 256, 57, 57, 256),      # odd extents: ceil-mode windows that reach past the tensor, four slices
    (5, 64, 33, 95, 128),       # 64 -> 128 channels (128-channel tile, single slice), ragged patches on both edges
])
def test_conv2d_data_gradient_unpools_on_load_bit_identical(ops, B, C, H, W, Cout):
    """dct_conv_desc.unpool_codes: the data-gradient convolution of {gradient at the pooled tensor, routing codes} against the same
    convolution of the un-pooled gradient written by dct_maxpool2x2_bwd_codes first -- every output bit, with the ReLU-gate bits of the
    consumer and with accumulation."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    dp, codes, dd = _pooled_gradient_case(B, C, H, W, 51)
    g = torch.Generator().manual_seed(52)
    w = kmajor(q(torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(C * 9), dtype), dtype)
    act = to_dev(q(torch.randn(B, Cout, H + 2, W + 2, generator=g), dtype), dtype)       # the layer input whose ReLU gates the data gradient
    bits = _pack_bits(act)
    old = to_dev(q(torch.randn(B, Cout, H + 2, W + 2, generator=g), dtype), dtype)
    lib = _lib.load()
    try:
        lib.dct_tune_set(19, 1)          # shared-halo kernel whatever the block count ...
        lib.dct_tune_set(1, 1)           # ... and no split-K (these test shapes have few tiles; the layers that take the form have thousands)
        outs = []
        for unpool in (None, (codes, H, W)):
            src = dd if unpool is None else dp
            y = torch.full((B, H + 2, W + 2, Cout), float("nan"), dtype=dtype, device=DEV)
            ops.conv2d(src, w, None, y, pad_h=2, pad_w=2, mask=act, mask_bits=bits, unpool=unpool)
            z = old.clone()
            ops.conv2d(src, w, None, z, pad_h=2, pad_w=2, accumulate=True, unpool=unpool)
            outs.append((y, z))
    finally:
        lib.dct_tune_set(19, 400)
        lib.dct_tune_set(1, -1)
    assert not torch.isnan(outs[0][0].float()).any() and outs[0][0].float().abs().max().item() > 0
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)), "masked data gradient differs"
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16)), "accumulated data gradient differs"


def test_conv2d_unpool_on_load_is_refused_where_no_kernel_can():
    """A layer that does not take the shared-halo kernel (too few blocks here) must refuse -- before launching anything -- so that the
    caller can un-pool into a buffer: hip_ops raises UnpoolOnLoadUnsupported and leaves y untouched."""
    from dct_amd import hip_ops as K
    dtype = torch.bfloat16
    dp, codes, dd = _pooled_gradient_case(1, 128, 20, 20, 53)
    w = kmajor(q(torch.randn(128, 128, 3, 3, generator=torch.Generator().manual_seed(54)) / 34.0, dtype), dtype)
    y = torch.full((1, 22, 22, 128), 7.0, dtype=dtype, device=DEV)
    with pytest.raises(K.UnpoolOnLoadUnsupported):
        K.conv2d(dp, w, None, y, pad_h=2, pad_w=2, unpool=(codes, 20, 20))
    torch.cuda.synchronize()
    assert (y == 7.0).all()
    dw = torch.zeros(128, 3, 3, 128, device=DEV)
    a = torch.randn(1, 34, 32, 128, device=DEV).to(dtype)
    dp2, codes2, _ = _pooled_gradient_case(1, 128, 32, 30, 55)       # rows of 30 pixels: two image rows per K-step -- not expanded on load
    with pytest.raises(K.UnpoolOnLoadUnsupported):
        K.conv2d_wgrad(dp2, a, dw, unpool=(codes2, 32, 30))
    torch.cuda.synchronize()
    assert (dw == 0).all()


@pytest.mark.parametrize("B,C,H,W,Cq", [
    (4, 64, 60, 140, 64),       # wide rows: 140 = 64 + 64 + 12 pixels per row, a one-sub-step tail
    (3, 128, 41, 122, 64),      # odd height (the last pooled row has one image row), two dy channel tiles
    (2, 256, 57, 57, 128),      # one whole row per K-step (pitch 59), odd width: the last window holds one pixel
    (6, 64, 44, 40, 192),       # rows of 40 at pitch 42
])
def test_conv2d_wgrad_unpools_on_load_bit_identical(ops, B, C, H, W, Cq):
    """dct_conv_desc.unpool_codes on the weight gradient: dy = {gradient at the pooled tensor, routing codes}, expanded by the
    filter-row kernel while it stages, against the un-pooled gradient read from memory: dW and db bit for bit (same kernel, same
    K order, same operands), writing and accumulating."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    dp, codes, dd = _pooled_gradient_case(B, C, H, W, 56)
    x = torch.randn(B, H + 2, W + 2, Cq, generator=torch.Generator().manual_seed(57)).to(DEV).to(dtype)
    lib = _lib.load()
    outs = []
    try:
        lib.dct_tune_set(9, 1)           # filter-row kernel whatever the fill
        for unpool in (None, (codes, H, W)):
            src = dd if unpool is None else dp
            dw = torch.full((C, 3, 3, Cq), float("nan"), device=DEV)
            db = torch.full((C,), float("nan"), device=DEV)
            ops.conv2d_wgrad(src, x, dw, accumulate=False, db=db, unpool=unpool)
            dw2, db2 = torch.ones_like(dw), torch.ones_like(db)
            ops.conv2d_wgrad(src, x, dw2, accumulate=True, db=db2, unpool=unpool)
            outs.append((dw, db, dw2, db2))
    finally:
        lib.dct_tune_set(9, 70)
    assert torch.isfinite(outs[0][0]).all() and outs[0][0].abs().max().item() > 0
    for a, b, what in zip(outs[0], outs[1], ("dW", "db", "dW accumulated", "db accumulated")):
        assert torch.equal(a, b), what


@pytest.mark.parametrize("C,accumulate,ignored", [(4, False, False), (2, True, True), (5, False, True)])
def test_ce_step_is_bit_identical_to_forward_then_backward(ops, C, accumulate, ignored):
    """dct_ce_step (the backward kernel folds the forward kernel's block partials itself) against dct_ce_fwd + dct_ce_bwd: loss, count and
    every gradient bit."""
    g = torch.Generator().manual_seed(71)
    P = 5 * 131 * 67
    logits = (3 * torch.randn(P, C, generator=g)).to(DEV)
    t = torch.randint(0, C, (P,), generator=g)
    if ignored:
        t[torch.rand(P, generator=g) < 0.2] = 255
    t = t.to(DEV)
    gscale = torch.tensor([0.61], device=DEV)
    old = torch.randn(P, C, generator=g).to(DEV)
    want_o = ops.ce_fwd(logits, t, C, 255)
    want_g = ops.ce_bwd(logits, t, C, want_o[1:2], old.clone(), gscale=gscale, gmul=8.0, ignore_index=255, accumulate=accumulate)
    got_g = old.clone()
    got_o = ops.ce_step(logits, t, C, got_g, gscale=gscale, gmul=8.0, ignore_index=255, accumulate=accumulate)
    torch.cuda.synchronize()
    assert torch.equal(got_o, want_o) and float(got_o[0]) > 0 and float(got_o[1]) == float((t != 255).sum())
    assert torch.equal(got_g, want_g)


@pytest.mark.parametrize("S,C,accumulate", [(2, 4, False), (3, 2, True), (6, 4, False)])
def test_jsd_step_in_one_pass_is_bit_identical(ops, S, C, accumulate):
    """dct_jsd_logits_step (mean JSD + the S softmax maps + the S logit gradients in one pass) against dct_jsd_logits_fwd, dct_softmax_fwd
    and dct_jsd_logits_bwd: every output bit -- the value because the pass keeps the forward kernel's grid and summation order."""
    g = torch.Generator().manual_seed(61)
    P = 3 * 97 * 101
    logits = [torch.randn(P, C, generator=g).to(DEV) for _ in range(S)]
    gscale = torch.tensor([0.37], device=DEV)
    old = [torch.randn(P, C, generator=g).to(DEV) for _ in range(S)]
    want_v = ops.jsd_logits_fwd(logits, C)
    want_p = [ops.softmax_fwd(lp, C) for lp in logits]
    want_g = ops.jsd_logits_bwd(logits, C, [o.clone() for o in old], gscale=gscale, gmul=4.0, accumulate=accumulate)
    got_g = [o.clone() for o in old]
    got_v, got_p = ops.jsd_logits_step(logits, C, got_g, True, gscale=gscale, gmul=4.0, accumulate=accumulate)
    assert torch.equal(got_v, want_v) and float(got_v) > 0
    for a, b in zip(got_p, want_p):
        assert torch.equal(a, b)
    for a, b in zip(got_g, want_g):
        assert torch.equal(a, b)
    v2, p2 = ops.jsd_logits_step(logits, C, None, False)          # value only
    assert torch.equal(v2, want_v) and p2 is None


def _pack_bits(t_nhwc):
    """ReLU-gate bits of a dense NHWC tensor as the kernels lay them out: byte (pixel, c // 8), bit c % 8."""
    pos = (t_nhwc.float() > 0).to(torch.int32)
    n, h, w, c = pos.shape
    return (pos.view(n, h, w, c // 8, 8) << torch.arange(8, device=pos.device, dtype=torch.int32)).sum(-1).to(torch.uint8)


# every conv path: shared halo (128- and 64-channel tiles), packed rows (with and without its split), per-tap staged, split-K
@pytest.mark.parametrize("B,Cin,H,W,Cout,pad,k,stride", [
    (4, 64, 130, 130, 128, 0, 3, 1), (4, 64, 101, 117, 64, 0, 3, 1), (5, 128, 122, 90, 128, 2, 3, 1),
    (16, 512, 13, 13, 1024, 0, 3, 1), (16, 1024, 11, 11, 1024, 2, 3, 1), (16, 256, 29, 29, 512, 0, 3, 1),
    (2, 64, 20, 22, 64, 0, 3, 1), (16, 256, 48, 48, 128, 0, 3, 1), (8, 128, 24, 24, 256, 0, 2, 2)])
def test_conv2d_relu_gate_bits(ops, B, Cin, H, W, Cout, pad, k, stride):
    """A forward launch with ``relu_bits_out`` leaves exactly the bits of (y > 0); a data-gradient launch with ``mask_bits``
    gives bit for bit what the launch masked by the activation itself gives."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(23)
    x = to_dev(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = kmajor(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    b = torch.randn(Cout, generator=g).to(DEV)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y0 = torch.empty(B, Ho, Wo, Cout, dtype=dtype, device=DEV)
    ops.conv2d(x, w, b, y0, R=k, S=k, stride=stride, pad_h=pad, pad_w=pad, relu=True)
    y1 = torch.empty_like(y0)
    bits = ops.relu_bits_like(y1)
    bits.fill_(0xA5)
    ops.conv2d(x, w, b, y1, R=k, S=k, stride=stride, pad_h=pad, pad_w=pad, relu=True, relu_bits_out=bits)
    assert torch.equal(y1, y0)
    assert torch.equal(bits, _pack_bits(y1))
    assert 0.2 < (y1 > 0).float().mean().item() < 0.8
    # the same launch as a masked one (what the data gradient of the NEXT layer does with y as its mask)
    act = to_dev(torch.randn(B, Cout, Ho, Wo, generator=g), dtype)
    want = torch.empty_like(y0)
    ops.conv2d(x, w, None, want, R=k, S=k, stride=stride, pad_h=pad, pad_w=pad, mask=act, mask_scale=2.0)
    got = torch.full_like(y0, float("nan"))
    ops.conv2d(x, w, None, got, R=k, S=k, stride=stride, pad_h=pad, pad_w=pad, mask=act, mask_scale=2.0, mask_bits=_pack_bits(act))
    assert torch.equal(got, want)


def test_stem_relu_gate_bits(ops):
    g = torch.Generator().manual_seed(24)
    B, H, W, Cout = 3, 37, 30, 64
    x = torch.rand(B, H, W, 1, generator=g).to(DEV)
    w = (torch.randn(Cout, 9, generator=g) / 3).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    y0 = torch.empty(B, H - 2, W - 2, Cout, dtype=torch.bfloat16, device=DEV)
    ops.conv_cin1_fwd(x, w, b, y0, relu=True)
    y1 = torch.empty_like(y0)
    bits = ops.relu_bits_like(y1)
    ops.conv_cin1_fwd(x, w, b, y1, relu=True, relu_bits_out=bits)
    assert torch.equal(y1, y0) and torch.equal(bits, _pack_bits(y1))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 128, 10, 9, 64), (1, 1024, 9, 9, 512)])
def test_convT2x2_fwd_scatter(ops, dtype, B, Cin, H, W, Cout):
    g = torch.Generator().manual_seed(3)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(Cin, Cout, 2, 2, generator=g) / math.sqrt(Cin), dtype)   # torch ConvTranspose2d layout
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv_transpose2d(x, w, b, stride=2))
    wp = w.permute(2, 3, 1, 0).reshape(4 * Cout, Cin).contiguous().to(dtype).to(DEV)    # [(a,b,co)][ci]
    y = torch.empty(B, 2 * H, 2 * W, Cout, dtype=dtype, device=DEV)
    ops.conv2d(to_dev(x, dtype), wp, b.to(DEV), y, R=1, S=1, relu=True, scatter2x2=True)
    close(to_cpu(y), ref, dtype, "convT scatter fwd")


# ------------------------------------------------------------------------------------ wgrad
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Cin,H,W,Cout,pad", [
    (2, 64, 20, 18, 128, 0), (2, 64, 20, 18, 64, 0), (1, 128, 14, 14, 128, 0),
    (2, 128, 60, 60, 64, 0),      # many pixel chunks; bf16: filter-row kernel (wgrad3), one 58-pixel run per row
    (1, 64, 12, 12, 64, 1),       # padded conv
    (2, 64, 34, 130, 64, 0),      # wgrad3: two full 64-pixel runs per row, 64 x 64 tile
    (1, 128, 24, 124, 128, 0),    # wgrad3: runs of 64 + 58, 128 x 128 tile (8 waves)
    (2, 64, 20, 62, 128, 1),      # wgrad3: padded conv -- the x strip starts outside the image
    (1, 128, 12, 254, 64, 0),     # wgrad3: four runs per row (64, 64, 64, 60), 64 x 128 tile
])
def test_conv2d_wgrad(ops, dtype, B, Cin, H, W, Cout, pad):
    g = torch.Generator().manual_seed(4)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype).requires_grad_(False)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, padding=pad)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    (ref,) = torch.autograd.grad(y, w, dy)
    old = torch.randn(Cout, 3, 3, Cin, generator=g)
    dw = old.clone().to(DEV)
    ops.conv2d_wgrad(to_dev(dy, dtype), to_dev(x, dtype), dw, pad_h=pad, pad_w=pad, accumulate=True)
    close(dw.cpu() - old, ref.permute(0, 2, 3, 1), dtype, "wgrad accumulate", rtol32=3e-4)
    dw2 = torch.full_like(dw, float("nan"))
    ops.conv2d_wgrad(to_dev(dy, dtype), to_dev(x, dtype), dw2, pad_h=pad, pad_w=pad, accumulate=False)
    close(dw2.cpu(), ref.permute(0, 2, 3, 1), dtype, "wgrad overwrite", rtol32=3e-4)
    if dtype == torch.bfloat16:      # fused bias gradient (column sums of dy) in the same launch, both epilogue modes
        bref = dy.sum((0, 2, 3))
        for acc in (True, False):
            dw3 = old.clone().to(DEV) if acc else torch.full_like(dw, float("nan"))
            oldb = torch.randn(Cout, generator=g)
            db = oldb.clone().to(DEV) if acc else torch.full((Cout,), float("nan"), device=DEV)
            ops.conv2d_wgrad(to_dev(dy, dtype), to_dev(x, dtype), dw3, pad_h=pad, pad_w=pad, accumulate=acc, db=db)
            close(dw3.cpu() - (old if acc else 0), ref.permute(0, 2, 3, 1), dtype, "wgrad+bias: dw")
            close(db.cpu() - (oldb if acc else 0), bref, dtype, "wgrad+bias: db", rtol16=1e-2)


@pytest.mark.parametrize("B,Cin,H,W,Cout,pad", [
    (2, 64, 29, 29, 64, 0),       # 27-wide dy: two image rows per K-step (pitch 29), odd row count -> a one-row last step
    (2, 64, 16, 16, 128, 0),      # 14-wide: four rows per step
    (1, 128, 13, 11, 64, 1),      # padded, 11-wide: five rows per step, strip starts outside the image
    (3, 64, 27, 48, 64, 0),       # 46-wide: one row per step
])
def test_conv2d_wgrad_filter_row_kernel_narrow_images(ops, B, Cin, H, W, Cout, pad):
    """wgrad3_kernel's packed-rows mode, forced also on layers the planner would leave to the per-tap kernel
    (dct_tune_set(DCT_TUNE_WGRAD_ROWS_FILL, 30)); same reference as test_conv2d_wgrad."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(14)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, padding=pad)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    (ref,) = torch.autograd.grad(y, w, dy)
    lib = _lib.load()
    lib.dct_tune_set(9, 30)
    try:
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=DEV)
        db = torch.full((Cout,), float("nan"), device=DEV)
        ops.conv2d_wgrad(to_dev(dy, dtype), to_dev(x, dtype), dw, pad_h=pad, pad_w=pad, accumulate=False, db=db)
    finally:
        lib.dct_tune_set(9, 70)
    close(dw.cpu(), ref.permute(0, 2, 3, 1), dtype, "filter-row wgrad, packed rows")
    close(db.cpu(), dy.sum((0, 2, 3)), dtype, "filter-row wgrad, packed rows: db", rtol16=1e-2)


@pytest.mark.parametrize("B,Cin,H,W,Cout,pad", [(4, 64, 130, 132, 64, 0), (3, 128, 70, 101, 128, 1), (8, 256, 27, 27, 128, 0), (16, 128, 18, 16, 256, 2)])
def test_conv2d_wgrad_filter_row_window_vs_per_tap_kernel(ops, B, Cin, H, W, Cout, pad):
    """wgrad3_kernel: a row's three x fragments come from ONE 12-pixel window per lane (tap 2 = the window moved by a dword, tap 1 =
    four v_alignbit_b32).  Against autograd, against the per-tap kernel (DCT_TUNE_WGRAD_ROWS = 0: other K order, so last-bit
    differences only), on wide images (runs of a row) and narrow ones (packed rows); a second launch reproduces the first."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(15)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, padding=pad)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    (ref,) = torch.autograd.grad(y, w, dy)
    lib = _lib.load()
    lib.dct_tune_set(9, 30)
    outs = []
    try:
        for rows in (1, 0, 1):
            lib.dct_tune_set(8, rows)
            dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=DEV)
            db = torch.full((Cout,), float("nan"), device=DEV)
            ops.conv2d_wgrad(to_dev(dy, dtype), to_dev(x, dtype), dw, pad_h=pad, pad_w=pad, accumulate=False, db=db)
            torch.cuda.synchronize()
            outs.append((dw, db))
    finally:
        lib.dct_tune_set(9, 70)
        lib.dct_tune_set(8, 1)
    close(outs[0][0].cpu(), ref.permute(0, 2, 3, 1), dtype, "filter-row wgrad, window form")
    assert torch.equal(outs[2][0], outs[0][0]) and torch.equal(outs[2][1], outs[0][1])
    scale = outs[1][0].abs().max().item()
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-4 * scale


@pytest.mark.parametrize("dtype", DTYPES)
def test_convT_wgrad_and_dgrad(ops, dtype):
    g = torch.Generator().manual_seed(5)
    B, Cin, H, W, Cout = 2, 128, 10, 9, 64
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype).requires_grad_(True)
    w = q(torch.randn(Cin, Cout, 2, 2, generator=g) / 16, dtype).requires_grad_(True)
    y = F.conv_transpose2d(x, w, stride=2)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    rx, rw = torch.autograd.grad(y, [x, w], dy)
    # wgrad: P = x, Q = dy, taps 2x2 stride 2 -> [ci][a][b][co]
    dw = torch.empty(Cin, 2, 2, Cout, device=DEV)
    ops.conv2d_wgrad(to_dev(x.detach(), dtype), to_dev(dy, dtype), dw, R=2, S=2, stride=2)
    close(dw.cpu(), rw.permute(0, 2, 3, 1), dtype, "convT wgrad", rtol32=3e-4)
    # dgrad: 2x2 stride-2 conv of dy with weights [ci][a][b][co]
    wd = w.detach().permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
    dx = torch.empty(B, H, W, Cin, dtype=dtype, device=DEV)
    ops.conv2d(to_dev(dy, dtype), wd, None, dx, R=2, S=2, stride=2)
    close(to_cpu(dx), rx, dtype, "convT dgrad")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_dgrad_via_packed_weights(ops, dtype):
    g = torch.Generator().manual_seed(6)
    B, Cin, H, W, Cout = 2, 64, 14, 12, 128
    x = torch.zeros(B, Cin, H, W, requires_grad=True)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / 24, dtype)
    y = F.conv2d(x, w)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    (ref,) = torch.autograd.grad(y, x, dy)
    master = w.permute(0, 2, 3, 1).contiguous().to(DEV)                # fp32 [co][r][s][ci]
    wd = torch.empty(Cin, 3, 3, Cout, dtype=dtype, device=DEV)
    ops.pack_weight(master, wd, Cout, 9, Cin, transpose=True, flip_taps=True)
    dx = torch.empty(B, H, W, Cin, dtype=dtype, device=DEV)
    ops.conv2d(to_dev(dy, dtype), wd, None, dx, pad_h=2, pad_w=2)
    close(to_cpu(dx), ref, dtype, "conv dgrad")
    wf = torch.empty(Cout, 3, 3, Cin, dtype=dtype, device=DEV)
    ops.pack_weight(master, wf, Cout, 9, Cin)
    assert torch.equal(wf.float().cpu(), w.permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize("edge_ok", [True, False])
def test_pack_weights_batched(ops, edge_ok):
    """All transposed packs of a network in one launch (arch/unet.py::_ensure_packs): the 16-bit 64 x 64-tile kernel with
    16-byte accesses (dct_pack_weights_batched64) when every job allows it, the 32 x 32 kernel otherwise -- bit-exact moves
    against torch.permute for both pack layouts (conv dgrad [ci][flipped taps][co], convT forward [(a,b,co)][ci])."""
    g = torch.Generator().manual_seed(16)
    shapes = [(128, 9, 64, 1, True), (64, 9, 192, 1, True), (256, 4, 128, 2, False)] if edge_ok else [(96, 9, 64, 1, True), (64, 4, 32, 2, False)]
    jobs, want = [], []
    for P, T, Q, mode, flip in shapes:
        src = torch.randn(P, T, Q, generator=g).to(torch.bfloat16).to(DEV)
        dst = torch.empty(P * T * Q, dtype=torch.bfloat16, device=DEV)
        jobs.append((src, dst, P, T, Q, mode, flip))
        t = src.flip(1) if flip else src
        want.append(t.permute(2, 1, 0).contiguous().flatten() if mode == 1 else t.permute(1, 2, 0).contiguous().flatten())
    table, n, tiles, edge = ops.pack_jobs_table(jobs, torch.device(DEV))
    assert edge == (64 if edge_ok else 32)
    ops.pack_weights_batched(table, n, tiles, torch.bfloat16, edge)
    torch.cuda.synchronize()
    for (src, dst, *_), w in zip(jobs, want):
        assert torch.equal(dst, w)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [64, 128, 512])
def test_bias_grad(ops, dtype, C):
    g = torch.Generator().manual_seed(7)
    dy = q(torch.randn(3, C, 33, 17, generator=g), dtype)
    old = torch.randn(C, generator=g)
    db = old.clone().to(DEV)
    ops.bias_grad(to_dev(dy, dtype), db, accumulate=True)
    close(db.cpu(), old + dy.sum((0, 2, 3)), dtype, "bias grad", rtol16=2e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_bias_grad_batched_is_bit_identical_to_single_calls(ops, dtype, accumulate):
    """dct_bias_grad_batched (a UNet's four up-convolutions' bias gradients in one launch pair) against one dct_bias_grad per tensor: every
    bit.  The tensors are channel slices of wider ones (the decoder's concatenations), of different sizes and channel counts."""
    g = torch.Generator().manual_seed(17)
    shapes = [(3, 44, 40, 64), (3, 24, 24, 128), (2, 14, 14, 256), (2, 9, 9, 512)]
    dys, olds = [], []
    for (B, H, W, Cc) in shapes:
        wide = to_dev(q(torch.randn(B, 2 * Cc, H, W, generator=g), dtype), dtype)        # NHWC physical, [..., :Cc] is a strided slice
        dys.append(wide[..., :Cc])
        olds.append(torch.randn(Cc, generator=g).to(DEV))
    want = [o.clone() for o in olds]
    for dy, db in zip(dys, want):
        ops.bias_grad(dy, db, accumulate=accumulate)
    got = [o.clone() for o in olds]
    ops.bias_grad_batched(dys, got, accumulate=accumulate)
    torch.cuda.synchronize()
    for a, b, dy, o in zip(got, want, dys, olds):
        assert torch.equal(a, b)
        ref = dy.float().sum((0, 1, 2)).cpu() + (o.cpu() if accumulate else 0)
        close(a.cpu(), ref, dtype, "batched bias grad", rtol16=2e-4)


# ------------------------------------------------------------------------------------ stem / head
@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_cin1(ops, dtype):
    g = torch.Generator().manual_seed(8)
    B, H, W, Cout = 2, 30, 26, 64
    x = torch.rand(B, 1, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, 1, 3, 3, generator=g) / 3).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    y = F.conv2d(x, w, b)
    yr = F.relu(y)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    yd = torch.empty(B, H - 2, W - 2, Cout, dtype=dtype, device=DEV)
    ops.conv_cin1_fwd(xd, w.detach().reshape(Cout, 9).contiguous().to(DEV), b.detach().to(DEV), yd, relu=True)
    close(to_cpu(yd), yr, dtype, "stem fwd", rtol16=8e-3)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    rx, rw, rb = torch.autograd.grad(y, [x, w, b], dy)
    dx = torch.empty(B, H, W, 1, device=DEV)
    ops.conv_cin1_dgrad(to_dev(dy, dtype), w.detach().reshape(Cout, 9).contiguous().to(DEV), dx, pad_h=0, pad_w=0)
    close(to_cpu(dx), rx, torch.float32, "stem dgrad")
    dw = torch.zeros(Cout, 9, device=DEV)
    db = torch.zeros(Cout, device=DEV)
    ops.conv_cin1_wgrad(xd, to_dev(dy, dtype), dw, db)
    close(dw.cpu(), rw.reshape(Cout, 9), torch.float32, "stem wgrad")
    close(db.cpu(), rb, torch.float32, "stem bgrad")


@pytest.mark.parametrize("B,H,W", [(3, 37, 30), (2, 50, 67), (5, 130, 66), (1, 19, 16)])
def test_stem_dgrad_mfma_form(ops, B, H, W):
    """bf16 stem data gradient (the FGSM pass's d/dx) on the matrix pipe (csrc/pointwise.hip stem_dgrad_mfma_kernel: per-pixel tap products as a
    GEMM with the fp32 weights split into three bf16 parts, nine LDS reads per pixel) against F.conv_transpose2d in fp64 and against the
    vector-ALU kernel it replaces (knob 1008 = 0): ragged 16 x 16 tiles, image borders, several images."""
    from dct_amd import _lib
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(83)
    Cout = 64
    w = torch.randn(Cout, 1, 3, 3, generator=g) / 3
    dy = q(torch.randn(B, Cout, H - 2, W - 2, generator=g), dtype)
    ref = F.conv_transpose2d(dy.double(), w.double()).float()          # d/dx of a valid 3x3 convolution
    wd = w.reshape(Cout, 9).contiguous().to(DEV)
    lib = _lib.load()
    outs = []
    try:
        for form in (1, 0):
            assert lib.dct_tune_set(1008, form) == 0
            dx = torch.full((B, H, W, 1), float("nan"), device=DEV)
            ops.conv_cin1_dgrad(to_dev(dy, dtype), wd, dx, pad_h=0, pad_w=0)
            torch.cuda.synchronize()
            outs.append(to_cpu(dx))
    finally:
        lib.dct_tune_set(1008, 1)
    scale = ref.abs().max().item()
    assert (outs[0] - ref).abs().max().item() <= 3e-6 * scale, "matrix-pipe form vs fp64 reference"
    assert (outs[1] - ref).abs().max().item() <= 3e-6 * scale, "vector-ALU form vs fp64 reference"


@pytest.mark.parametrize("B,H,W,pad", [(3, 37, 30, 0), (2, 50, 67, 1), (16, 66, 130, 0), (1, 19, 16, 2)])
def test_stem_wgrad_mfma_form(ops, B, H, W, pad):
    """bf16 stem weight gradient on the matrix pipe (csrc/reduce.hip stem_wgrad_mfma_kernel): units of 16 pixels of a row through the
    transposing LDS read, x split into bf16 high + low parts -- held to the fp32 tolerance of the VALU kernel it replaces, on ragged
    widths (the last unit of a row is partly empty), padded windows and many blocks, with and without accumulation."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(81)
    Cout = 64
    x = torch.rand(B, 1, H, W, generator=g)
    w = torch.zeros(Cout, 1, 3, 3, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    y = F.conv2d(x, w, b, padding=pad)
    dy = q(torch.randn(y.shape, generator=g), dtype)
    rw, rb = torch.autograd.grad(y, [w, b], dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw = torch.full((Cout, 9), 0.5, device=DEV)
    db = torch.full((Cout,), -2.0, device=DEV)
    ops.conv_cin1_wgrad(xd, to_dev(dy, dtype), dw, db, pad_h=pad, pad_w=pad, accumulate=True)
    close(dw.cpu() - 0.5, rw.reshape(Cout, 9), torch.float32, "stem wgrad (MFMA), accumulated", rtol32=3e-4)
    close(db.cpu() + 2.0, rb, torch.float32, "stem bgrad (MFMA), accumulated", rtol32=3e-4)
    dw2, db2 = torch.full_like(dw, float("nan")), torch.full_like(db, float("nan"))
    ops.conv_cin1_wgrad(xd, to_dev(dy, dtype), dw2, db2, pad_h=pad, pad_w=pad, accumulate=False)
    close(dw2.cpu(), rw.reshape(Cout, 9), torch.float32, "stem wgrad (MFMA)")
    close(db2.cpu(), rb, torch.float32, "stem bgrad (MFMA)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [2, 4])
def test_head(ops, dtype, C):
    g = torch.Generator().manual_seed(9)
    B, Cin, H, W = 2, 64, 21, 19
    x = q(F.relu(torch.randn(B, Cin, H, W, generator=g)), dtype).requires_grad_(True)
    w = (torch.randn(C, Cin, 1, 1, generator=g) / 8).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True)
    y = F.conv2d(x, w, b)
    yd = torch.empty(B, H, W, C, device=DEV)
    wd = w.detach().reshape(C, Cin).contiguous().to(DEV)
    ops.head_fwd(to_dev(x.detach(), dtype), wd, b.detach().to(DEV), yd)
    close(to_cpu(yd), y, torch.float32, "head fwd")
    dy = torch.randn(y.shape, generator=g)
    rx, rw, rb = torch.autograd.grad(y, [x, w, b], dy)
    rx = rx * (x.detach() > 0)
    dx = torch.empty(B, H, W, Cin, dtype=dtype, device=DEV)
    dw = torch.zeros(C, Cin, device=DEV)
    db = torch.zeros(C, device=DEV)
    ops.head_bwd(to_dev(x.detach(), dtype), dy.permute(0, 2, 3, 1).contiguous().to(DEV), wd, dx, dw, db, relu_mask=True)
    close(to_cpu(dx), rx, dtype, "head dx", rtol16=8e-3)
    close(dw.cpu(), rw.reshape(C, Cin), torch.float32, "head dw")
    close(db.cpu(), rb, torch.float32, "head db")


# ------------------------------------------------------------------------------------ pool / bilinear / dropout
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W", [(12, 10), (57, 25), (29, 29)])
def test_maxpool(ops, dtype, H, W):
    g = torch.Generator().manual_seed(10)
    x = q(F.relu(torch.randn(2, 64, H, W, generator=g)), dtype).requires_grad_(True)
    y = F.max_pool2d(x, 2, stride=2, ceil_mode=True)
    yd = torch.empty(2, y.shape[2], y.shape[3], 64, dtype=dtype, device=DEV)
    xd = to_dev(x.detach(), dtype)
    ops.maxpool_fwd(xd, yd)
    assert torch.equal(to_cpu(yd), y.detach()), "maxpool fwd must be exact"
    dy = q(torch.randn(y.shape, generator=g), dtype)
    (rx,) = torch.autograd.grad(y, x, dy)
    dx = torch.empty_like(xd)
    ops.maxpool_bwd(xd, to_dev(dy, dtype), dx, relu_mask=False)
    # ties only happen at 0 after ReLU; compare where x > 0, and check the relu-masked variant everywhere
    pos = x.detach() > 0
    assert torch.equal(to_cpu(dx)[pos], rx[pos])
    ops.maxpool_bwd(xd, to_dev(dy, dtype), dx, relu_mask=True, scale=2.0)
    close(to_cpu(dx), rx * pos * 2.0, dtype, "maxpool bwd masked", rtol16=1e-6, rtol32=1e-7)
    # the routing codes kept by the forward pass give the same two gradients bit for bit, without x
    codes = torch.empty(yd.shape, dtype=torch.uint8, device=DEV)
    yc = torch.empty_like(yd)
    ops.maxpool_fwd(xd, yc, codes=codes)
    assert torch.equal(yc, yd)
    for masked, scale in ((False, 1.0), (True, 2.0)):
        want = ops.maxpool_bwd(xd, to_dev(dy, dtype), torch.empty_like(xd), relu_mask=masked, scale=scale)
        got = ops.maxpool_bwd(None, to_dev(dy, dtype), torch.full_like(xd, float("nan")), relu_mask=masked, scale=scale, codes=codes)
        assert torch.equal(got, want), (masked, scale)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,hin,win,hout,wout", [(64, 126, 126, 88, 88), (128, 13, 13, 18, 18), (4, 84, 84, 256, 256), (2, 17, 23, 40, 31), (64, 29, 29, 28, 28)])
def test_bilinear(ops, dtype, C, hin, win, hout, wout):
    if C < 8 and dtype == torch.bfloat16:
        pytest.skip("classifier-sized resize runs in fp32")
    g = torch.Generator().manual_seed(11)
    x = q(torch.randn(2, C, hin, win, generator=g), dtype).requires_grad_(True)
    y = F.interpolate(x, size=(hout, wout), mode="bilinear", align_corners=True)
    big = torch.zeros(2, hout, wout, C + 8, dtype=dtype, device=DEV)      # write into a channel slice (concat)
    ops.bilinear_fwd(to_dev(x.detach(), dtype), big[..., 8:])
    close(to_cpu(big[..., 8:]), y, dtype, "bilinear fwd", rtol32=1e-5, rtol16=8e-3)
    assert float(big[..., :8].abs().sum()) == 0.0
    dy = q(torch.randn(y.shape, generator=g), dtype)
    (rx,) = torch.autograd.grad(y, x, dy)
    dx = torch.empty(2, hin, win, C, dtype=dtype, device=DEV)
    ops.bilinear_bwd(to_dev(dy, dtype), dx)
    close(to_cpu(dx), rx, dtype, "bilinear bwd", rtol32=1e-5, rtol16=8e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bilinear_fwd_batched_is_bit_identical_to_single_calls(ops, dtype):
    """dct_bilinear_fwd_batched (a UNet's four skip-connection resizes in one launch) against one dct_bilinear_fwd per tensor: every bit.
    Destinations are channel slices of wider tensors (the decoder's concatenations); sizes shrink, grow and stay."""
    g = torch.Generator().manual_seed(31)
    cases = [(2, 64, 37, 41, 26, 30), (2, 128, 19, 19, 16, 16), (1, 256, 9, 11, 12, 12), (3, 64, 8, 8, 8, 8)]
    xs, wides = [], []
    for (B, Cc, hi, wi, ho, wo) in cases:
        xs.append(to_dev(q(torch.randn(B, Cc, hi, wi, generator=g), dtype), dtype))
        wides.append((B, ho, wo, 2 * Cc))
    want = [torch.zeros(*w, dtype=dtype, device=DEV) for w in wides]
    got = [torch.zeros(*w, dtype=dtype, device=DEV) for w in wides]
    for x, y in zip(xs, want):
        ops.bilinear_fwd(x, y[..., y.shape[3] // 2:])
    ops.bilinear_fwd_batched(xs, [y[..., y.shape[3] // 2:] for y in got])
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert b.float().abs().max().item() > 0 and torch.equal(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout(ops, dtype):
    x = q(torch.rand(2, 512, 25, 25) + 0.5, dtype)
    xd = to_dev(x, dtype)
    yd = torch.empty_like(xd)
    mask = torch.empty(xd.shape, dtype=torch.uint8, device=DEV)
    ops.dropout_fwd(xd, yd, 0.5, seed=1234, offset=0, mask_out=mask)
    keep = mask.float().mean().item()
    assert abs(keep - 0.5) < 0.01, keep
    ref = xd.float() * mask.float() * 2.0
    close(yd.float(), ref, dtype, "dropout apply", rtol16=8e-3)
    y2 = torch.empty_like(xd)
    ops.dropout_fwd(xd, y2, 0.5, seed=1234, offset=0)
    assert torch.equal(y2, yd), "same (seed, offset) must reproduce the mask"
    ops.dropout_fwd(xd, y2, 0.5, seed=1234, offset=xd.numel())
    assert not torch.equal(y2, yd)
    y3 = torch.empty_like(xd)
    ops.dropout_apply(xd, y3, mask, 0.5)
    assert torch.equal(y3, yd)


def test_dropout_device_counter_advances_itself(ops):
    """The graph-replayable form (dct_dropout_fwd_dev): launch k of a counter that starts at c draws the mask of offset (c + k) << 40 and leaves
    c + k in the counter word the NEXT launch reads (the two words are used in turn) -- over 40 back-to-back launches of two sizes."""
    for shape in ((1, 512, 3, 3), (2, 512, 25, 25)):
        xd = to_dev(torch.rand(*shape) + 0.5, torch.bfloat16)
        state = torch.tensor([6, 6], dtype=torch.int64, device=DEV)
        got, want = torch.empty_like(xd), torch.empty_like(xd)
        for k in range(1, 41):
            ops.dropout_fwd(xd, got, 0.5, seed=99, offset=0, calls_dev=state, parity=(k - 1) & 1)
            if k in (1, 2, 17, 40):
                ops.dropout_fwd(xd, want, 0.5, seed=99, offset=(6 + k) << 40)
                assert torch.equal(got, want), k
        assert state.tolist() == [46, 45]


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_and_pool_in_one_pass_is_bit_identical(ops, dtype):
    """dct_dropout_maxpool2x2_fwd_codes against dct_dropout_fwd_dev + dct_maxpool2x2_fwd_codes from the same counter: pooled values, routing codes
    and the counter word the next launch reads -- odd sizes (ceil-mode edge windows), two successive calls."""
    x = to_dev(q(torch.randn(2, 512, 25, 23), dtype), dtype)
    hp, wp = 13, 12
    for start in (0, 7):
        s_a = torch.tensor([start, start], dtype=torch.int64, device=DEV)
        s_b = s_a.clone()
        for parity in (0, 1):
            dropped = torch.empty_like(x)
            ops.dropout_fwd(x, dropped, 0.5, seed=4242, offset=0, calls_dev=s_a, parity=parity)
            want_p = torch.empty(2, hp, wp, 512, dtype=dtype, device=DEV)
            want_c = torch.empty(2, hp, wp, 512, dtype=torch.uint8, device=DEV)
            ops.maxpool_fwd(dropped, want_p, codes=want_c)
            got_p, got_c = torch.empty_like(want_p), torch.empty_like(want_c)
            ops.dropout_maxpool_fwd(x, got_p, got_c, 0.5, 4242, s_b, parity)
            torch.cuda.synchronize()
            assert torch.equal(got_p, want_p) and torch.equal(got_c, want_c) and got_p.float().abs().max().item() > 0
            assert s_a.tolist() == s_b.tolist()


@pytest.mark.parametrize("dtype", DTYPES)
def test_relu_bwd_and_cast(ops, dtype):
    g = torch.Generator().manual_seed(12)
    a = q(torch.randn(2, 64, 9, 7, generator=g), dtype)
    gr = q(torch.randn(2, 64, 9, 7, generator=g), dtype)
    out = torch.empty(2, 9, 7, 64, dtype=dtype, device=DEV)
    ops.relu_bwd(to_dev(gr, dtype), to_dev(a, dtype), out, scale=1.0)
    assert torch.equal(to_cpu(out), gr * (a > 0))
    src = torch.randn(2, 5, 6, 16, device=DEV)
    dst = torch.empty(2, 5, 6, 16, dtype=torch.bfloat16, device=DEV)
    ops.cast(src, dst)
    assert torch.equal(dst, src.to(torch.bfloat16))


# ------------------------------------------------------------------------------------ losses
def _pc(t_nchw):
    return t_nchw.permute(0, 2, 3, 1).contiguous().reshape(-1, t_nchw.shape[1]).to(DEV)


def _back(t_pc, like):
    B, C, H, W = like.shape
    return t_pc.cpu().reshape(B, H, W, C).permute(0, 3, 1, 2)


@pytest.mark.parametrize("C", [2, 4, 3])
def test_ce(ops, C):
    g = torch.Generator().manual_seed(13)
    x = (torch.randn(3, C, 17, 19, generator=g) * 3).requires_grad_(True)
    t = torch.randint(0, C, (3, 17, 19), generator=g)
    t[0, :3] = 255
    ref = oracle.cross_entropy_2d(x, t)
    (rg,) = torch.autograd.grad(ref, x)
    xd, td = _pc(x.detach()), t.reshape(-1).to(DEV)
    out = ops.ce_fwd(xd, td, C)
    np.testing.assert_allclose(out[0].item(), ref.item(), rtol=2e-6)
    assert out[1].item() == (t != 255).sum().item()
    gs = torch.tensor([0.5], device=DEV)
    d = torch.empty_like(xd)
    ops.ce_bwd(xd, td, C, out[1:2], d, gscale=gs, gmul=2.0)
    np.testing.assert_allclose(_back(d, x).numpy(), rg.numpy(), rtol=1e-5, atol=1e-9)


def test_losses_match_golden(ops, golden):
    """the reference's own numbers (tests/golden/g1_losses.npz) through the HIP kernels"""
    g1 = golden("g1_losses")
    torch.manual_seed(int(g1["seed"]))
    a, b, c = (torch.randn(2, 4, 8, 8) for _ in range(3))
    t = torch.randint(0, 4, (2, 8, 8))
    ad, bd, cd = _pc(a), _pc(b), _pc(c)
    np.testing.assert_allclose(ops.ce_fwd(ad, t.reshape(-1).to(DEV), 4)[0].item(), g1["ce"], rtol=2e-6)
    np.testing.assert_allclose(ops.jsd_logits_fwd([ad, bd], 4).item(), g1["jsd2_map"].mean(), rtol=1e-5)
    np.testing.assert_allclose(ops.jsd_logits_fwd([ad, bd, cd], 4).item(), g1["jsd3_map"].mean(), rtol=1e-5)
    np.testing.assert_allclose(ops.kl_logits_fwd(ad, bd, 4).item(), g1["kl"], rtol=1e-5)
    pa, pb = ops.softmax_fwd(ad, 4), ops.softmax_fwd(bd, 4)
    np.testing.assert_allclose(ops.jsd_map_fwd([pa, pb], 4).cpu().reshape(2, 8, 8).numpy(), g1["jsd2_map"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(ops.kl_map_fwd(pa, pb, 4).cpu().reshape(2, 8, 8).numpy(), g1["kl_map"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(ops.entropy_fwd(pa, 4).cpu().reshape(2, 8, 8).numpy(), g1["entropy_a"], rtol=1e-5, atol=1e-7)
    da, db_ = torch.zeros_like(ad), torch.zeros_like(bd)
    ops.jsd_logits_bwd([ad, bd], 4, [da, db_])
    np.testing.assert_allclose(_back(da, a).numpy(), g1["jsd2_grad_a"], rtol=2e-4, atol=1e-9)
    np.testing.assert_allclose(_back(db_, a).numpy(), g1["jsd2_grad_b"], rtol=2e-4, atol=1e-9)
    dk = torch.zeros_like(ad)
    ops.kl_logits_bwd(ad, bd, 4, dk)
    np.testing.assert_allclose(_back(dk, a).numpy(), g1["kl_grad_a"], rtol=2e-4, atol=1e-9)


def test_multiview_jsd_matches_golden(ops, golden):
    """JSD over 4 and 6 models -- the reference's multi-view sweeps (script/GM/run_multiview.sh:2-6) -- against the maps and logit
    gradients captured from the reference's JSD_2D (tests/golden/g1_multiview.npz), through both kernel forms."""
    g = golden("g1_multiview")
    torch.manual_seed(int(g["seed"]))
    xs = [torch.randn(2, 3, 9, 7) * 1.5 for _ in range(6)]
    for S in (4, 6):
        lds = [_pc(x) for x in xs[:S]]
        np.testing.assert_allclose(ops.jsd_logits_fwd(lds, 3).item(), g[f"jsd{S}_map"].mean(), rtol=1e-5)
        pds = [ops.softmax_fwd(l, 3) for l in lds]
        np.testing.assert_allclose(ops.jsd_map_fwd(pds, 3).cpu().reshape(2, 9, 7).numpy(), g[f"jsd{S}_map"], rtol=1e-4, atol=1e-7)
        dls = [torch.zeros_like(l) for l in lds]
        ops.jsd_logits_bwd(lds, 3, dls)
        for k in range(S):
            np.testing.assert_allclose(_back(dls[k], xs[0]).numpy(), g[f"jsd{S}_grad_{k}"], rtol=2e-4, atol=1e-9)
    # the module API (what CoTrainer's unfused path and user code call)
    from dct_amd.loss import JSD_2D
    probs = [torch.softmax(x.to(DEV), 1).requires_grad_(True) for x in xs]
    jm = JSD_2D()(probs)
    np.testing.assert_allclose(jm.detach().cpu().numpy(), g["jsd6_map"], rtol=1e-4, atol=1e-7)
    jm.mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in probs)


@pytest.mark.parametrize("S,C", [(2, 4), (3, 2), (4, 4), (6, 2), (8, 3)])
def test_jsd_kl_vs_oracle(ops, S, C):
    g = torch.Generator().manual_seed(14)
    ls = [(torch.randn(2, C, 33, 31, generator=g) * 2).requires_grad_(True) for _ in range(S)]
    probs = [oracle.softmax_channels(l) for l in ls]
    jm = oracle.jsd_2d(probs)
    j = jm.mean()
    gr = torch.autograd.grad(j, ls, retain_graph=True)
    lds = [_pc(l.detach()) for l in ls]
    np.testing.assert_allclose(ops.jsd_logits_fwd(lds, C).item(), j.item(), rtol=2e-5)
    dls = [torch.randn_like(l) for l in lds]
    olds = [d.clone() for d in dls]
    gs = torch.tensor([0.25], device=DEV)
    ops.jsd_logits_bwd(lds, C, dls, gscale=gs, gmul=4.0, accumulate=True)
    for s in range(S):
        # (old + g) - old re-rounds at |old| ~ 1: allow one ulp of the accumulator
        np.testing.assert_allclose(_back(dls[s] - olds[s], ls[0]).numpy(), gr[s].numpy(), rtol=3e-4, atol=5e-7)
    # module-API variants (probs in, maps out)
    pds = [ops.softmax_fwd(l, C) for l in lds]
    np.testing.assert_allclose(_back(pds[0], ls[0]).numpy(), probs[0].detach().numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(ops.jsd_map_fwd(pds, C).cpu().numpy(), jm.detach().reshape(-1).numpy(), rtol=2e-4, atol=2e-7)
    dmap = torch.randn(jm.shape, generator=g)
    gp = torch.autograd.grad(jm, probs, dmap, retain_graph=True)
    dps = ops.jsd_map_bwd(pds, dmap.reshape(-1).to(DEV), C)
    for s in range(S):
        np.testing.assert_allclose(_back(dps[s], ls[0]).numpy(), gp[s].numpy(), rtol=3e-4, atol=1e-6)
    # softmax backward
    (gl,) = torch.autograd.grad(probs[0], ls[0], gp[0], retain_graph=True)
    np.testing.assert_allclose(_back(ops.softmax_bwd(pds[0], dps[0], C), ls[0]).numpy(), gl.numpy(), rtol=3e-4, atol=1e-6)
    # KL
    k = oracle.kl_divergence_2d(probs[0], probs[1].detach(), reduce=True)
    (gk,) = torch.autograd.grad(k, ls[0], retain_graph=True)
    np.testing.assert_allclose(ops.kl_logits_fwd(lds[0], lds[1], C).item(), k.item(), rtol=2e-5)
    dk = torch.zeros_like(lds[0])
    ops.kl_logits_bwd(lds[0], lds[1], C, dk)
    np.testing.assert_allclose(_back(dk, ls[0]).numpy(), gk.numpy(), rtol=3e-4, atol=2e-9)
    km = oracle.kl_divergence_2d(probs[0], probs[1].detach())
    (gkp,) = torch.autograd.grad(km, probs[0], dmap)
    np.testing.assert_allclose(_back(ops.kl_map_bwd(pds[0], pds[1], dmap.reshape(-1).to(DEV), C), ls[0]).numpy(), gkp.numpy(), rtol=3e-4, atol=1e-6)


def test_argmax_fgsm_dice(ops):
    g = torch.Generator().manual_seed(15)
    x = torch.randn(3, 4, 16, 16, generator=g)
    assert torch.equal(ops.argmax(_pc(x), 4).cpu().reshape(3, 16, 16), x.argmax(1))
    img = torch.rand(1000, generator=g)
    gr = torch.randn(1000, generator=g)
    gr[::7] = 0
    xa, nz = ops.fgsm_step(img.to(DEV), gr.to(DEV), 0.03)
    assert torch.equal(nz.cpu(), 0.03 * gr.sign())
    assert torch.equal(xa.cpu(), img + 0.03 * gr.sign())
    gt = torch.randint(0, 4, (3, 1, 16, 16), generator=g)
    inter, ps, gsm = ops.dice_counts(_pc(x).reshape(3, 256, 4), gt.reshape(3, 256).to(DEV), 3, 4)
    d2 = (2 * inter.float() + 1e-8) / ((ps + gsm).float() + 1e-8)
    np.testing.assert_allclose(d2.cpu().numpy(), oracle.dice_2d(x, gt).numpy(), rtol=1e-6)
    d3 = (2 * inter.sum(0).float() + 1e-8) / ((ps.sum(0) + gsm.sum(0)).float() + 1e-8)
    np.testing.assert_allclose(d3.cpu().numpy(), oracle.dice_3d(x, gt).numpy(), rtol=1e-6)


def test_adam_flat(ops):
    g = torch.Generator().manual_seed(16)
    n = 100003
    p = torch.randn(n, generator=g)
    m = torch.zeros(n)
    v = torch.zeros(n)
    pd, md, vd = (t.clone().to(DEV) for t in (p, m, v))
    shadow = torch.empty(n + 5, dtype=torch.bfloat16, device=DEV)[:n]
    pt = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([pt], lr=1e-3, weight_decay=1e-4)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * 0.1
        oracle.adam_reference_step(p, gr, m, v, step)
        pt.grad = gr.clone()
        opt.step()
        bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
        ops.adam_flat(pd, gr.to(DEV), md, vd, 1e-3 / bc1, math.sqrt(bc2), 0.9, 0.999, 1e-8, 1e-4, bf16_shadow=shadow)
    np.testing.assert_allclose(p.numpy(), pt.detach().numpy(), rtol=1e-6, atol=1e-8)   # oracle adam == torch.optim.Adam
    np.testing.assert_allclose(pd.cpu().numpy(), p.numpy(), rtol=2e-6, atol=2e-8)
    np.testing.assert_allclose(md.cpu().numpy(), m.numpy(), rtol=2e-6, atol=1e-8)   # fma contraction: ~2 ulp
    np.testing.assert_allclose(vd.cpu().numpy(), v.numpy(), rtol=2e-6, atol=1e-12)
    assert torch.equal(shadow.cpu(), pd.cpu().to(torch.bfloat16))


def test_loss_kernels_are_exact_beside_the_conv_kernels(ops):
    """Regression for the packed-FP32 hazard (csrc/Makefile NOPK): the reductions of the loss kernels launched on one
    stream while MFMA convolutions run on another give bit for bit what they give on a quiet device.  (Before the library was built
    without v_pk_*_f32, the JSD came out 5-50 % low beside a stream of dct_conv2d launches in 59 of 60 tries.)"""
    g = torch.Generator(device=DEV).manual_seed(3)
    C = 4
    lps = [torch.randn(8, 256, 256, C, device=DEV, generator=g) * 0.3 for _ in range(3)]
    tgt = torch.randint(0, C, (8 * 256 * 256,), device=DEV)
    x = torch.randn(16, 124, 124, 128, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(128, 3, 3, 128, device=DEV, generator=g) / 34).to(torch.bfloat16)
    y = torch.empty(16, 122, 122, 128, device=DEV, dtype=torch.bfloat16)
    dw = torch.zeros(128 * 9 * 128, device=DEV)
    victims = {"jsd": lambda: ops.jsd_logits_fwd(lps[:2], C)[0], "jsd3": lambda: ops.jsd_logits_fwd(lps, C)[0],
               "kl": lambda: ops.kl_logits_fwd(lps[0], lps[1], C)[0], "ce": lambda: ops.ce_fwd(lps[0].reshape(-1, C), tgt, C)[0],
               "entropy": lambda: ops.entropy_fwd(torch.softmax(lps[0], 3).contiguous(), C).double().sum()}
    quiet = {}
    for k, f in victims.items():
        quiet[k] = float(f())
        torch.cuda.synchronize()
    sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    for k, f in victims.items():
        for it in range(25):
            with torch.cuda.stream(sb):
                for _ in range(6):
                    ops.conv2d(x, w, None, y, relu=True)
            with torch.cuda.stream(sc):
                for _ in range(2):
                    ops.conv2d_wgrad(y, x, dw, accumulate=True)
            with torch.cuda.stream(sa):
                v = f()
            torch.cuda.synchronize()
            assert float(v) == quiet[k], (k, it, float(v), quiet[k])


@pytest.mark.parametrize("n,off", [(1, 0), (3, 1), (4, 0), (7, 3), (1000, 0), (1001, 2), (1 << 20, 0), ((1 << 20) + 5, 1), (31_000_003, 3)])
def test_flat_scale(ops, n, off):
    """dct_flat_scale: x *= s over any 4-byte aligned slice (the gradient average after a SUM all-reduce, ddp.py): exactly the
    fp32 product, and nothing outside the slice is touched."""
    torch.manual_seed(n)
    base = torch.randn(n + off + 9, device=DEV)
    ref = base.clone()
    ops.flat_scale(base[off:off + n], 1.0 / 3.0)
    ref[off:off + n] = ref[off:off + n].cpu().mul(torch.tensor(1.0 / 3.0, dtype=torch.float32)).to(DEV)
    torch.cuda.synchronize()
    assert torch.equal(base, ref)

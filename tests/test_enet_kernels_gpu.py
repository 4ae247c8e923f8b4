"""Per-kernel parity of the Enet C-ABI entry points (dct_enet_*) against plain PyTorch fp32 CPU math, in
fp32 storage (parity path) and in the mixed bf16 mode (block outputs / gradients bf16, raw conv outputs
fp32) -- the mixed mode is compared with the same math on bf16-rounded inputs."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def K():
    from dct_amd import hip_ops
    return hip_ops


def q(t, dt):
    return t.to(dt).float()


def nhwc(t, dt):
    return t.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, ref, dt, what, r32=1e-4, r16=2e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    rtol = r32 if dt == torch.float32 else r16
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs()
    bad = err > rtol * scale + rtol * ref.abs()
    assert not bad.any() and torch.isfinite(got).all(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max():.3e}, scale {scale:.3e}"


def make_tf(K, C, mode, g):
    scale = (torch.rand(C, generator=g) + 0.5)
    shift = torch.randn(C, generator=g) * 0.3
    slope = torch.rand(C, generator=g) * 0.5
    tf = K.Tf(scale.to(DEV), shift.to(DEV), slope.to(DEV) if mode == 2 else None, mode)

    def apply(x):   # x NCHW
        z = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if mode == 2:
            z = torch.where(z > 0, z, z * slope.view(1, -1, 1, 1))
        elif mode == 3:
            z = z.relu()
        return z
    return tf, apply, (scale, shift, slope)


def mfma_form(dt, cin, cout, taps):
    """dct_enet_conv's MFMA form (csrc/enet.hip::enet_mconv_kernel): low-precision modes, >= 16 input channels in whole 8-channel
    groups, K = taps * cin in whole steps of 16.  It rounds BOTH operands of the contraction to the compute dtype (fp32 accumulate);
    the VALU form multiplies the stored values in fp32.  dct_enet_wgrad's MFMA form (enet_mwgrad_kernel) takes every shape."""
    return dt != torch.float32 and cin >= 16 and cin % 8 == 0 and (taps * cin) % 16 == 0 and cout <= 128


def set_mfma(on):
    from dct_amd import _lib
    _lib.check(_lib.load().dct_tune_set(26, int(on)), "dct_tune_set(ENET_MFMA)")


@pytest.mark.parametrize("dt,mfma", [(torch.float32, 3), (torch.bfloat16, 3), (torch.float16, 3), (torch.bfloat16, 0)])
@pytest.mark.parametrize("cin,cout,kh,kw,stride,pad,dil,in_f32,mode", [
    (64, 16, 1, 1, 1, (0, 0), 1, False, 0),      # block1x1_1 on a block output
    (16, 16, 3, 3, 1, (1, 1), 1, True, 2),       # middle 3x3 on a raw input through BN+PReLU
    (32, 32, 3, 3, 1, (4, 4), 4, True, 2),       # dilated
    (32, 32, 5, 1, 1, (2, 0), 1, True, 3),       # asymmetric 5x1
    (16, 64, 1, 1, 1, (0, 0), 1, True, 2),       # block1x1_2: two 32-channel MFMA column tiles
    (32, 128, 1, 1, 1, (0, 0), 1, True, 3),      # ... four
    (24, 40, 3, 3, 1, (1, 1), 1, False, 0),      # ragged: K = 216 is not a multiple of 16 (VALU form), 40 channels out
    (32, 24, 3, 3, 2, (1, 1), 1, False, 0),      # strided 3x3, a partly filled column tile
    (32, 40, 1, 1, 2, (0, 0), 1, False, 0),      # strided 1x1, a partly filled second column tile
    (14, 16, 2, 2, 2, (0, 0), 1, False, 0),      # down-sampling 2x2 s2
    (1, 13, 3, 3, 2, (1, 1), 1, True, 0),        # initial conv on the fp32 image
    (3, 14, 1, 1, 1, (0, 0), 1, True, 3),        # tiny widths of the last up-sampling bottleneck
])
def test_enet_conv_direct_dgrad_wgrad(K, dt, mfma, cin, cout, kh, kw, stride, pad, dil, in_f32, mode):
    g = torch.Generator().manual_seed(1)
    B, H, W = 2, 12, 10
    xs = torch.float32 if in_f32 else dt
    x = q(torch.randn(B, cin, H, W, generator=g), xs)
    w = torch.randn(cout, cin, kh, kw, generator=g) / math.sqrt(cin * kh * kw)
    bias = torch.randn(cout, generator=g)
    tf, apply, _ = make_tf(K, cin, mode, g) if mode else (None, (lambda t: t), None)
    xin = apply(x)
    fq = dt if (mfma & 1 and mfma_form(dt, cin, cout, kh * kw)) else torch.float32  # operand rounding of the forward contraction
    bq = dt if (mfma & 1 and mfma_form(dt, cout, cin, kh * kw)) else torch.float32  # ... of the data gradient (reduces over cout)
    wq = dt if mfma & 2 else torch.float32                                          # ... of the weight gradient (any shape)
    conv = lambda a, b, bb=None: F.conv2d(a, b, bb, stride=stride, padding=pad, dilation=dil)
    ref = conv(q(xin, fq), q(w, fq), bias)
    wk = w.permute(0, 2, 3, 1).contiguous().to(DEV)          # [co][kh][kw][ci]
    y = torch.empty(B, ref.shape[2], ref.shape[3], cout, dtype=torch.float32, device=DEV)   # raw outputs: fp32
    xd = nhwc(x, xs)
    set_mfma(mfma)
    try:
        K.enet_conv(xd, wk, bias.to(DEV), tf, y, R=kh, S=kw, stride=stride, dil=dil, pad_h=pad[0], pad_w=pad[1],
                    ws=(kh * kw * cin, cin, 1), compute=dt)
        close(nchw(y), ref, torch.float32, "conv fwd", r32=2e-4)
        gy = q(torch.randn(ref.shape, generator=g), dt)
        xr = xin.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        gx_ref, = torch.autograd.grad(conv(xr, q(w, bq)), xr, gy)
        gw_ref, = torch.autograd.grad(conv(q(xin, wq), wr), wr, gy)
        gyd = nhwc(gy, dt)
        dx = torch.empty(B, H, W, cin, dtype=dt, device=DEV)
        K.enet_conv(gyd, wk, None, None, dx, R=kh, S=kw, stride=stride, dil=dil, pad_h=pad[0], pad_w=pad[1], transposed=True,
                    ws=(1, cin, kh * kw * cin))
        close(nchw(dx), gx_ref, dt, "conv dgrad")
        dx32 = torch.empty(B, H, W, cin, dtype=torch.float32, device=DEV)       # fp32 destination: the rounding is the operands' only
        K.enet_conv(gyd, wk, None, None, dx32, R=kh, S=kw, stride=stride, dil=dil, pad_h=pad[0], pad_w=pad[1], transposed=True,
                    ws=(1, cin, kh * kw * cin))
        close(nchw(dx32), gx_ref, torch.float32, "conv dgrad (fp32 out)", r32=2e-4)
        dw = torch.zeros(cout * kh * kw * cin, device=DEV)
        K.enet_wgrad(gyd, None, xd, tf, dw, R=kh, S=kw, stride=stride, dil=dil, pad_h=pad[0], pad_w=pad[1])
        K.enet_wgrad(gyd, None, xd, tf, dw, R=kh, S=kw, stride=stride, dil=dil, pad_h=pad[0], pad_w=pad[1])   # "+=" twice
        close(dw.view(cout, kh, kw, cin).permute(0, 3, 1, 2).cpu(), 2 * gw_ref, torch.float32, "conv wgrad", r32=5e-4)
        db = torch.zeros(cout, device=DEV)
        K.enet_channel_sum(gyd, db)
        close(db.cpu(), gy.sum((0, 2, 3)), torch.float32, "bias grad", r32=1e-4)
    finally:
        set_mfma(3)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin,cout,k,pad,opad,in_f32", [(16, 16, 3, 1, 1, True), (32, 16, 3, 1, 1, False), (3, 3, 3, 1, 1, True), (14, 4, 2, 0, 0, False)])
def test_enet_convT(K, dt, cin, cout, k, pad, opad, in_f32):
    g = torch.Generator().manual_seed(2)
    B, H, W = 2, 9, 7
    xs = torch.float32 if in_f32 else dt
    x = q(torch.randn(B, cin, H, W, generator=g), xs).requires_grad_(True)
    w = (torch.randn(cin, cout, k, k, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    bias = torch.randn(cout, generator=g)
    fq = dt if mfma_form(dt, cin, cout, k * k) else torch.float32
    bq = dt if mfma_form(dt, cout, cin, k * k) else torch.float32
    ref = F.conv_transpose2d(q(x.detach(), fq), q(w.detach(), fq), bias, stride=2, padding=pad, output_padding=opad)
    wk = w.detach().permute(0, 2, 3, 1).contiguous().to(DEV)     # [ci][kh][kw][co]
    y = torch.empty(B, ref.shape[2], ref.shape[3], cout, dtype=torch.float32, device=DEV)
    xd = nhwc(x.detach(), xs)
    K.enet_conv(xd, wk, bias.to(DEV), None, y, R=k, S=k, stride=2, pad_h=pad, pad_w=pad, transposed=True, ws=(1, cout, k * k * cout), compute=dt)
    close(nchw(y), ref.detach(), torch.float32, "convT fwd", r32=2e-4)
    gy = q(torch.randn(ref.shape, generator=g), dt)
    convT = lambda a, b: F.conv_transpose2d(a, b, None, stride=2, padding=pad, output_padding=opad)
    gx_ref, = torch.autograd.grad(convT(x, q(w.detach(), bq)), x, gy)
    gw_ref, = torch.autograd.grad(convT(q(x.detach(), dt), w), w, gy)       # the MFMA weight gradient rounds the layer input too
    gyd = nhwc(gy, dt)
    dx = torch.empty(B, H, W, cin, dtype=dt, device=DEV)
    K.enet_conv(gyd, wk, None, None, dx, R=k, S=k, stride=2, pad_h=pad, pad_w=pad, ws=(k * k * cout, cout, 1))
    close(nchw(dx), gx_ref, dt, "convT dgrad")
    dw = torch.zeros(cin * k * k * cout, device=DEV)
    K.enet_wgrad(xd, None, gyd, None, dw, R=k, S=k, stride=2, pad_h=pad, pad_w=pad)
    close(dw.view(cin, k, k, cout).permute(0, 3, 1, 2).cpu(), gw_ref, torch.float32, "convT wgrad", r32=5e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C,act", [(16, 2), (64, 3), (13, 2), (3, 3), (64, 0)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(2, 9, 11), (3, 41, 37)])
def test_enet_bn_fwd_bwd(K, dt, C, act, training, shape):
    """BatchNorm statistics / apply / backward: split reduction + one-block fold + apply (whole 8-channel groups take the vector
    reduction kernel, the 13- and 3-channel cases the scalar one)."""
    _bn_fwd_bwd(K, dt, C, act, training, shape)


def _bn_fwd_bwd(K, dt, C, act, training, shape):
    g = torch.Generator().manual_seed(3)
    B, H, W = shape
    raw = (torch.randn(B, C, H, W, generator=g) * 2 + 3).requires_grad_(True)       # raw conv outputs: fp32, mean >> 0
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    slope = (torch.rand(C, generator=g) * 0.5).requires_grad_(True)
    rm0, rv0 = torch.randn(C, generator=g) * 0.1 + 3, torch.rand(C, generator=g) + 3.5
    rm, rv = rm0.clone(), rv0.clone()
    z = F.batch_norm(raw, rm, rv, gamma, beta, training, 0.1, 1e-3)
    a = torch.where(z > 0, z, z * slope.view(1, -1, 1, 1)) if act == 2 else (z.relu() if act == 3 else z)
    rawd = nhwc(raw.detach(), torch.float32)
    vec = torch.empty(4, C, device=DEV)
    rmd, rvd = rm0.to(DEV), rv0.to(DEV)
    K.enet_bn_fwd_stats(rawd, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-3, 0.1, rmd, rvd, training, vec[0], vec[1], vec[2], vec[3])
    tf = K.Tf(vec[0], vec[1], slope.detach().to(DEV) if act == 2 else None, {2: 2, 3: 3, 0: 1}[act])
    zz = rawd * vec[0] + vec[1]
    close(nchw(zz), z.detach(), torch.float32, "bn scale/shift", r32=2e-5)
    if training:
        close(rmd.cpu(), rm, torch.float32, "running mean", r32=1e-5)
        close(rvd.cpu(), rv, torch.float32, "running var", r32=1e-5)
    # backward, gated by a ReLU mask as in the bottleneck tail
    up = q(torch.randn(B, C, H, W, generator=g), dt)
    mask = q(torch.randn(B, C, H, W, generator=g), dt)
    a.backward(up * (mask > 0))
    dg, db, ds = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    draw = torch.empty(B, H, W, C, dtype=dt, device=DEV)
    K.enet_bn_bwd(rawd, nhwc(up, dt), nhwc(mask, dt), tf, vec[2], vec[3], dg, db, ds if act == 2 else None,
                  torch.empty(2 * C, device=DEV), draw, training=training)
    close(nchw(draw), raw.grad, dt, "bn bwd draw", r32=3e-4, r16=2e-2)
    close(dg.cpu(), gamma.grad, torch.float32, "dgamma", r32=3e-4)
    close(db.cpu(), beta.grad, torch.float32, "dbeta", r32=3e-4)
    if act == 2:
        close(ds.cpu(), slope.grad, torch.float32, "dslope", r32=3e-4)


@pytest.mark.parametrize("dt", DTYPES)
def test_enet_tails(K, dt):
    g = torch.Generator().manual_seed(4)
    B, h, w = 2, 6, 5
    # ---- regular
    C = 64
    raw = torch.randn(B, C, h, w, generator=g)
    x = q(torch.randn(B, C, h, w, generator=g), dt)
    tf, apply, _ = make_tf(K, C, 2, g)
    ref = (x + apply(raw)).relu()
    out = torch.empty(B, h, w, C, dtype=dt, device=DEV)
    K.enet_tail_fwd(nhwc(raw, torch.float32), tf, nhwc(x, dt), None, None, None, 0, 0, out)
    close(nchw(out), q(ref, dt), dt, "tail regular", r16=1e-2)
    # ---- down: maxpool with indices + zero channel pad; and its backward routing
    Cm, Co = 14, 64
    xin = q(torch.randn(B, Cm, 2 * h, 2 * w, generator=g), dt)
    raw = torch.randn(B, Co, h, w, generator=g)
    tf, apply, _ = make_tf(K, Co, 2, g)
    pooled, idx_ref = F.max_pool2d(xin, 2, stride=2, return_indices=True)
    main = torch.cat([pooled, torch.zeros(B, Co - Cm, h, w)], 1)
    ref = (main + apply(raw)).relu()
    out = torch.empty(B, h, w, Co, dtype=dt, device=DEV)
    idx = torch.empty(B, h, w, Cm, dtype=torch.uint8, device=DEV)
    K.enet_tail_fwd(nhwc(raw, torch.float32), tf, nhwc(xin, dt), None, None, idx, Cm, 1, out)
    close(nchw(out), q(ref, dt), dt, "tail down", r16=1e-2)
    dout = q(torch.randn(B, Co, h, w, generator=g), dt)
    gz = dout * (q(ref, dt) > 0)
    dx_ref = F.max_unpool2d(gz[:, :Cm], idx_ref, 2)
    dx = torch.empty(B, 2 * h, 2 * w, Cm, dtype=dt, device=DEV)
    K.enet_tail_bwd(nhwc(dout, dt), out, idx, Cm, 1, dx)
    close(nchw(dx), dx_ref, dt, "tail down bwd")
    # ---- up: unpool of bn(rawm) with those indices
    rawm = torch.randn(B, Cm, h, w, generator=g)
    tfm, applym, _ = make_tf(K, Cm, 0, g)
    tfm.mode = 1
    raw2 = torch.randn(B, Cm, 2 * h, 2 * w, generator=g)
    tf2, apply2, _ = make_tf(K, Cm, 3, g)
    zm = rawm * tfm.scale.cpu().view(1, -1, 1, 1) + tfm.shift.cpu().view(1, -1, 1, 1)
    ref = (F.max_unpool2d(zm, idx_ref, 2) + apply2(raw2)).relu()
    out2 = torch.empty(B, 2 * h, 2 * w, Cm, dtype=dt, device=DEV)
    K.enet_tail_fwd(nhwc(raw2, torch.float32), tf2, None, nhwc(rawm, torch.float32), tfm, idx, Cm, 2, out2)
    close(nchw(out2), q(ref, dt), dt, "tail up", r16=1e-2)
    dout2 = q(torch.randn(B, Cm, 2 * h, 2 * w, generator=g), dt)
    gz2 = dout2 * (q(ref, dt) > 0)
    gm_ref = torch.gather(gz2.flatten(2), 2, idx_ref.flatten(2)).view(B, Cm, h, w)
    gm = torch.empty(B, h, w, Cm, dtype=dt, device=DEV)
    K.enet_tail_bwd(nhwc(dout2, dt), out2, idx, Cm, 2, gm)
    close(nchw(gm), gm_ref, dt, "tail up bwd")
    # ---- initial block
    img = torch.rand(B, 1, 2 * h, 2 * w, generator=g)
    raw13 = torch.randn(B, 13, h, w, generator=g)
    tf13, apply13, _ = make_tf(K, 13, 2, g)
    ref = torch.cat([apply13(raw13), F.max_pool2d(img, 2, stride=2)], 1)
    out14 = torch.empty(B, h, w, 14, dtype=dt, device=DEV)
    imgd = nhwc(img, torch.float32)
    K.enet_tail_fwd(nhwc(raw13, torch.float32), tf13, imgd, None, None, None, 13, 3, out14)
    close(nchw(out14), q(ref, dt), dt, "initial tail", r16=1e-2)
    d14 = q(torch.randn(B, 14, h, w, generator=g), dt)
    _, iidx = F.max_pool2d(img, 2, stride=2, return_indices=True)
    dimg_ref = F.max_unpool2d(d14[:, 13:14], iidx, 2) + 1.0
    dimg = torch.ones(B, 2 * h, 2 * w, 1, device=DEV)
    K.enet_tail_bwd(nhwc(d14, dt), imgd, None, 13, 3, dimg, accumulate=True)
    close(nchw(dimg), dimg_ref, torch.float32, "initial tail bwd")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,k", [(32, 32, 3), (16, 64, 1), (32, 128, 1), (16, 24, 3)])
def test_enet_conv_epilogue_bn_statistics(K, dt, cin, cout, k):
    """dct_enet_conv_stats: the MFMA convolution's epilogue writes the consumer BatchNorm's partial sums (one row per tile of 32
    pixels, the reduction's [row][C][3] double layout) and dct_enet_bn_fwd_stats_rows folds them -- against the separate
    reduction over the stored tensor.  429 pixels: 14 tiles, the last one ragged."""
    g = torch.Generator().manual_seed(5)
    B, H, W = 3, 13, 11
    x = torch.randn(B, cin, H, W, generator=g) * 2 + 1
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=g)
    tf, _, _ = make_tf(K, cin, 2, g)
    gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(DEV), torch.randn(cout, generator=g).to(DEV)
    wk = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    xd = nhwc(x, torch.float32)
    kw = dict(R=k, S=k, pad_h=k // 2, pad_w=k // 2, ws=(k * k * cin, cin, 1), compute=dt)
    y1 = torch.empty(B, H, W, cout, dtype=torch.float32, device=DEV)
    K.enet_conv(xd, wk, bias.to(DEV), tf, y1, **kw)
    v1 = torch.empty(5, cout, device=DEV)
    K.enet_bn_fwd_stats(y1, gamma, beta, 1e-3, 0.1, None, None, True, v1[0], v1[1], v1[2], v1[3], save_var=v1[4])
    tiles = (B * H * W + 31) // 32
    stats = torch.full((tiles * cout * 3,), float("nan"), dtype=torch.float64, device=DEV)
    y2 = torch.empty_like(y1)
    rows = K.enet_conv_stats(xd, wk, bias.to(DEV), tf, y2, stats, **kw)
    assert rows == tiles == 14
    assert torch.equal(y1, y2)
    st = stats.view(tiles, cout, 3)
    assert torch.isfinite(st).all() and (st[:, :, 2] == 0).all()
    yp = y1.reshape(-1, cout).double()
    np.testing.assert_allclose(st[:, :, 0].sum(0).cpu().numpy(), yp.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(st[:, :, 1].sum(0).cpu().numpy(), (yp * yp).sum(0).cpu().numpy(), rtol=1e-5)
    v2 = torch.empty(5, cout, device=DEV)
    K.enet_bn_fwd_stats(y2, gamma, beta, 1e-3, 0.1, None, None, True, v2[0], v2[1], v2[2], v2[3], save_var=v2[4], partial=stats, partial_rows=rows)
    np.testing.assert_allclose(v2.cpu().numpy(), v1.cpu().numpy(), rtol=2e-5, atol=2e-6)
    # a tensor with more tiles than the scratch holds: nothing written, the caller reduces as usual
    small = torch.empty(5 * cout * 3, dtype=torch.float64, device=DEV)
    assert K.enet_conv_stats(xd, wk, bias.to(DEV), tf, y2, small, **kw) == 0 and torch.equal(y1, y2)
    # fp32 mode keeps the VALU kernel: no rows either
    assert K.enet_conv_stats(xd, wk, bias.to(DEV), tf, y2, stats, **dict(kw, compute=torch.float32)) == 0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cg,cx,k,act", [(64, 16, 1, 2), (32, 32, 3, 2), (128, 32, 1, 3), (16, 24, 3, 0)])
def test_enet_dgrad_epilogue_bn_backward_sums(K, dt, cg, cx, k, act):
    """dct_enet_conv_bnbwd_stats: the data-gradient convolution's epilogue writes the BatchNorm-backward partial sums of the layer
    whose activation gradient it produces; dct_enet_bn_bwd_rows then folds and applies -- against the separate reduction."""
    g0 = torch.Generator().manual_seed(6)
    B, H, W = 3, 13, 11
    gy = q(torch.randn(B, cg, H, W, generator=g0), dt)                         # gradient wrt the conv's output
    w = torch.randn(cg, cx, k, k, generator=g0) / math.sqrt(cx * k * k)         # the forward conv cx -> cg
    raw = torch.randn(B, cx, H, W, generator=g0) * 2 + 1                        # producing layer's raw (pre-BatchNorm) output
    gamma, beta = torch.rand(cx, generator=g0) + 0.5, torch.randn(cx, generator=g0)
    slope = torch.rand(cx, generator=g0) * 0.5
    wk = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    gyd, rawd = nhwc(gy, dt), nhwc(raw, torch.float32)
    vec = torch.empty(5, cx, device=DEV)
    K.enet_bn_fwd_stats(rawd, gamma.to(DEV), beta.to(DEV), 1e-3, 0.1, None, None, True, vec[0], vec[1], vec[2], vec[3], save_var=vec[4])
    tf = K.Tf(vec[0], vec[1], slope.to(DEV) if act == 2 else None, {2: 2, 3: 3, 0: 1}[act])
    kw = dict(R=k, S=k, pad_h=k // 2, pad_w=k // 2, transposed=True, ws=(1, cx, k * k * cx), compute=dt)

    def grads():
        return torch.zeros(cx, device=DEV), torch.zeros(cx, device=DEV), torch.zeros(cx, device=DEV)
    # separate: dgrad, then reduction + fold + apply
    g1 = torch.empty(B, H, W, cx, dtype=dt, device=DEV)
    K.enet_conv(gyd, wk, None, None, g1, **kw)
    dg1, db1, ds1 = grads()
    d1 = K.enet_bn_bwd(rawd, g1, None, tf, vec[2], vec[3], dg1, db1, ds1 if act == 2 else None, torch.empty(2 * cx, device=DEV),
                       torch.empty(B, H, W, cx, dtype=dt, device=DEV))
    # fused
    tiles = (B * H * W + 31) // 32
    stats = torch.full((tiles * cx * 3,), float("nan"), dtype=torch.float64, device=DEV)
    g2 = torch.empty_like(g1)
    rows = K.enet_conv_bnbwd_stats(gyd, wk, g2, stats, rawd, tf, vec[2], vec[3], **kw)
    assert rows == tiles and torch.equal(g1, g2) and torch.isfinite(stats).all()
    dg2, db2, ds2 = grads()
    d2 = K.enet_bn_bwd(rawd, g2, None, tf, vec[2], vec[3], dg2, db2, ds2 if act == 2 else None, torch.empty(2 * cx, device=DEV),
                       torch.empty(B, H, W, cx, dtype=dt, device=DEV), partial=stats, partial_rows=rows)
    close(dg2, dg1, torch.float32, "dgamma", r32=2e-5)
    close(db2, db1, torch.float32, "dbeta", r32=2e-5)
    if act == 2:
        close(ds2, ds1, torch.float32, "dslope", r32=2e-5)
    close(d2, d1, dt, "draw", r32=1e-5, r16=1e-2)


def test_bn_running_update_and_flat_sum():
    """One launch for the BatchNorm bookkeeping of a whole network (dct_bn_running_update) == nn.BatchNorm2d's
    r <- (1 - m) r + m b and num_batches_tracked += 1 per layer; dct_flat_sum == ((a + b) + c) bit for bit."""
    from dct_amd import hip_ops as K
    g = torch.Generator().manual_seed(21)
    cs = [13, 16, 64, 128, 5]
    offs, off = [], 0
    for c in cs:
        offs.append(off)
        off += (c + 3) // 4 * 4
    stats = torch.randn(5 * off, generator=g).to(DEV)
    layers, want = [], []
    for c, o in zip(cs, offs):
        rm, rv = torch.randn(c, generator=g).to(DEV), (torch.rand(c, generator=g) + 0.5).to(DEV)
        nbt = torch.tensor(7, dtype=torch.int64, device=DEV)
        cp, base = (c + 3) // 4 * 4, 5 * o
        layers.append((rm, rv, nbt, c, base + 2 * cp, base + 4 * cp))
        want.append((0.9 * rm.cpu().double() + 0.1 * stats[base + 2 * cp:base + 2 * cp + c].cpu().double(),
                     0.9 * rv.cpu().double() + 0.1 * stats[base + 4 * cp:base + 4 * cp + c].cpu().double()))
    table = K.bn_running_table(layers, torch.device(DEV))
    K.bn_running_update(table, len(layers), stats, 0.1)
    torch.cuda.synchronize()
    for (rm, rv, nbt, *_), (wm, wv) in zip(layers, want):
        np.testing.assert_allclose(rm.cpu().numpy(), wm.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rv.cpu().numpy(), wv.numpy(), rtol=1e-6, atol=1e-7)
        assert int(nbt) == 8
    a, b, c = (torch.randn(4096 + 64, generator=g).to(DEV) for _ in range(3))
    out = torch.empty_like(a)
    assert torch.equal(K.flat_sum(out, a, b, c), (a + b) + c)
    assert torch.equal(K.flat_sum(out, a, b), a + b)

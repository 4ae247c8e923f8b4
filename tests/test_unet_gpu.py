"""UNet HIP plan vs the reference's golden vectors (tests/golden/g3_unet.npz, captured from the
imported reference) and vs the CPU oracle: logits, d/dx, per-tensor weight-gradient norms.
fp32 mode carries the parity claim; bf16 mode is checked against the same oracle with a bf16
tolerance (stated per assert)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402

DEV = "cuda:0"


def _oracle_net(C, seed, p=0.0):
    torch.manual_seed(seed)
    return oracle.build_net("unet", C, dropout_p=p)


def _hip_net(onet, C, dtype, p=0.0):
    from dct_amd.arch import get_arch
    net = get_arch("unet", {"num_classes": C, "compute_dtype": dtype, "dropout_p": p})
    net.load_state_dict(onet.state_dict())
    return net.to(DEV)


def _rel(a, b):
    """max-norm relative error"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _rel2(a, b):
    """L2 relative error: robust to the isolated ReLU/max-pool decision flips that 1-ulp
    differences in a pre-activation cause (SURVEY.md 7 'chaotic parity points')"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


def test_unet_fp32_matches_reference_golden(golden):
    g = golden("g3_unet")
    C, seed = int(g["C"]), int(g["seed"])
    onet = _oracle_net(C, seed).eval()
    net = _hip_net(onet, C, torch.float32).eval()
    H = 176
    torch.manual_seed(100 + H)
    x = torch.rand(1, 1, H, H)
    t = torch.randint(0, C, (1, H, H))
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd)
    assert y.shape == (1, C, H, H)
    # fp32 tolerance: accumulation order differs from ATen's (K = 9*Cin up to 9216 terms)
    assert _rel(y.detach().cpu().numpy(), g["eval176_logits"]) < 5e-6
    loss = torch.nn.functional.cross_entropy(y.float(), t.to(DEV))  # torch CE only to seed the backward here
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["eval176_ce"], rtol=1e-5)
    assert _rel2(xd.grad.cpu().numpy(), g["eval176_grad_x"]) < 1e-5
    names = [k for k, _ in net.named_parameters()]
    assert names == list(g["eval176_grad_names"])
    norms = np.array([p.grad.double().norm().item() for _, p in net.named_parameters()])
    np.testing.assert_allclose(norms, g["eval176_grad_norms"], rtol=2e-5, atol=1e-12)


def test_unet_fp32_256_digest(golden):
    g = golden("g3_unet")
    C, seed = int(g["C"]), int(g["seed"])
    onet = _oracle_net(C, seed).eval()
    net = _hip_net(onet, C, torch.float32).eval()
    torch.manual_seed(100 + 256)
    x = torch.rand(1, 1, 256, 256)
    with torch.no_grad():
        y = net(x.to(DEV)).double().cpu().flatten()
    d = np.array([y.sum().item(), y.abs().sum().item(), y.norm().item(), y.abs().max().item()])
    np.testing.assert_allclose(d[1:], g["eval256_logits_digest"][1:], rtol=2e-5)


# Tolerances (L2-relative per tensor).
#  fp32: the kernels agree with ATen to ~1e-6 per layer (tools/debug_unet_layers.py), but ONE ReLU
#        decision on a pre-activation within an ulp of 0 flipping between CPU and GPU puts ~3e-3 on
#        every gradient downstream of it (measured: 1 flipped element of 50,176 in enc2 for the
#        200x200 seed below; 0 flips -> <1e-6 for the 176 golden case) => 5e-3.
#  bf16: activations AND back-propagated gradients are stored in bf16 (2^-9 relative per element,
#        plus many such flips); measured 0.5-1.5e-1 on the deepest (1e-6-magnitude) gradients => 0.25,
#        logits 1.4e-3 measured => 4e-2.
@pytest.mark.parametrize("dtype,tol_logit,tol_grad", [(torch.float32, 3e-6, 5e-3), (torch.bfloat16, 4e-2, 0.25)])
@pytest.mark.parametrize("B,H,W", [(2, 200, 200), (1, 184, 216)])
def test_unet_vs_oracle_fwd_bwd(dtype, tol_logit, tol_grad, B, H, W):
    C = 2 if H == 200 else 4
    onet = _oracle_net(C, 5).eval()
    net = _hip_net(onet, C, dtype).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 1, H, W, generator=g)
    t = torch.randint(0, C, (B, H, W), generator=g)
    xo = x.clone().requires_grad_(True)
    yo = onet(xo)
    lo = oracle.cross_entropy_2d(yo, t)
    lo.backward()
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd)
    assert _rel2(y.detach().float().cpu().numpy(), yo.detach().numpy()) < tol_logit
    # identical upstream gradient for both: d(CE)/d(logits) computed by the oracle
    yo2 = yo.detach().clone().requires_grad_(True)
    gl = torch.autograd.grad(oracle.cross_entropy_2d(yo2, t), yo2)[0]
    y.backward(gl.to(DEV))
    errs = {"grad_x": _rel2(xd.grad.cpu().numpy(), xo.grad.numpy())}
    for (k, p), (_, po) in zip(net.named_parameters(), onet.named_parameters()):
        errs[k] = _rel2(p.grad.cpu().numpy(), po.grad.numpy())
    # a second backward pass accumulates (the step back-propagates 2-3 graphs per model)
    y2 = net(xd)
    y2.backward(gl.to(DEV))
    for (k, p), (_, po) in zip(net.named_parameters(), onet.named_parameters()):
        errs["acc:" + k] = _rel2(p.grad.cpu().numpy(), 2 * po.grad.numpy())
    bad = {k: f"{v:.2e}" for k, v in errs.items() if not v < tol_grad}
    worst = max(errs, key=errs.get)
    assert not bad, f"L2-rel errors above {tol_grad}: {bad}; worst {worst}={errs[worst]:.2e}"


def test_unet_train_mode_dropout_replay():
    """GPU Philox masks cannot equal CPU masks: record the GPU masks and replay them in the oracle."""
    C = 4
    onet = _oracle_net(C, 6, p=0.5).train()
    net = _hip_net(onet, C, torch.float32, p=0.5).train()
    net.record_dropout_masks = True
    x = torch.rand(1, 1, 176, 176, generator=torch.Generator().manual_seed(4))
    xd = x.to(DEV)
    y = net(xd)
    masks = [m.permute(0, 3, 1, 2).float().cpu() for m in net.last_dropout_masks]
    assert len(masks) == 2 and all(abs(m.mean().item() - 0.5) < 0.05 for m in masks)
    yo = onet(x, dropout_masks=masks)
    assert _rel(y.detach().cpu().numpy(), yo.detach().numpy()) < 3e-5
    gl = torch.randn(yo.shape, generator=torch.Generator().manual_seed(5))
    yo.backward(gl)
    y.backward(gl.to(DEV))
    for (k, p), (_, po) in zip(net.named_parameters(), onet.named_parameters()):
        r = _rel(p.grad.cpu().numpy(), po.grad.numpy())
        assert r < 3e-4, f"{k}: rel err {r:.3e}"
    # two forwards draw different masks
    y2 = net(xd)
    assert not torch.equal(y2, y)
    # eval mode is deterministic and dropout-free
    net.eval()
    assert torch.equal(net(xd), net(xd))


def test_unet_gate_bits_and_pool_codes_leave_every_gradient_bit_identical():
    """The backward pass with ReLU-gate bits (instead of re-read activations) and max-pool routing codes (instead of the
    re-read pool input) gives bit for bit the gradients of the plan without them -- train mode, with dropout."""
    C = 4
    onet = _oracle_net(C, 7, p=0.5).train()
    x = torch.rand(2, 1, 192, 208, generator=torch.Generator().manual_seed(6)).to(DEV)
    gl = torch.randn(2, C, 192, 208, generator=torch.Generator().manual_seed(7)).to(DEV)
    grads = []
    for flag in (False, True):
        net = _hip_net(onet, C, torch.bfloat16, p=0.5).train()
        net.relu_bits = net.pool_codes = flag
        net.unpool_on_load = False          # (needs the codes: it would change the flow of one arm only)
        net.fuse_skip_grad = False      # (needs the codes; it changes the rounding of an intermediate: its own test below)
        net.dropout_seed = 99
        xd = x.clone().requires_grad_(True)
        y = net(xd)
        y.backward(gl)
        grads.append([y.detach().clone(), xd.grad.clone()] + [p.grad.clone() for p in net.parameters()])
    names = ["logits", "grad_x"] + [k for k, _ in net.named_parameters()]
    for k, a, b in zip(names, *grads):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unet_pooling_in_the_conv_call_leaves_everything_bit_identical(dtype):
    """Each encoder block's max pooling taken from its second convolution's call (dct_conv_desc.pool_out: out of the staged tile in
    the shared-halo kernel, a launch behind the conv elsewhere) against the separate pooling launch: logits, d/dx and every
    gradient bit for bit -- train mode (the fourth level keeps its own launch behind the dropout) and eval mode."""
    C = 4
    onet = _oracle_net(C, 13, p=0.5)
    x = torch.rand(2, 1, 200, 216, generator=torch.Generator().manual_seed(26)).to(DEV)
    gl = torch.randn(2, C, 200, 216, generator=torch.Generator().manual_seed(27)).to(DEV)
    for train in (True, False):
        outs = []
        for flag in (False, True):
            net = _hip_net(onet, C, dtype, p=0.5)
            net = net.train() if train else net.eval()
            net.fuse_pool = flag                # (with it, pool_only: the encoder blocks' full-resolution outputs are not stored)
            net.unpool_on_load = False          # (follows pool_only: it would sum the skip gradient in memory in one arm only)
            net.dropout_seed = 99
            xd = x.clone().requires_grad_(True)
            y = net(xd)
            y.backward(gl)
            outs.append([y.detach().clone(), xd.grad.clone()] + [p.grad.clone() for p in net.parameters()])
        names = ["logits", "grad_x"] + [k for k, _ in net.named_parameters()]
        for k, a, b in zip(names, *outs):
            assert torch.isfinite(a).all(), (train, k)
            assert torch.equal(a, b), (train, k)


def test_unet_stem_weight_gradient_from_the_data_gradient_tile():
    """The stem's weight / bias gradient taken from the epilogue of the second convolution's data gradient (whose output is then not
    stored) against the two separate launches: every other gradient bit for bit, the stem's two within fp32 rounding of each other."""
    C = 4
    onet = _oracle_net(C, 17, p=0.5).train()
    x = torch.rand(4, 1, 256, 256, generator=torch.Generator().manual_seed(36)).to(DEV)
    gl = torch.randn(4, C, 256, 256, generator=torch.Generator().manual_seed(37)).to(DEV)
    outs = []
    for flag in (False, True):
        net = _hip_net(onet, C, torch.bfloat16, p=0.5).train()
        net.fuse_stem_wgrad = flag
        net.dropout_seed = 99
        for _ in range(2):                      # the second pass accumulates
            y = net(x)
            y.backward(gl)
        outs.append([p.grad.clone() for p in net.parameters()])
    stem = 0
    for (k, prm), a, b in zip(net.named_parameters(), *outs):
        assert torch.isfinite(b).all(), k
        if k.startswith("dec1.down.0."):
            stem += 1
            assert (a - b).abs().max().item() <= 1e-4 * a.abs().max().item(), k
        else:
            assert torch.equal(a, b), k
    assert stem == 2, [k for k, _ in net.named_parameters()][:4]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_unet_skip_gradient_gathered_by_the_unpooling(dtype, tol):
    """The skip connections' bilinear backward gathered inside the un-pooling launch (dct_maxpool2x2_bwd_codes_skip) against the sum
    formed in memory (dct_bilinear_bwd into dp, the data gradient accumulating onto it): the same gradients up to the rounding of
    that intermediate -- fp32: to the order of the additions; bf16: the 16-bit rounding of dp that the fused form no longer makes."""
    C = 4
    onet = _oracle_net(C, 19, p=0.5).train()
    x = torch.rand(2, 1, 200, 216, generator=torch.Generator().manual_seed(46)).to(DEV)
    gl = torch.randn(2, C, 200, 216, generator=torch.Generator().manual_seed(47)).to(DEV)
    outs = []
    for flag in (False, True):
        net = _hip_net(onet, C, dtype, p=0.5).train()
        net.fuse_skip_grad = flag
        net.unpool_on_load = False          # (levels 1-3 would otherwise sum their skip gradient in memory whatever the flag)
        net.fuse_stem_wgrad = False
        net.dropout_seed = 99
        xd = x.clone().requires_grad_(True)
        y = net(xd)
        y.backward(gl)
        outs.append([xd.grad.clone()] + [p.grad.clone() for p in net.parameters()])
    names = ["grad_x"] + [k for k, _ in net.named_parameters()]
    changed = 0
    for k, a, b in zip(names, *outs):
        assert torch.isfinite(b).all(), k
        assert _rel2(b.cpu().numpy(), a.cpu().numpy()) <= tol, (k, _rel2(b.cpu().numpy(), a.cpu().numpy()))
        changed += int(not torch.equal(a, b))
    assert k and (dtype == torch.float32 or changed > 0)


@pytest.mark.parametrize("B,H,W,need_dx", [(2, 200, 216, True), (6, 256, 256, False)])
def test_unet_unpooled_gradients_expanded_on_load_are_bit_identical(B, H, W, need_dx):
    """Levels 1-3 of the encoder never write their un-pooled gradient (UNet.unpool_on_load): the block's data gradient and weight
    gradient expand {pooled gradient + routing codes} while they stage.  Against the same plan with the un-pooling launch in front of
    them (both with the skip gradient summed in memory, so that the pooled gradients are the same numbers): every gradient bit for bit
    -- whichever kernels take the expansion and whichever fall back to un-pooling into a buffer (the small levels of the first case)."""
    C = 4
    onet = _oracle_net(C, 23, p=0.5).train()
    x = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(48)).to(DEV)
    gl = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(49)).to(DEV)
    outs = []
    for flag in (False, True):
        net = _hip_net(onet, C, torch.bfloat16, p=0.5).train()
        net.unpool_on_load, net.unpool_max_level, net.fuse_skip_grad = flag, 3, False
        net.dropout_seed = 77
        _, tape = net.plan_forward(x, True)
        dx = net.plan_backward(tape, gl, need_dx=need_dx, need_dw=True, overwrite=True)
        torch.cuda.synchronize()
        outs.append(([dx.clone()] if need_dx else []) + [net.flat_params.gflat.clone()])
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and a.abs().max().item() > 0
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unet_launch_batching_is_bit_identical(dtype):
    """Round 5's launch-order switches: the transposed weight packs launched behind the encoder (UNet.late_packs) and the four up-convolutions'
    bias gradients in one launch pair (UNet.batch_bias_grads), the four skip connections' resizes in one launch (UNet.batch_skip_resize),
    the fourth level's dropout + max-pool in one launch (UNet.fuse_drop_pool), against the round-4 order: logits and every gradient bit for bit, over two
    optimizer-free passes (the second one re-uses the packs of the first)."""
    C, B, H, W = 4, 2, 192, 180
    onet = _oracle_net(C, 29, p=0.5).train()
    x = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(58)).to(DEV)
    gl = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(59)).to(DEV)
    outs = []
    for flag in (False, True):
        net = _hip_net(onet, C, dtype, p=0.5).train()
        net.late_packs = net.batch_bias_grads = net.batch_skip_resize = net.fuse_drop_pool = flag
        net.dropout_seed = 78
        got = []
        for rep in range(2):
            lp, tape = net.plan_forward(x, True)
            net.plan_backward(tape, gl, need_dx=False, need_dw=True, overwrite=True)
            torch.cuda.synchronize()
            got += [lp.clone(), net.flat_params.gflat.clone()]
        outs.append(got)
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and a.abs().max().item() > 0
        assert torch.equal(a, b)


def test_unet_second_pass_over_the_same_input_reuses_the_encoder_bit_for_bit():
    """plan_forward(reuse=tape of a pass over the same input tensor): the stem and the eight encoder convolutions in front of the first dropout come from
    that tape.  Against a net that runs both passes in full (same weights, same dropout seed and counter): the second pass's logits, its input gradient
    (the FGSM generator's use) and its weight gradients bit for bit; a different input tensor or changed weights fall back to the full pass."""
    C, B, H, W = 4, 2, 192, 180
    onet = _oracle_net(C, 31, p=0.5).train()
    x = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(68)).to(DEV)
    gl = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(69)).to(DEV)
    outs = []
    for share in (False, True):
        net = _hip_net(onet, C, torch.bfloat16, p=0.5).train()
        net.dropout_seed = 79
        lp1, t1 = net.plan_forward(x, True, keep_predrop=share)
        lp2, t2 = net.plan_forward(x, True, reuse=t1 if share else None)
        if share:
            assert t2["a1"] is t1["a1"] and t2["p3"] is t1["p3"]          # (the encoder was not run again)
        dx = net.plan_backward(t2, gl, need_dx=True, need_dw=True, overwrite=True)
        torch.cuda.synchronize()
        outs.append([lp1.clone(), lp2.clone(), dx.clone(), net.flat_params.gflat.clone()])
        if share:
            lp3, t3 = net.plan_forward(x.clone(), True, reuse=t1)             # another tensor: the full pass
            assert t3["a1"] is not t1["a1"]
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and a.abs().max().item() > 0
        assert torch.equal(a, b)
    assert not torch.equal(outs[0][0], outs[0][1])                             # (the two passes drew different masks)


def test_unet_rejects_small_and_cpu_inputs():
    from dct_amd.arch import get_arch
    net = get_arch("unet", {"num_classes": 4}).to(DEV)
    with pytest.raises(RuntimeError):
        net(torch.rand(1, 1, 64, 64, device=DEV))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 1, 176, 176))

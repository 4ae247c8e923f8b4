"""Enet HIP plan vs the reference's golden vectors (tests/golden/g3_enet.npz, captured from the
imported reference: train- and eval-mode logits, d/dx, per-tensor weight-gradient norms, BatchNorm
running statistics after one forward) and vs the CPU oracle.  fp32 mode carries the parity claim;
bf16 mode is checked against the same oracle with a bf16 tolerance (stated per assert)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from helpers import round_conv_operands  # noqa: E402

DEV = "cuda:0"


def _oracle_net(C, seed):
    torch.manual_seed(seed)
    return oracle.build_net("enet", C)


def _hip_net(onet, C, dtype):
    from dct_amd.arch import get_arch
    net = get_arch("enet", {"num_classes": C, "compute_dtype": dtype})
    net.load_state_dict(onet.state_dict())
    return net.to(DEV)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _rel2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_enet_fp32_matches_reference_golden(golden, mode):
    g = golden("g3_enet")
    C, seed = int(g["C"]), int(g["seed"])
    onet = _oracle_net(C, seed)
    net = _hip_net(onet, C, torch.float32)
    net.train() if mode == "train" else net.eval()
    B, H = 2, 64
    torch.manual_seed(100 + H)
    x = torch.rand(B, 1, H, H)
    t = torch.randint(0, C, (B, H, H))
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd)
    assert y.shape == (B, C, H, H)
    tag = f"{mode}{H}"
    # fp32 tolerance: 84 BatchNorm layers (double-accumulated statistics here, ATen's cascade sums there)
    assert _rel(y.detach().cpu().numpy(), g[f"{tag}_logits"]) < 2e-5
    loss = torch.nn.functional.cross_entropy(y.float(), t.to(DEV))   # torch CE only seeds the backward here
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"{tag}_ce"], rtol=1e-5)
    # ReLU / PReLU / max-pool decisions on values within an ulp of the threshold flip between ATen and the
    # kernels and put O(1e-3) on everything downstream (same effect as in tests/test_unet_gpu.py)
    assert _rel2(xd.grad.cpu().numpy(), g[f"{tag}_grad_x"]) < 5e-3
    names = [k for k, _ in net.named_parameters()]
    assert names == list(g[f"{tag}_grad_names"])
    norms = np.array([p.grad.double().norm().item() for _, p in net.named_parameters()])
    ref = g[f"{tag}_grad_norms"]
    # conv biases in front of a BatchNorm have a mathematically-zero gradient (pure rounding noise in the
    # reference, ~1e-9): compare those absolutely, everything else relatively
    big = ref > 1e-6
    np.testing.assert_allclose(norms[big], ref[big], rtol=1e-2)
    assert np.all(norms[~big] < 1e-5)
    if mode == "train":
        sd = net.state_dict()
        np.testing.assert_allclose(sd["encoder.initial.batch_norm.running_mean"].cpu().numpy(), g["train64_bn_init_mean"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(sd["encoder.initial.batch_norm.running_var"].cpu().numpy(), g["train64_bn_init_var"], rtol=1e-5)
        np.testing.assert_allclose(sd["decoder.layers.4.block1x1_2.1.running_mean"].cpu().numpy(), g["train64_bn_last_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sd["decoder.layers.4.block1x1_2.1.running_var"].cpu().numpy(), g["train64_bn_last_var"], rtol=1e-4)
        assert int(sd["encoder.initial.batch_norm.num_batches_tracked"]) == 1


def test_enet_fp32_256_digest(golden):
    g = golden("g3_enet")
    C, seed = int(g["C"]), int(g["seed"])
    net = _hip_net(_oracle_net(C, seed), C, torch.float32).eval()
    torch.manual_seed(100 + 256)
    x = torch.rand(1, 1, 256, 256)
    with torch.no_grad():
        y = net(x.to(DEV)).double().cpu().flatten()
    d = np.array([y.sum().item(), y.abs().sum().item(), y.norm().item(), y.abs().max().item()])
    np.testing.assert_allclose(d[1:], g["eval256_logits_digest"][1:], rtol=5e-5)


def _bf16_points(onet):
    """Round the oracle's block outputs to bf16 exactly where the bf16 plan stores bf16 (the output of the
    initial block and of every bottleneck; raw conv outputs and the image stay fp32 in both)."""
    def rnd(_m, _i, o):
        if isinstance(o, tuple):
            return (o[0].bfloat16().float(),) + tuple(o[1:])
        return o.bfloat16().float()
    for m in onet.modules():
        if m.__class__.__name__ in ("_Bottleneck", "_Initial"):
            m.register_forward_hook(rnd)
    return round_conv_operands(onet, torch.bfloat16)   # ... and the MFMA convolutions' operands


# Tolerances (L2-relative per tensor).
#  fp32: ReLU / PReLU / max-pool decision flips on values within an ulp of the threshold put O(1e-3) on
#        downstream gradients (see tests/test_unet_gpu.py).
#  bf16: a randomly initialised Enet in train mode is a strong amplifier -- rounding ONLY the input image to
#        bf16 moves the oracle's own logits by 12 % and rounding the block outputs by 15 % (L2; measured on
#        the CPU oracle).  The bf16 plan is therefore compared with the oracle run with the SAME rounding
#        points (block outputs -> bf16).  The first blocks then agree to 1e-7, but the un-normalised residual
#        stream grows to O(100) where one bf16 ulp is 0.5, and every rounding tie that fp32 noise flips is a
#        full ulp fed back through that amplifier: measured per block with tools/debug_enet_blocks.py, the
#        error climbs smoothly 1e-7 -> 9e-3 over the encoder and reaches 6-10 % at the logits => 0.2 on
#        logits.  Going back through the same amplifier the EARLY-layer gradients decorrelate completely: on
#        the CPU oracle alone, a 1e-6 relative perturbation in front of the bf16 rounding (tie flips only)
#        changes its first-layer gradients by 60-70 % and its logits by 10 %, its last-block gradients by
#        1-6 %.  In bf16 mode only the final transposed conv's gradients are compared (=> 0.2) and the rest
#        must be finite; every kernel is checked on its own in mixed bf16 mode by
#        tests/test_enet_kernels_gpu.py.  The parity claim is the fp32 mode.
#  The fp32 oracle itself is part of that chaos: ATen's CPU reductions change their summation order with the thread count, and
#  on the single-image case below the fp32 oracle's OWN gradients move by 5e-3 .. 1.7e-2 between 8 and 16+ threads.  So the
#  fp32 (parity) mode is held to the oracle evaluated in FLOAT64 -- same modules, same weights, rounding noise 1e-16: a
#  deterministic reference whatever the host -- and the gate stays at 5e-3 (round 2 had widened it to 3e-2
#  against the thread-dependent fp32 oracle).
@pytest.mark.parametrize("dtype,tol_logit,tol_grad", [(torch.float32, 2e-5, 5e-3), (torch.bfloat16, 0.2, 0.2)])
@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 2), (1, 96, 128, 4)])
def test_enet_vs_oracle_fwd_bwd_train(dtype, tol_logit, tol_grad, B, H, W, C):
    onet = _oracle_net(C, 7).train()
    net = _hip_net(onet, C, dtype).train()
    f64 = dtype == torch.float32
    if dtype == torch.bfloat16:
        _bf16_points(onet)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 1, H, W, generator=g)
    t = torch.randint(0, C, (B, H, W), generator=g)
    if f64:
        onet = onet.double()
    xo = (x.double() if f64 else x.clone()).requires_grad_(True)
    yo = onet(xo)
    oracle.cross_entropy_2d(yo, t).backward()
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd)
    assert _rel2(y.detach().float().cpu().numpy(), yo.detach().float().numpy()) < tol_logit
    yo2 = yo.detach().clone().requires_grad_(True)
    gl = torch.autograd.grad(oracle.cross_entropy_2d(yo2, t), yo2)[0]
    y.backward(gl.float().to(DEV))
    errs = {"grad_x": _rel2(xd.grad.cpu().numpy(), xo.grad.float().numpy())}
    for (k, p), (_, po) in zip(net.named_parameters(), onet.named_parameters()):
        if po.grad.norm() < 1e-6:      # biases in front of a BatchNorm: zero gradient up to rounding
            assert p.grad.norm().item() < (1e-5 if dtype == torch.float32 else 1e-2), k
            continue
        errs[k] = _rel2(p.grad.cpu().numpy(), po.grad.float().numpy())
    assert all(np.isfinite(v) for v in errs.values())
    if dtype == torch.bfloat16:
        errs = {k: v for k, v in errs.items() if k.startswith("decoder.layers.5")}
        assert len(errs) == 2
    bad = {k: f"{v:.2e}" for k, v in errs.items() if not v < tol_grad}
    worst = max(errs, key=errs.get)
    assert not bad, f"L2-rel errors above {tol_grad}: {bad}; worst {worst}={errs[worst]:.2e}"
    # running statistics of every BatchNorm after the one forward
    if dtype == torch.float32:
        sd, so = net.state_dict(), onet.state_dict()
        for k in so:
            if k.endswith("running_mean") or k.endswith("running_var"):
                np.testing.assert_allclose(sd[k].cpu().numpy(), so[k].float().numpy(), rtol=2e-4, atol=1e-6, err_msg=k)


def test_enet_rejects_bad_inputs():
    from dct_amd.arch import get_arch
    net = get_arch("enet", {"num_classes": 4}).to(DEV)
    with pytest.raises(RuntimeError):
        net(torch.rand(1, 1, 60, 64, device=DEV))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 1, 64, 64))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_enet_grouped_passes_are_bit_identical(dtype):
    """Three independent passes (two of one network, one of another) recorded into a K.PassGroup and issued as ONE chain of
    grouped launches give bit for bit the logits, the statistics and the per-pass gradient buffers of the same passes
    launched one after the other (include/dct.h "grouped passes")."""
    from dct_amd import hip_ops as K
    C = 2
    nets = [_hip_net(_oracle_net(C, 31 + i), C, dtype).train() for i in range(2)]
    for n in nets:
        n.flat_params.ensure()
    g = torch.Generator().manual_seed(9)
    passes = [(nets[0], torch.rand(3, 1, 72, 88, generator=g).to(DEV)), (nets[0], torch.rand(3, 1, 72, 88, generator=g).to(DEV)),
              (nets[1], torch.rand(3, 1, 72, 88, generator=g).to(DEV))]
    dls = [torch.randn(3, 72, 88, C, generator=g).to(DEV) for _ in passes]
    dls_c = [d if dtype == torch.float32 else d.to(dtype) for d in dls]

    def run(grouped):
        outs, tapes, bufs = [], [], []
        if grouped:
            with K.PassGroup(len(passes)) as grp:
                for m, (net, x) in enumerate(passes):
                    grp.member(m)
                    lp, tape = net.plan_forward(x, True, defer_running=True)
                    outs.append(lp); tapes.append(tape)
            assert grp.grouped > 100 and grp.single == 0, (grp.grouped, grp.single)
        else:
            for net, x in passes:
                lp, tape = net.plan_forward(x, True, defer_running=True)
                outs.append(lp); tapes.append(tape)
        stats = [t[-1]["bn_stats"].clone() for t in tapes]
        for net, _ in passes:
            bufs.append(torch.zeros(net.flat_params.total, dtype=torch.float32, device=DEV))
        if grouped:
            with K.PassGroup(len(passes)) as grp:
                for m, (net, x) in enumerate(passes):
                    grp.member(m)
                    net.plan_backward(tapes[m], dls_c[m], need_dx=False, need_dw=True, grad_buffer=bufs[m])
            assert grp.grouped > 250 and grp.single == 0, (grp.grouped, grp.single)
        else:
            for m, (net, x) in enumerate(passes):
                net.plan_backward(tapes[m], dls_c[m], need_dx=False, need_dw=True, grad_buffer=bufs[m])
        torch.cuda.synchronize()
        return [o.clone() for o in outs], stats, bufs

    a = run(False)
    b = run(True)
    for m in range(len(passes)):
        assert torch.equal(a[0][m], b[0][m]), f"logits of pass {m}"
        net = passes[m][0]
        for bn in net._bn_list:                 # (the flat buffer pads every vector to 4 floats: compare the vectors)
            c, base = bn.num_features, 5 * net._bn_off[id(bn)]
            cp = (c + 3) // 4 * 4
            va, vb = a[1][m][base:base + 5 * cp].view(5, cp)[:, :c], b[1][m][base:base + 5 * cp].view(5, cp)[:, :c]
            assert torch.equal(va, vb), f"batch statistics of pass {m}"
        assert torch.equal(a[2][m], b[2][m]), f"gradient buffer of pass {m}"
        assert a[2][m].abs().max().item() > 0
    # anything that launches at once is refused while a group is open
    with pytest.raises(RuntimeError):
        with K.PassGroup(2):
            K.cast(dls[0], torch.empty_like(dls[0], dtype=torch.bfloat16))


def test_enet_backward_leaves_on_a_side_stream_are_bit_identical():
    """K.LeafSide: the weight / bias gradients of a backward pass held back and issued on another stream in four batches give the
    gradient buffer of the plain pass bit for bit (and the data-gradient chain itself launches at once)."""
    from dct_amd import hip_ops as K
    C = 2
    net = _hip_net(_oracle_net(C, 41), C, torch.bfloat16).train()
    net.flat_params.ensure()
    g = torch.Generator().manual_seed(10)
    x = torch.rand(4, 1, 80, 96, generator=g).to(DEV)
    dl = torch.randn(4, 80, 96, C, generator=g).to(DEV).to(torch.bfloat16)
    lp, tape = net.plan_forward(x, True, defer_running=True)
    want = torch.zeros(net.flat_params.total, dtype=torch.float32, device=DEV)
    net.plan_backward(tape, dl, need_dx=False, need_dw=True, grad_buffer=want)
    got = torch.zeros_like(want)
    main, side_stream = torch.cuda.current_stream(), torch.cuda.Stream()
    flushes = []
    with K.LeafSide() as side:
        def out():
            ev = torch.cuda.Event()
            ev.record(main)
            side_stream.wait_event(ev)
            with torch.cuda.stream(side_stream):
                before = side.launches
                side.flush()
                flushes.append(side.launches - before)
        net.plan_backward(tape, dl, need_dx=False, need_dw=True, grad_buffer=got, leaf_hook=out)
        out()
    keep = side.kept
    main.wait_stream(side_stream)
    torch.cuda.synchronize()
    assert len(flushes) >= 3 and all(n > 0 for n in flushes) and side.launches > 120, (flushes, side.launches)
    assert torch.equal(got, want)
    del keep


@pytest.mark.parametrize("need_dw", [True, False])
def test_enet_denormalise_on_load_is_bit_identical(need_dw):
    """Data-gradient convolutions computing the BatchNorm-backward result of their input on load (K.enet_conv_bwd_in) give bit for
    bit the gradients of the plan that materialises it with the apply kernel first (one shared device function for the arithmetic)."""
    C = 2
    outs = []
    for on_load in (False, True):
        net = _hip_net(_oracle_net(C, 61), C, torch.bfloat16).train()
        net.denorm_on_load, net.denorm_on_load_all = on_load, True
        net.flat_params.ensure()
        g = torch.Generator().manual_seed(13)
        x = torch.rand(4, 1, 96, 88, generator=g).to(DEV)
        dl = torch.randn(4, 96, 88, C, generator=g).to(DEV).to(torch.bfloat16)
        lp, tape = net.plan_forward(x, True, defer_running=True)
        buf = torch.zeros(net.flat_params.total, dtype=torch.float32, device=DEV)
        dx = net.plan_backward(tape, dl, need_dx=True, need_dw=need_dw, grad_buffer=buf)
        torch.cuda.synchronize()
        outs.append((lp.clone(), buf, dx.clone()))
    for a, b, what in zip(outs[0], outs[1], ("logits", "gradients", "dx")):
        assert torch.equal(a, b), what
    assert outs[0][2].abs().max().item() > 0 and (not need_dw or outs[0][1].abs().max().item() > 0)

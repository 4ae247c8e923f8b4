"""fp16 compute mode (DCT_F16, BASELINE configs[4]: "3x Enet, Prostate 320x320 fp16"): Enet activations and activation
gradients in IEEE half, fp32 raw conv outputs / statistics / parameters, power-of-two loss scaling inside the fused step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from helpers import FakeLoader, batches, blob_batches, round_conv_operands  # noqa: E402

DEV = "cuda:0"


def _rel2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def _round_points(onet, dtype):
    """Round the oracle's block outputs exactly where the low-precision plan stores them (tests/test_enet_gpu.py)."""
    def rnd(_m, _i, o):
        if isinstance(o, tuple):
            return (o[0].to(dtype).float(),) + tuple(o[1:])
        return o.to(dtype).float()
    for m in onet.modules():
        if m.__class__.__name__ in ("_Bottleneck", "_Initial"):
            m.register_forward_hook(rnd)
    return round_conv_operands(onet, dtype)        # ... and the MFMA convolutions' operands


def test_enet_f16_forward_backward_vs_oracle():
    """Half has 11 significand bits against bf16's 8: the same comparison as the bf16 plan's (block outputs rounded at the
    same points in the oracle: block outputs and the operands of the MFMA convolutions), at a 4x tighter bound (bf16: 0.2;
    0.038 measured -- each of the ~70 operand roundings adds ties that fp32 noise flips one way here, the other way there)."""
    from dct_amd.arch import get_arch
    C, B, H = 4, 2, 64
    torch.manual_seed(7)
    onet = oracle.build_net("enet", C).train()
    net = get_arch("enet", {"num_classes": C, "compute_dtype": torch.float16})
    net.load_state_dict(onet.state_dict())
    net = net.to(DEV).train()
    _round_points(onet, torch.float16)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 1, H, H, generator=g)
    t = torch.randint(0, C, (B, H, H), generator=g)
    xo = x.clone().requires_grad_(True)
    yo = onet(xo)
    oracle.cross_entropy_2d(yo, t).backward()
    xd = x.to(DEV).requires_grad_(True)
    y = net(xd)
    assert y.dtype == torch.float32
    assert _rel2(y.detach().cpu().numpy(), yo.detach().numpy()) < 0.05
    yo2 = yo.detach().clone().requires_grad_(True)
    gl = torch.autograd.grad(oracle.cross_entropy_2d(yo2, t), yo2)[0]
    y.backward((gl * 1024.0).to(DEV))          # the autograd entry point does not scale: 1 / (B H W) = 1.2e-4 is near half's subnormals
    last = {k: p.grad.cpu().numpy() / 1024.0 for k, p in net.named_parameters() if k.startswith("decoder.layers.5")}
    ref = {k: p.grad.numpy() for k, p in onet.named_parameters() if k.startswith("decoder.layers.5")}
    assert len(last) == 2
    for k in last:
        assert _rel2(last[k], ref[k]) < 0.05, k
    for p in net.parameters():
        assert torch.isfinite(p.grad).all()


def _trainer(tmp_path, dtype, n, arch="enet", C=3, B=2, H=64):
    from dct_amd.loss import get_loss_fn
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    segs = []
    for seed in (11, 12, 13):
        torch.manual_seed(seed)
        sd = oracle.build_net(arch, C).state_dict()
        seg = Segmentator({"name": arch, "num_classes": C, "compute_dtype": dtype}, {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4},
                          {"name": "StepLR", "step_size": 90, "gamma": 0.1})
        seg.torchnet.load_state_dict(sd)
        segs.append(seg)
    lab = [FakeLoader(blob_batches(81 + i, n, B, H, C), B) for i in range(3)]
    unl = FakeLoader(blob_batches(91, n, 2 * B, H, C), 2 * B)
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=str(tmp_path), device=DEV, axises=list(range(1, C)),
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=n)
    for s in segs:
        s.train()
    return tr, lab, unl


def _run(tr, lab, unl, n, pair=(0, 2)):
    log = []
    for k in range(n):
        lb = [(lab[i][k][0][0], lab[i][k][0][1]) for i in range(3)]
        out = tr._run_step(lb, (unl[k][0][0], unl[k][0][1]), True, True, pair)
        log.append([float(v) for v in out["sup"]] + [float(out["jsd"]), float(out["adv"])])
    torch.cuda.synchronize()
    w = [torch.cat([p.detach().flatten() for p in s.torchnet.parameters()]).cpu() for s in tr.segmentators]
    return np.array(log), w


def test_three_view_f16_step_tracks_fp32(tmp_path):
    """cfg5's shape of step (3 x Enet, lab : unlab 1 : 2 here, JSD over three views + FGSM on a pair) in fp16 against the
    fp32 kernels from the same weights: losses of five steps (eager, capture, replays) and the weights they lead to."""
    n = 5
    res = {}
    for dtype in (torch.float32, torch.float16):
        tr, lab, unl = _trainer(tmp_path, dtype, n)
        res[dtype] = _run(tr, lab, unl, n)
        if dtype == torch.float16:
            assert tr._loss_scale == 2.0 ** 13 and tr._grad_unscale == 2.0 ** -13     # 2 x 64 x 64 labeled pixels
            assert tr._step_graphs is not None and tr._step_graphs.replays >= 2
    (la, wa), (lb, wb) = res[torch.float32], res[torch.float16]
    assert np.isfinite(lb).all()
    np.testing.assert_allclose(lb[0, :3], la[0, :3], rtol=5e-3)           # supervised losses, identical weights
    np.testing.assert_allclose(lb[:, :3], la[:, :3], rtol=5e-2)
    np.testing.assert_allclose(lb[:, 3], la[:, 3], rtol=0.2, atol=1e-6)   # JSD
    for a, b in zip(wa, wb):
        assert ((a - b).norm() / a.norm()).item() < 2e-2


def test_power_of_two_loss_scale_is_exact(tmp_path):
    """The scale applied to every loss gradient and divided out inside the fused Adam must not change a bit where no value
    under- or overflows: bf16 storage (fp32's exponent range) with and without a forced 2^12 scale."""
    n = 3
    res = []
    for scale in (None, 4096.0):
        tr, lab, unl = _trainer(tmp_path, torch.bfloat16, n)
        tr.force_loss_scale = scale
        res.append(_run(tr, lab, unl, n))
    (la, wa), (lb, wb) = res
    assert (la == lb).all()
    for a, b in zip(wa, wb):
        assert torch.equal(a, b)


def test_unet_rejects_f16():
    from dct_amd.arch import get_arch
    with pytest.raises(ValueError, match="fp16 is an Enet mode"):
        get_arch("unet", {"num_classes": 4, "compute_dtype": torch.float16})

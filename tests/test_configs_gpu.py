"""BASELINE.json's configurations as they are benchmarked, held to the CPU oracle at full size ("next" 2):

* cfg1  S = 1, cross-entropy only (no JSD, no FGSM) through both ``_run_step`` paths.  The reference ``unet`` cannot run at
        64 x 64 (valid convolutions: H >= 176), so cfg1's "4x1x64x64, 2-class" plumbing case runs ``enet``
        at 64 x 64 and ``unet`` at its 176 x 176 minimum -- never a padded UNet.
* cfg2 / cfg3  exactly what ``bench.py`` times: ``bench.make_trainer`` (2 x UNet, 8 + 8 images of 256 x 256, C = 4, dropout
        on, one HIP stream per model, labeled + unlabeled batch in one pass, HIP-graph replay after three eager steps), five
        steps against ``oracle.cotrain_step`` on the same weights, batches and (replayed) dropout masks: losses of every
        step, logits of step 0, per-tensor weight displacement after the last step; bf16 (the benchmarked mode) and fp32.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from helpers import FakeLoader, MaskReplayNet, batches, blob_batches  # noqa: E402

DEV = "cuda:0"
REPORT = bool(os.environ.get("DCT_PARITY_REPORT"))


# one-step update of the re-synced bf16 step against the oracle, worst tensor of >= 4096 elements.  Measured (cfg2 / cfg3, every step):
# displacement cosine 0.9922 at step 0 (no history: lr * sign(g)), >= 0.9958 from step 1 on; first moment within 4.1e-2 (step 0: it IS
# 0.1 x the bf16 gradient of that tensor) / 3.3e-2 in relative L2
COS_STEP0, COS_STEP, EXP_AVG_REL = 0.98, 0.99, 0.1
ADV_RTOL = 1e-2     # adversarial KL of the re-synced bf16 step against the oracle: FGSM sign flips; measured <= 1.8e-3 at every step


def _say(*a):
    if REPORT:
        print(*a, flush=True)


def _crit():
    from dct_amd.loss import get_loss_fn
    return {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}


def _cotrainer(tmp_path, segs, lab, unl, C, n):
    from dct_amd.trainer import CoTrainer
    return CoTrainer(segs, lab, unl, unl, _crit(), max_epoch=1, save_dir=str(tmp_path), device=DEV, axises=list(range(1, C)),
                     cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                     adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                     adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=n)


# ------------------------------------------------------------------------------------------------ cfg1
@pytest.mark.parametrize("arch,H", [("enet", 64), ("unet", 176)])
@pytest.mark.parametrize("fused", [True, False])
def test_cfg1_single_model_supervised_only(tmp_path, arch, H, fused):
    """BASELINE configs[0]: one model, fully supervised CE, batch 4, 2 classes.  Five steps (two eager, the capture, two
    replays on the fused path) against the oracle's step with S = 1, train_jsd = train_adv = False."""
    from dct_amd import ModelMode
    from dct_amd.models import Segmentator
    C, B, n = 2, 4, 5
    torch.manual_seed(3)
    onet = oracle.build_net(arch, C, **({"dropout_p": 0.0} if arch == "unet" else {})).train()
    arch_dict = {"name": arch, "num_classes": C, "compute_dtype": torch.float32}
    if arch == "unet":
        arch_dict["dropout_p"] = 0.0
    seg = Segmentator(arch_dict, {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4}, {"name": "StepLR", "step_size": 90, "gamma": 0.1})
    seg.torchnet.load_state_dict(onet.state_dict())
    om = oracle.OracleModel.make(onet)
    lab = [FakeLoader(blob_batches(17, n, B, H, C), B)]
    unl = FakeLoader(batches(18, n, B, H, C), B)
    tr = _cotrainer(tmp_path, [seg], lab, unl, C, n)
    assert tr._fused_ok()
    if not fused:
        tr._fused_ok = lambda: False
    seg.train()
    for k in range(n):
        img, gt = lab[0][k][0]
        out = tr._run_step([(img, gt)], None, False, False)
        ref = oracle.cotrain_step([om], [(img, gt)], None, False, False)
        assert out["jsd"] == 0 and out["adv"] == 0 and len(out["sup"]) == 1 and out["unlab_probs"] == []
        tol = 2e-5 if k == 0 else (1e-2 if arch == "enet" else 2e-3)     # later steps inherit Adam's sign-like first update
        np.testing.assert_allclose(out["sup"][0].item(), ref["sup"][0].item(), rtol=tol)
        if k == 0:
            np.testing.assert_allclose(out["preds"][0].cpu().numpy(), ref["preds"][0].numpy(), rtol=0, atol=2e-5 * ref["preds"][0].abs().max().item())
    if fused:
        assert tr._step_graphs is not None and tr._step_graphs.replays >= 2
    a = torch.cat([p.detach().flatten().cpu() for p in seg.torchnet.parameters()]).double()
    b = torch.cat([p.detach().flatten() for p in onet.parameters()]).double()
    # five Adam steps of lr 1e-3: an element whose tiny gradient flips sign between the two arithmetics moves by up to 2 n lr
    assert ((a - b).norm() / b.norm()).item() < (2e-2 if arch == "enet" else 1e-2)
    # the epoch loop with the same switches (what train_ACDC_cotraining.py runs for a fully-supervised baseline)
    d_lab, d_unl = tr._train_loop(lab, unl, epoch=0, mode=ModelMode.TRAIN, save=False, train_jsd=False, train_adv=False)
    assert d_lab.shape == (1, C, 2) and torch.isfinite(d_lab).all()


# ------------------------------------------------------------------------------------------------ cfg2 / cfg3 as benchmarked
def _oracle_models(tr, dropout_p):
    oms = []
    for seg in tr.segmentators:
        onet = oracle.build_net("unet", tr.C, dropout_p=dropout_p).train()
        onet.load_state_dict({k: v.detach().cpu() for k, v in seg.torchnet.state_dict().items()})
        oms.append(oracle.OracleModel.make(MaskReplayNet(onet)))
    return oms


def _nchw_mask(m):
    return m.permute(0, 3, 1, 2).float().cpu()


@pytest.mark.parametrize("config,dtype", [("cfg2", "bf16"), ("cfg3", "bf16"), ("cfg2", "f32")])
def test_bench_path_full_size_vs_oracle(config, dtype):
    import bench
    cfg = bench.CONFIGS[config]
    tdtype = torch.bfloat16 if dtype == "bf16" else torch.float32
    n = 5
    tr, lab, unl = bench.make_trainer(cfg, tdtype, torch.device(DEV), 0, 1, None, n_batches=n)
    assert tr._fused_ok() and tr.use_hip_graph and tr.model_streams and tr.batch_lab_unlab
    S, B_l, adv = cfg["S"], cfg["B_l"], cfg["train_adv"]
    nets = [s.torchnet for s in tr.segmentators]
    assert all(net.dropout_p == 0.5 and net.training for net in nets)
    for net in nets:
        net.record_dropout_masks = True
    oms = _oracle_models(tr, 0.5)
    w0 = [{k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for net in nets]
    bf = dtype == "bf16"
    for k in range(n):
        lb = [(lab[m][k][0][0], lab[m][k][0][1]) for m in range(S)]
        ub = (unl[k][0][0], unl[k][0][1])
        graphs = tr._step_graphs
        replay = graphs is not None and graphs.captures > 0
        if not replay:
            for net in nets:
                net.dropout_mask_log.clear()
        out = tr._run_step(lb, ub, True, adv, (0, 1) if adv else None)
        torch.cuda.synchronize()
        # the masks this step drew (static buffers of the captured graph once it replays) -> the oracle's forwards, in its order:
        # labeled pass, unlabeled pass [, third pass: FGSM forward of model b / adversarial forward of model a]
        for m, net in enumerate(nets):
            log = net.dropout_mask_log
            assert len(log) == (2 if adv else 1)
            joint = [_nchw_mask(t) for t in log[0]]
            q = [[t[:B_l] for t in joint], [t[B_l:] for t in joint]]
            if adv:
                q.append([_nchw_mask(t) for t in log[1]])
            oms[m].net.queue = q
        ref = oracle.cotrain_step(oms, [(a.cpu(), b.cpu()) for a, b in lb], ub[0].cpu(), True, adv, lam_cot=0.5, lam_adv=0.05, eps=0.03)
        assert all(len(om.net.queue) == 0 for om in oms)
        sup, rsup = [float(v) for v in out["sup"]], [float(v) for v in ref["sup"]]
        jsd, rjsd = float(out["jsd"]), float(ref["jsd"])
        _say(config, dtype, "step", k, "replay" if replay else "eager", "sup", sup, rsup, "jsd", jsd, rjsd,
             "adv", float(out["adv"]) if adv else None, float(ref["adv"]) if adv else None)
        # bf16 activations carry ~2^-8 relative rounding per layer; from step 1 on both sides also carry Adam's sign-like first
        # updates of near-zero gradient elements.  fp32: kernels vs ATen on the same weights.
        tol_sup = (1e-2 if k == 0 else 6e-2) if bf else (2e-5 if k == 0 else 2e-3)     # (bf16, step 3 -- a loss spike after three sign-like Adam steps: 1.8-3.2 % measured)
        np.testing.assert_allclose(sup, rsup, rtol=tol_sup)
        # JSD: a small difference of two near-equal predictions -- bf16 rounding of the logits moves it by tens of percent
        # once the nets have taken a few (sign-like) Adam steps (measured 24 % at step 4); fp32 stays within 2e-4
        # From step 1 on the bf16 bands below are a sanity check only (same order of magnitude): the two trajectories diverge chaotically
        # and every process draws its own dropout masks, so a 35 % band held in some runs and failed in others (cfg3, round 3: JSD
        # 0.00094 vs 0.00198 at step 3).  The bound that means something is test_bf16_per_step_resync_vs_oracle below: oracle weights
        # loaded before every step, JSD within 2e-2 and the adversarial KL within its FGSM band at EVERY step.
        def band(a, b, rtol):
            if bf and k > 0:
                assert b / 3.0 - 1e-6 <= a <= 3.0 * b + 1e-6, (a, b)
            else:
                np.testing.assert_allclose(a, b, rtol=rtol, atol=1e-6)
        band(jsd, rjsd, (0.05 if bf else (1e-4 if k == 0 else 2e-2)))
        if adv:
            band(float(out["adv"]), float(ref["adv"]), (0.35 if bf else 5e-2))
        if k == 0:
            for m in range(S):
                a, b = out["preds"][m].float().cpu(), ref["preds"][m]
                err = ((a - b).abs().max() / b.abs().max()).item()
                _say("  logits model", m, "max-rel err", err)
                assert err < (4e-2 if bf else 1e-5)
    # (the first step allocates gradient buffers and Adam moments, which changes the step signature: two eager steps of the
    # steady signature follow, then the capture -- bench.py's set-up phase covers the same steps)
    assert tr._step_graphs is not None and tr._step_graphs.captures == 1 and tr._step_graphs.replays == n - 3
    # weights after n steps: displacement from the initial weights, per tensor, HIP vs oracle
    for m, net in enumerate(nets):
        sd = net.state_dict()
        osd = oms[m].net.net.state_dict()
        worst_cos, worst_rel = 1.0, 0.0
        for name, w_init in w0[m].items():
            d_hip = (sd[name].detach().cpu() - w_init).double().flatten()
            d_ref = (osd[name] - w_init).double().flatten()
            cos = float(d_hip @ d_ref / (d_hip.norm() * d_ref.norm() + 1e-30))
            rel = float((sd[name].detach().cpu().double().flatten() - osd[name].double().flatten()).norm() / osd[name].double().norm())
            _say("  model", m, name, "numel", d_ref.numel(), "cos(delta)", round(cos, 4), "rel(w)", rel)
            if d_ref.numel() >= 4096:
                worst_cos = min(worst_cos, cos)
            worst_rel = max(worst_rel, rel)
            # every element moved at most n * lr (Adam) on both sides
            assert float(d_hip.abs().max()) <= 1.05 * n * 1e-3 + 1e-7
        _say(" model", m, "worst cos", worst_cos, "worst rel", worst_rel)
        # fp32 (measured): cos >= 0.9995 on every tensor; relative weight error <= 6e-3 on the weight tensors and 2.2e-2 on the
        # centre biases, whose gradients (behind dropout and mostly-dead ReLUs) are tiny: Adam turns their last bits into
        # +-lr steps (chaotic parity points)
        # bf16 (measured): cos >= 0.989 on every tensor of >= 4096 elements, relative weight error <= 0.17 (small bias tensors)
        assert worst_cos > (0.95 if bf else 0.995)
        assert worst_rel < (0.3 if bf else 5e-2)


# ------------------------------------------------------------------------------------------------ ADVICE r1
def test_fused_step_with_a_stock_torch_optimizer(tmp_path):
    """optim_dict names any torch.optim class (models/segmentators.py:37-43).  With SGD the fused step must stay eager (no
    graph: torch's optimizers keep lr / step on the host) and match the oracle's SGD step."""
    from dct_amd.models import Segmentator
    C, B, H, n = 3, 2, 176, 4
    segs, oms = [], []
    for seed in (41, 42):
        torch.manual_seed(seed)
        onet = oracle.build_net("unet", C, dropout_p=0.0).train()
        seg = Segmentator({"name": "unet", "num_classes": C, "compute_dtype": torch.float32, "dropout_p": 0.0},
                          {"name": "SGD", "lr": 1e-2, "momentum": 0.9}, {"name": "StepLR", "step_size": 90, "gamma": 0.1})
        assert type(seg.optimizer) is torch.optim.SGD
        seg.torchnet.load_state_dict(onet.state_dict())
        segs.append(seg)
        oms.append(oracle.OracleModel(onet, torch.optim.SGD(onet.parameters(), lr=1e-2, momentum=0.9)))
    lab = [FakeLoader(batches(51 + i, n, B, H, C), B) for i in range(2)]
    unl = FakeLoader(batches(61, n, B, H, C), B)
    tr = _cotrainer(tmp_path, segs, lab, unl, C, n)
    for s in segs:
        s.train()
    for k in range(n):
        lb = [(lab[i][k][0][0], lab[i][k][0][1]) for i in range(2)]
        out = tr._run_step(lb, (unl[k][0][0], unl[k][0][1]), True, False)
        ref = oracle.cotrain_step(oms, lb, unl[k][0][0], True, False, lam_cot=0.5)
        np.testing.assert_allclose([float(v) for v in out["sup"]], [float(v) for v in ref["sup"]], rtol=1e-4)
    assert tr._step_graphs is None or tr._step_graphs.captures == 0
    for seg, om in zip(segs, oms):
        a = torch.cat([p.detach().flatten().cpu() for p in seg.torchnet.parameters()]).double()
        b = torch.cat([p.detach().flatten() for p in om.net.parameters()]).double()
        assert ((a - b).norm() / b.norm()).item() < 1e-4


def test_cotrained_models_draw_different_dropout_masks():
    from dct_amd.arch import get_arch
    torch.manual_seed(5)
    nets = [get_arch("unet", {"num_classes": 4}).to(DEV).train() for _ in range(2)]
    x = torch.rand(2, 1, 176, 176, device=DEV)
    masks = []
    for net in nets:
        net.record_dropout_masks = True
        net.plan_forward(x, False)
        masks.append([m.clone() for m in net.last_dropout_masks])
    for a, b in zip(*masks):
        agree = (a == b).float().mean().item()
        assert 0.45 < agree < 0.55          # independent Bernoulli(0.5) masks agree on half the elements


@pytest.mark.parametrize("method,axes", [("2d", [1, 2, 3]), ("3d", "all")])
def test_dice_meter_running_moments_equal_the_history_formulas(method, axes):
    """DiceMeter.value() from running sums (O(1) per call) == the reference's formulas over the concatenated history
    (metrics/dice_meter.py:61-74): per-class mean / unbiased std, and mean / std of the per-row report mean."""
    from dct_amd.metrics import DiceMeter
    g = torch.Generator().manual_seed(3)
    m = DiceMeter(method=method, report_axises=axes, C=4)
    rows = []
    for k in range(7):
        pred = torch.randn(3, 4, 32, 32, generator=g)
        gt = torch.randint(0, 4, (3, 1, 32, 32), generator=g)
        m.add(pred.to(DEV), gt.to(DEV))
        rows.append(oracle.dice_2d(pred, gt) if method == "2d" else oracle.dice_3d(pred, gt).unsqueeze(0))
    log = torch.cat(rows)
    rep = log.mean(1) if axes == "all" else log[:, axes].mean(1)
    (rm, rs), (cm, cs) = m.value()
    np.testing.assert_allclose(cm.cpu().numpy(), log.mean(0).numpy(), rtol=1e-5)
    np.testing.assert_allclose(cs.cpu().numpy(), log.std(0).numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(rm), float(rep.mean()), rtol=1e-5)
    np.testing.assert_allclose(float(rs), float(rep.std()), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.log.cpu().numpy(), log.numpy(), rtol=1e-5)


@pytest.mark.parametrize("config,dtype", [("cfg4", "bf16"), ("cfg5", "bf16"), ("cfg5", "f16")])
def test_enet_configs_as_benchmarked(config, dtype):
    """BASELINE configs[3] / [4] exactly as bench.py builds them (cfg4: 2 x Enet, 200 x 200, 8 + 8, JSD + FGSM; cfg5: 3 x Enet,
    320 x 320, lab : unlab 4 : 16, FGSM on the pair (0, 1); bf16 and cfg5's fp16): five steps (eager, capture, replay, one
    backward stream per pass) against the SAME trainer in fp32 compute mode -- the parity mode that the small-size tests hold
    to the reference goldens and the oracle.  Step 0 starts from identical weights; later steps carry Adam's sign-like updates."""
    import bench
    cfg = bench.CONFIGS[config]
    n, S = 5, cfg["S"]
    logs = {}
    for dt in ("f32", dtype):
        tdtype = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[dt]
        tr, lab, unl = bench.make_trainer(cfg, tdtype, torch.device(DEV), 0, 1, None, n_batches=n)
        rows = []
        for k in range(n if dt != "f32" else 2):
            lb = [(lab[m][k][0][0], lab[m][k][0][1]) for m in range(S)]
            out = tr._run_step(lb, (unl[k][0][0], unl[k][0][1]), True, True, (0, 1))
            rows.append([float(v) for v in out["sup"]] + [float(out["jsd"]), float(out["adv"])])
        torch.cuda.synchronize()
        logs[dt] = np.array(rows)
        if dt != "f32":
            assert tr._step_graphs is not None and tr._step_graphs.captures == 1 and tr._step_graphs.replays >= 1
            assert tr._loss_scale == (1.0 if dt == "bf16" else 2.0 ** (cfg["B_l"] * cfg["H"] ** 2 - 1).bit_length())
    a, b = logs["f32"], logs[dtype]
    _say(config, dtype, "fp32", a.tolist(), dtype, b.tolist())
    assert np.isfinite(b).all()
    np.testing.assert_allclose(b[0, :S], a[0, :S], rtol=2e-2 if dtype == "bf16" else 5e-3)     # supervised losses, same weights
    np.testing.assert_allclose(b[1, :S], a[1, :S], rtol=6e-2)
    np.testing.assert_allclose(b[0, S], a[0, S], rtol=0.3, atol=1e-6)                            # JSD of near-equal predictions
    assert b[-1, :S].mean() < b[0, :S].mean()                                                    # and it trains


# ------------------------------------------------------------------------------------------------ cfg4 / cfg5 vs the ORACLE at full size
def _sync_weights_from_oracle(tr, oms, moments=False):
    """HIP nets <- the oracle's current weights and BatchNorm buffers, IN PLACE (addresses, and with them a captured step,
    stay valid).  With the same weights in front of every step, step k's logits / losses / gradients are as comparable as
    step 0's -- no trajectory divergence -- whichever way the step is executed (eager, capture, replay)."""
    from dct_amd import hip_ops as K
    for seg, om in zip(tr.segmentators, oms):
        net = seg.torchnet
        inner = om.net.net if hasattr(om.net, "queue") else om.net
        net.load_state_dict({k: v.detach().to(torch.float32) for k, v in inner.state_dict().items()})
        fp = net.flat_params
        if getattr(fp, "shadow", None) is not None:       # UNet bf16: the replayed step reads the shadow the fused Adam writes
            K.pack_weight(fp.flat, fp.shadow, 1, 1, fp.total)
        if moments:                                       # ... and Adam's moments (views into the fused optimizer's flat buffers)
            for p, po in zip(net.parameters(), inner.parameters()):
                st, sto = seg.optimizer.state.get(p), om.optimizer.state.get(po)
                if st and sto and "exp_avg" in st and "exp_avg" in sto:
                    st["exp_avg"].copy_(sto["exp_avg"])
                    st["exp_avg_sq"].copy_(sto["exp_avg_sq"])
    torch.cuda.synchronize()


def _grad_cosines(tr, oms, scale):
    """per-tensor cosine between the HIP step's accumulated gradient (flat buffer, still in place after the optimizer ran) and
    the oracle's p.grad; tensors whose oracle gradient is rounding noise (conv biases in front of a BatchNorm) are skipped.
    -> (worst over tensors of >= 1024 elements, worst over the smaller ones, name of the overall worst)"""
    big, small, worst, worst_name = 1.0, 1.0, 2.0, None
    for m, (seg, om) in enumerate(zip(tr.segmentators, oms)):
        inner = om.net.net if hasattr(om.net, "queue") else om.net
        for (name, p), (_, po) in zip(seg.torchnet.named_parameters(), inner.named_parameters()):
            go = po.grad.detach().double().flatten()
            if go.norm() < 1e-6 or go.numel() < 16:
                continue
            gh = (p.grad.detach().double().cpu().flatten()) / scale
            cos = float(gh @ go / (gh.norm() * go.norm() + 1e-300))
            if go.numel() >= 1024:
                big = min(big, cos)
            else:
                small = min(small, cos)
            if cos < worst:
                worst, worst_name = cos, f"model{m}.{name}"
    return big, small, worst_name


@pytest.mark.parametrize("config,dtype", [("cfg4", "f32"), ("cfg5", "f32"), ("cfg4", "bf16"), ("cfg5", "bf16"), ("cfg5", "f16")])
def test_enet_configs_full_size_vs_oracle(config, dtype):
    """VERDICT r2 (weak 1): cfg4 (2 x Enet 200 x 200, 8 + 8, CE + JSD + FGSM) and cfg5 (3 x Enet 320 x 320, 4 + 16) exactly as
    ``bench.py`` builds them -- MFMA forms, four-queue layout, deferred running statistics, the program-of-graphs replay --
    against ``oracle.cotrain_step`` on the same weights and batches, five steps with the oracle's weights loaded in front of
    every step (so the captured / replayed steps are held as tightly as step 0).

    fp32 mode vs the oracle in FLOAT64 (deterministic whatever the host's thread count).  Bounds, with what was measured on
    the MI355X (profiles/r03_full_size_parity_and_dsc.txt):
      supervised losses <= 1e-5 (2.4e-6), JSD <= 1e-4 (1.6e-6);
      adversarial KL <= 5e-3 (1.8e-3): FGSM takes sign(d loss / d x), which flips wherever the two arithmetics put a
        near-zero gradient on different sides of 0, and every flipped pixel moves x_adv by 2 eps;
      logits: at 16 x 200 x 200 there are ~1e7 max-pool windows and PReLU / ReLU thresholds, so a handful of argmax ties within
        one fp32 ulp flip against float64; each moves ONE activation to a neighbouring pixel and shows as an O(1e-2) difference
        over that pixel's receptive field.  So: 99 % of the logits within 2e-5 of the oracle (relative to max |logit|), and
        L2-relative error of the whole tensor <= 5e-3 (1.4e-3 on steps with flips, 5e-6 on steps without);
      per-tensor gradient cosine >= 0.999 on tensors of >= 1024 elements, >= 0.99 on the small ones (BatchNorm scales / PReLU
        slopes of 16-128 elements: 0.9987 measured).
    bf16 / fp16 mode vs the fp32 oracle rounding the MFMA convolutions' operands where the kernels do
    (helpers.round_conv_operands): supervised losses <= 2e-3 (4.4e-4), JSD <= 5e-3 (8e-4), adversarial KL <= 2e-2 (5.5e-3).  The
    16-bit Enet is the chaotic amplifier of tests/test_enet_gpu.py -- its logits sit 5-20 % (L2) from the oracle's and its
    early-layer gradients decorrelate -- which is why the parity claim is the fp32 mode."""
    import bench
    from helpers import round_conv_operands
    cfg = bench.CONFIGS[config]
    tdtype = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[dtype]
    n, S, C = 5, cfg["S"], cfg["C"]
    tr, lab, unl = bench.make_trainer(cfg, tdtype, torch.device(DEV), 0, 1, None, n_batches=n)
    exact = dtype == "f32"
    oms = []
    for seg in tr.segmentators:
        onet = oracle.build_net("enet", C).train()
        onet.load_state_dict({k: v.detach().cpu() for k, v in seg.torchnet.state_dict().items()})
        if exact:
            onet = onet.double()
        else:
            round_conv_operands(onet, tdtype)
        oms.append(oracle.OracleModel.make(onet))
    odt = torch.float64 if exact else torch.float32
    rows = []
    for k in range(n):
        _sync_weights_from_oracle(tr, oms)
        lb = [(lab[m][k][0][0], lab[m][k][0][1]) for m in range(S)]
        ub = (unl[k][0][0], unl[k][0][1])
        replay = tr._step_graphs is not None and tr._step_graphs.captures > 0
        out = tr._run_step(lb, ub, True, True, (0, 1))
        torch.cuda.synchronize()
        ref = oracle.cotrain_step(oms, [(a.cpu().to(odt), b.cpu()) for a, b in lb], ub[0].cpu().to(odt), True, True,
                                  lam_cot=0.5, lam_adv=0.05, eps=0.03, adv_choice=(0, 1))
        sup, rsup = np.array([float(v) for v in out["sup"]]), np.array([float(v) for v in ref["sup"]])
        e_sup = float(np.max(np.abs(sup - rsup) / np.abs(rsup)))
        e_jsd = abs(float(out["jsd"]) - float(ref["jsd"])) / abs(float(ref["jsd"]))
        e_adv = abs(float(out["adv"]) - float(ref["adv"])) / abs(float(ref["adv"]))
        # logits: L2-relative per model.  (Max-norm is reported only: at 16 x 200 x 200 there are ~1e7 max-pool windows and
        # PReLU / ReLU thresholds, so a handful of argmax ties within one fp32 ulp flip against the float64 oracle -- each moves
        # ONE activation to a neighbouring pixel, an O(1e-2) local difference that says nothing about the arithmetic.)
        e_log, e_max, e_q99 = 0.0, 0.0, 0.0
        for m in range(S):
            a, b = out["preds"][m].float().cpu().double(), ref["preds"][m].double()
            err = (a - b).abs().flatten() / b.abs().max()
            e_log = max(e_log, float((a - b).norm() / b.norm()))
            e_max = max(e_max, float(err.max()))
            e_q99 = max(e_q99, float(torch.quantile(err[::7].float(), 0.99)))       # (every 7th element: quantile() caps its input size)
        cos_big, cos_small, cos_name = _grad_cosines(tr, oms, getattr(tr, "_loss_scale", 1.0) or 1.0)
        _say(config, dtype, "step", k, "replay" if replay else "eager", "logits L2", e_log, "q99", e_q99, "max", e_max, "sup", e_sup,
             "jsd", e_jsd, "adv", e_adv, "grad cos big / small tensors", cos_big, cos_small, cos_name)
        rows.append((k, e_log, e_q99, e_sup, e_jsd, e_adv, cos_big, cos_small, cos_name))
    for k, e_log, e_q99, e_sup, e_jsd, e_adv, cos_big, cos_small, cos_name in rows:
        assert np.isfinite([e_log, e_q99, e_sup, e_jsd, e_adv, cos_big, cos_small]).all()
        if exact:
            assert e_q99 <= 2e-5 and e_log <= 5e-3, (k, e_q99, e_log)
            assert e_sup <= 1e-5, (k, e_sup)
            assert e_jsd <= 1e-4 and e_adv <= 5e-3, (k, e_jsd, e_adv)
            assert cos_big >= 0.999 and cos_small >= 0.99, (k, cos_big, cos_small, cos_name)
        else:
            assert e_sup <= 2e-3 and e_jsd <= 5e-3 and e_adv <= 2e-2, (k, e_sup, e_jsd, e_adv)
    assert tr._step_graphs is not None and tr._step_graphs.captures == 1 and tr._step_graphs.replays >= 1


@pytest.mark.parametrize("config", ["cfg2", "cfg3"])
def test_bf16_per_step_resync_vs_oracle(config):
    """VERDICT r2 (weak 3): the benchmarked UNet paths (bf16, HIP-graph replay, model streams / the three-queue adversarial
    layout, dropout on) with the oracle's weights loaded in front of EVERY step, so steps 1-4 (capture and replays included)
    get the bound step 0 has instead of a trajectory-divergence band."""
    import bench
    cfg = bench.CONFIGS[config]
    n, S, B_l, adv = 5, cfg["S"], cfg["B_l"], cfg["train_adv"]
    tr, lab, unl = bench.make_trainer(cfg, torch.bfloat16, torch.device(DEV), 0, 1, None, n_batches=n)
    nets = [s.torchnet for s in tr.segmentators]
    for net in nets:
        net.record_dropout_masks = True
    oms = _oracle_models(tr, 0.5)
    inner = [om.net.net if hasattr(om.net, "queue") else om.net for om in oms]
    for k in range(n):
        _sync_weights_from_oracle(tr, oms, moments=True)
        w_before = [{name: p.detach().clone() for name, p in net_o.named_parameters()} for net_o in inner]
        lb = [(lab[m][k][0][0], lab[m][k][0][1]) for m in range(S)]
        ub = (unl[k][0][0], unl[k][0][1])
        replay = tr._step_graphs is not None and tr._step_graphs.captures > 0
        if not replay:
            for net in nets:
                net.dropout_mask_log.clear()
        out = tr._run_step(lb, ub, True, adv, (0, 1) if adv else None)
        torch.cuda.synchronize()
        for m, net in enumerate(nets):
            log = net.dropout_mask_log
            assert len(log) == (2 if adv else 1)
            joint = [_nchw_mask(t) for t in log[0]]
            q = [[t[:B_l] for t in joint], [t[B_l:] for t in joint]]
            if adv:
                q.append([_nchw_mask(t) for t in log[1]])
            oms[m].net.queue = q
        ref = oracle.cotrain_step(oms, [(a.cpu(), b.cpu()) for a, b in lb], ub[0].cpu(), True, adv, lam_cot=0.5, lam_adv=0.05, eps=0.03)
        sup, rsup = [float(v) for v in out["sup"]], [float(v) for v in ref["sup"]]
        _say(config, "bf16 resync step", k, "replay" if replay else "eager", "sup", sup, rsup, "jsd", float(out["jsd"]), float(ref["jsd"]),
             "adv", float(out["adv"]) if adv else None, float(ref["adv"]) if adv else None)
        np.testing.assert_allclose(sup, rsup, rtol=5e-4)                                   # (measured <= 1.2e-4 at every step: three times + margin)
        np.testing.assert_allclose(float(out["jsd"]), float(ref["jsd"]), rtol=6e-3, atol=1e-6)    # (measured <= 1.5e-3)
        if adv:
            # FGSM takes the SIGN of an input gradient computed in bf16 here and in fp32 there: pixels whose gradient is ~0 flip, and
            # the KL of the two nets on the perturbed batch moves with them (ADV_RTOL: three times the largest deviation measured)
            np.testing.assert_allclose(float(out["adv"]), float(ref["adv"]), rtol=ADV_RTOL, atol=1e-6)
        for m in range(S):
            a, b = out["preds"][m].float().cpu(), ref["preds"][m]
            assert ((a - b).abs().max() / b.abs().max()).item() < 4e-2
        # The optimizer path of the benchmarked dtype at EVERY step (capture and replays included): weights AND Adam moments were
        # equal in front of the step, so this step's update is comparable tensor by tensor -- displacement cosine on the tensors of
        # >= 4096 elements, first moment in relative L2.  (Step 0 has no history: the update is lr * sign(g), and the elements whose
        # gradient is rounding noise flip freely; from step 1 on the moments carry the direction.)
        worst_cos, worst_m, worst_name = 1.0, 0.0, None
        for m, (seg, om) in enumerate(zip(tr.segmentators, oms)):
            for (name, p), (_, po) in zip(seg.torchnet.named_parameters(), inner[m].named_parameters()):
                if po.numel() < 4096:
                    continue
                d_hip = (p.detach().cpu() - w_before[m][name]).double().flatten()
                d_ref = (po.detach() - w_before[m][name]).double().flatten()
                cos = float(d_hip @ d_ref / (d_hip.norm() * d_ref.norm() + 1e-300))
                ma, mo = seg.optimizer.state[p]["exp_avg"].detach().cpu().double().flatten(), om.optimizer.state[po]["exp_avg"].double().flatten()
                rel_m = float((ma - mo).norm() / (mo.norm() + 1e-300))
                if cos < worst_cos:
                    worst_cos, worst_name = cos, f"model {m} {name}"
                worst_m = max(worst_m, rel_m)
                assert float(d_hip.abs().max()) <= 1.05e-3 + 1e-7          # nobody moves further than lr in one Adam step
        _say(config, "bf16 resync step", k, "one-step update: worst displacement cosine", round(worst_cos, 5), worst_name,
             "worst exp_avg rel L2", worst_m)
        assert worst_cos >= (COS_STEP0 if k == 0 else COS_STEP) and worst_m <= EXP_AVG_REL, (k, worst_cos, worst_name, worst_m)
    if not adv:
        assert tr._step_graphs.captures == 1 and tr._step_graphs.replays == n - 3

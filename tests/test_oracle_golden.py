"""Pins the CPU oracle (oracle/) against golden vectors captured from the imported
reference (tools/capture_golden.py).  CPU only; never touches /root/reference."""
import numpy as np
import pytest
import torch

import oracle

T = torch.from_numpy


def _digest(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), t.norm().item(), t.abs().max().item()])


def _seeded_net(arch, C, seed):
    torch.manual_seed(seed)
    kw = {"dropout_p": 0.0} if arch == "unet" else {}
    return oracle.build_net(arch, C, **kw)


# tolerance: same torch build + same CPU kernels => expect agreement to float rounding
TIGHT = dict(rtol=1e-6, atol=1e-7)


def test_g1_losses(golden):
    g = golden("g1_losses")
    torch.manual_seed(int(g["seed"]))
    a, b, c = (torch.randn(2, 4, 8, 8) for _ in range(3))
    t = torch.randint(0, 4, (2, 8, 8))
    t_ign = t.clone()
    t_ign[0, :2] = 255
    la, lb, lc = (x.clone().requires_grad_(True) for x in (a, b, c))
    pa, pb, pc = (oracle.softmax_channels(x) for x in (la, lb, lc))
    ce = oracle.cross_entropy_2d(la, t)
    np.testing.assert_allclose(ce.item(), g["ce"], **TIGHT)
    np.testing.assert_allclose(torch.autograd.grad(ce, la, retain_graph=True)[0].numpy(), g["ce_grad"], **TIGHT)
    cei = oracle.cross_entropy_2d(la, t_ign)
    np.testing.assert_allclose(cei.item(), g["ce_ignore"], **TIGHT)
    np.testing.assert_allclose(torch.autograd.grad(cei, la, retain_graph=True)[0].numpy(), g["ce_ignore_grad"], **TIGHT)
    j2 = oracle.jsd_2d([pa, pb])
    np.testing.assert_allclose(j2.detach().numpy(), g["jsd2_map"], **TIGHT)
    ga, gb = torch.autograd.grad(j2.mean(), [la, lb], retain_graph=True)
    np.testing.assert_allclose(ga.numpy(), g["jsd2_grad_a"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gb.numpy(), g["jsd2_grad_b"], rtol=1e-5, atol=1e-8)
    j3 = oracle.jsd_2d([pa, pb, pc])
    np.testing.assert_allclose(j3.detach().numpy(), g["jsd3_map"], rtol=1e-5, atol=1e-7)
    kl = oracle.kl_divergence_2d(pa, pb.detach(), reduce=True)
    np.testing.assert_allclose(kl.item(), g["kl"], **TIGHT)
    np.testing.assert_allclose(torch.autograd.grad(kl, la, retain_graph=True)[0].numpy(), g["kl_grad_a"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(oracle.kl_divergence_2d(pa, pb).detach().numpy(), g["kl_map"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(oracle.entropy_2d(pa).detach().numpy(), g["entropy_a"], **TIGHT)
    # SURVEY 8c sanity anchors (probe values)
    assert abs(j2.mean().item() - 0.13992848) < 1e-6
    assert abs(ce.item() - 1.85805273) < 1e-6


def test_g1_multiview_jsd(golden):
    """JSD over 4 and 6 views as the reference computes it (script/GM/run_multiview.sh:2-6 run 2 / 4 / 6 segmentators)."""
    g = golden("g1_multiview")
    torch.manual_seed(int(g["seed"]))
    xs = [torch.randn(2, 3, 9, 7) * 1.5 for _ in range(6)]
    for S in (4, 6):
        ls = [x.clone().requires_grad_(True) for x in xs[:S]]
        jm = oracle.jsd_2d([oracle.softmax_channels(x) for x in ls])
        np.testing.assert_allclose(jm.detach().numpy(), g[f"jsd{S}_map"], rtol=1e-5, atol=1e-7)
        for k, gr in enumerate(torch.autograd.grad(jm.mean(), ls)):
            np.testing.assert_allclose(gr.numpy(), g[f"jsd{S}_grad_{k}"], rtol=1e-5, atol=1e-8)


def test_g2_schedulers(golden):
    g = golden("g2_schedulers")
    for tag in ("cot", "adv"):
        b, m, v, mult = g[tag + "_args"]
        mine = [oracle.ramp_value(e, int(b), int(m), v, mult) for e in range(60)]
        np.testing.assert_allclose(mine, g[tag], rtol=1e-12, atol=0)


@pytest.mark.parametrize("arch", ["unet", "enet"])
def test_g3_nets(golden, arch):
    g = golden(f"g3_{arch}")
    C, seed = int(g["C"]), int(g["seed"])
    cases = [("eval", 176), ("eval", 256)] if arch == "unet" else [("train", 64), ("eval", 64), ("eval", 256)]
    for mode, H in cases:
        tag = f"{mode}{H}"
        B = int(g[f"{tag}_shape"][0])
        net = _seeded_net(arch, C, seed)
        net.train() if mode == "train" else net.eval()
        torch.manual_seed(100 + H)
        x = torch.rand(B, 1, H, H)
        t = torch.randint(0, C, (B, H, H))
        x.requires_grad_(True)
        y = net(x)
        np.testing.assert_allclose(_digest(y), g[f"{tag}_logits_digest"], rtol=1e-5)
        if H <= 176:
            np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}_logits"], rtol=1e-5, atol=1e-6)
            loss = oracle.cross_entropy_2d(y, t)
            loss.backward()
            np.testing.assert_allclose(loss.item(), g[f"{tag}_ce"], rtol=1e-6)
            np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}_grad_x"], rtol=1e-4, atol=1e-9)
            names = [k for k, _ in net.named_parameters()]
            assert names == list(g[f"{tag}_grad_names"]), "state_dict key names/order must equal the reference's"
            norms = [p.grad.double().norm().item() for _, p in net.named_parameters()]
            np.testing.assert_allclose(norms, g[f"{tag}_grad_norms"], rtol=1e-4, atol=1e-10)
        if arch == "enet" and mode == "train":
            sd = net.state_dict()
            np.testing.assert_allclose(sd["encoder.initial.batch_norm.running_mean"].numpy(), g[f"{tag}_bn_init_mean"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(sd["encoder.initial.batch_norm.running_var"].numpy(), g[f"{tag}_bn_init_var"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(sd["decoder.layers.4.block1x1_2.1.running_var"].numpy(), g[f"{tag}_bn_last_var"], rtol=1e-4, atol=1e-7)


def test_g5_fgsm(golden):
    g = golden("g5_fgsm_enet")
    C, H = int(g["C"]), int(g["H"])
    net = _seeded_net("enet", C, int(g["seed_net"]))
    net.train()
    gen = torch.Generator().manual_seed(int(g["seed_data"]))
    img = torch.rand(4, 1, H, H, generator=gen)
    gt = torch.randint(0, C, (2, 1, H, H), generator=gen)
    x_adv, noise, probs, gx = oracle.fgsm_generate(net, img, gt, float(g["eps"]))
    np.testing.assert_allclose(probs.detach().numpy(), g["probs"], rtol=1e-5, atol=1e-7)
    # sign() is chaotic where |grad| ~ 0: require agreement wherever the gradient is not tiny
    solid = gx.abs().numpy() > 1e-7
    assert solid.mean() > 0.9
    assert (noise.numpy()[solid] == g["noise"][solid]).all()
    np.testing.assert_allclose(x_adv.numpy()[solid], g["x_adv"][solid], rtol=0, atol=1e-7)


def _batches(seed, n, B, H, C):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        img = torch.rand(B, 1, H, H, generator=gen)
        gt = torch.randint(0, C, (B, 1, H, H), generator=gen)
        out.append((img, gt))
    return out


@pytest.mark.parametrize("tag", ["g5_step_enet_jsd", "g5_step_enet_adv", "g5_step_unet_jsd", "g5_step_unet_adv"])
def test_g5_full_step(golden, tag):
    g = golden(tag)
    arch, C, H, B, n = str(g["arch"]), int(g["C"]), int(g["H"]), int(g["B"]), int(g["n_steps"])
    adv = bool(int(g["train_adv"]))
    models = []
    for s in g["net_seeds"]:
        net = _seeded_net(arch, C, int(s))
        net.train()
        models.append(oracle.OracleModel.make(net))
    lab = [_batches(int(s), n, B, H, C) for s in g["lab_seeds"]]
    unl = _batches(int(g["unl_seed"]), n, B, H, C)
    snap1 = None
    # chaotic points (FGSM sign, first Adam steps ~ lr*sign(g)) amplify ulp differences: compare digests
    for k in range(n):
        r = oracle.cotrain_step(models, [lab[0][k], lab[1][k]], unl[k][0], True, adv,
                                lam_cot=float(g["lam_cot"]), lam_adv=float(g["lam_adv"]), eps=float(g["eps"]))
        np.testing.assert_allclose([s.item() for s in r["sup"]], g["sup"][k][:2], rtol=2e-5)
        np.testing.assert_allclose(r["jsd"].item(), g["jsd"][k], rtol=2e-4, atol=1e-7)
        if adv:
            np.testing.assert_allclose(r["adv"].item(), g["adv"][k], rtol=2e-3, atol=1e-7)
        if k == 0:
            snap1 = [{kk: v.detach().clone() for kk, v in m.net.state_dict().items()} for m in models]
    for j, m in enumerate(models):
        names = list(g[f"m{j}_names"])
        sd = m.net.state_dict()
        d1 = np.stack([_digest(snap1[j][kk]) for kk in names])
        df = np.stack([_digest(sd[kk]) for kk in names])
        # columns 1..3 (abs-sum, l2, max-abs) are stable; signed sum can cancel -> scaled atol
        for col in (1, 2, 3):
            np.testing.assert_allclose(d1[:, col], g[f"m{j}_digest_step1"][:, col], rtol=2e-4, atol=1e-6)
            np.testing.assert_allclose(df[:, col], g[f"m{j}_digest_final"][:, col], rtol=1e-3, atol=1e-5)
        small = str(g[f"m{j}_small_name"])
        np.testing.assert_allclose(snap1[j][small].numpy(), g[f"m{j}_small_step1"], rtol=1e-4, atol=2e-6)
        ps = list(m.net.parameters())
        ea = np.stack([_digest(m.optimizer.state[p]["exp_avg"]) for p in ps])
        np.testing.assert_allclose(ea[:, 2], g[f"m{j}_exp_avg_digest"][:, 2], rtol=5e-3, atol=1e-9)


def test_g6_dice(golden):
    g = golden("g6_dice")
    torch.manual_seed(int(g["seed"]))
    rows2, rows3 = [], []
    for _ in range(2):
        logits = torch.randn(3, 4, 16, 16)
        gt = torch.randint(0, 4, (3, 1, 16, 16))
        rows2.append(oracle.dice_2d(logits, gt))
        rows3.append(oracle.dice_3d(logits, gt).unsqueeze(0))
    log2, log3 = torch.cat(rows2), torch.cat(rows3)
    np.testing.assert_allclose(log2.numpy(), g["log2d"], rtol=1e-6)
    np.testing.assert_allclose(log3.numpy(), g["log3d"], rtol=1e-6)
    np.testing.assert_allclose(log2.mean(0).numpy(), g["means2d"], rtol=1e-6)
    np.testing.assert_allclose(log2.std(0).numpy(), g["stds2d"], rtol=1e-5)


def test_g3_unet_bn(golden):
    """oracle UNet(batchnorm=True) == the reference's UNet_bn (network.py:243-290): train-mode logits / CE / gradients on a
    batch of two, the running statistics it leaves, an eval-mode forward, and the state_dict keys."""
    g = golden("g3_unet_bn")
    C, H = int(g["C"]), int(g["H"])
    torch.manual_seed(int(g["seed"]))
    net = oracle.build_net("unet_bn", C, dropout_p=0.0)
    assert list(net.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    net.train()
    torch.manual_seed(100 + H)
    x = torch.rand(2, 1, H, H)
    t = torch.randint(0, C, (2, H, H))
    x.requires_grad_(True)
    y = net(x)
    loss = oracle.cross_entropy_2d(y, t)
    loss.backward()
    np.testing.assert_allclose(y.detach().numpy(), g["train_logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(loss.item(), g["train_ce"], **TIGHT)
    np.testing.assert_allclose(x.grad.numpy(), g["train_grad_x"], rtol=1e-4, atol=1e-9)
    norms = {k: p.grad.double().norm().item() for k, p in net.named_parameters()}
    np.testing.assert_allclose([norms[str(k)] for k in g["train_grad_names"]], g["train_grad_norms"], rtol=1e-4, atol=1e-9)
    sd = net.state_dict()
    np.testing.assert_allclose(np.stack([_digest(sd[str(k)]) for k in g["bn_keys"]]), g["bn_digest"], rtol=1e-5, atol=1e-7)
    net.eval()
    with torch.no_grad():
        ye = net(x.detach()[:1])
    np.testing.assert_allclose(ye.numpy(), g["eval_logits"], rtol=1e-5, atol=1e-6)


def test_g7_ensembles(golden):
    """oracle soft / hard voting + Dice == Summary.py's Ensembleway + the reference DiceMeter on the same predictions."""
    g = golden("g7_eval")
    C, H = int(g["C"]), int(g["H"])
    nets = []
    for s in g["net_seeds"]:
        net = _seeded_net("enet", C, int(s)).eval()
        nets.append(net)

    def batches(seed, n, B):
        gen = torch.Generator().manual_seed(seed)
        return [(torch.rand(B, 1, H, H, generator=gen), torch.randint(0, C, (B, 1, H, H), generator=gen)) for _ in range(n)]

    val = [b for s, B in zip(g["val_seeds"], g["val_sizes"]) for b in batches(int(s), 1, int(B))]
    val1 = batches(int(g["hard_val_seed"]), int(g["hard_val_batches"]), 1)
    for way, loader in (("soft", val), ("hard", val1)):
        rows2, rows3 = [], []
        with torch.no_grad():
            for img, gt in loader:
                probs = [oracle.softmax_channels(n(img)) for n in nets]
                v = oracle.soft_vote(probs) if way == "soft" else oracle.hard_vote(probs, C)
                rows2.append(oracle.dice_2d(v, gt))
                rows3.append(oracle.dice_3d(v, gt).unsqueeze(0))
        for rows, key in ((rows2, f"{way}_dice2d"), (rows3, f"{way}_dice3d")):
            log = torch.cat(rows)
            np.testing.assert_allclose(torch.stack((log.mean(0), log.std(0)), dim=1).numpy(), g[key], rtol=1e-5, atol=1e-6)

#!/usr/bin/env python3
"""Capture golden vectors from the UNMODIFIED reference (runs only in the build container).

The reference at /root/reference is imported with import stubs for the third-party
modules this image lacks -- none of them is touched by the arithmetic --
driven on seeded synthetic inputs, and its outputs are written as small ``.npz``
fixtures under tests/golden/.  Weights come from the oracle's seeded initializer and are
loaded into the reference nets with ``load_state_dict`` (same key names), so fixtures
carry seeds instead of weight blobs.

Nothing from the reference is copied into the repo: fixtures are inputs (as seeds) and
expected outputs only.  Re-run:  python tools/capture_golden.py
"""
from __future__ import annotations

import collections
import collections.abc
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _install_stubs():
    for name in ("Mapping", "MutableMapping", "Iterable"):
        if not hasattr(collections, name):
            setattr(collections, name, getattr(collections.abc, name))

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

        def __getattr__(self, k):
            return _Anything()

    class _Writer:
        def __init__(self, *a, **k):
            pass

        def add_scalars(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    for name in ("torchvision", "torchvision.models", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchvision.utils", "torchvision.datasets",
                 "skimage", "skimage.io", "skimage.transform", "visdom", "easydict"):
        m = mod(name)
        def _ga(k, _A=_Anything):
            if k.startswith("__"):
                raise AttributeError(k)
            return _A
        m.__getattr__ = _ga  # type: ignore
    sys.modules["skimage.io"].imsave = lambda *a, **k: None
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    mod("tensorboardX", SummaryWriter=_Writer)
    sys.path.insert(0, REF)


_install_stubs()
warnings.filterwarnings("ignore")

import oracle  # noqa: E402  (build-owned initializer only)
from generalframework import ModelMode  # noqa: E402
from generalframework.arch import get_arch  # noqa: E402
from generalframework.loss import CrossEntropyLoss2d, JSD_2D, KL_Divergence_2D, Entropy_2D, get_loss_fn  # noqa: E402
from generalframework.metrics import DiceMeter  # noqa: E402
from generalframework.models import Segmentator  # noqa: E402
from generalframework.scheduler import RampScheduler  # noqa: E402
from generalframework.trainer import CoTrainer  # noqa: E402
from generalframework.trainer import cotraining_totalloss as ref_trainer_mod  # noqa: E402
from generalframework.utils.AEGenerator import FSGMGenerator  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def seeded_state(arch, C, seed):
    """Build-owned initializer: weights from the oracle net under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    kw = {"dropout_p": 0.0} if arch == "unet" else {}
    return oracle.build_net(arch, C, **kw).state_dict()


def ref_net(arch, C, seed):
    net = get_arch(arch, {"num_classes": C})
    net.load_state_dict(seeded_state(arch, C, seed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


# ----------------------------------------------------------------------------- G1 losses
def g1_losses():
    torch.manual_seed(0)
    a, b, c = (torch.randn(2, 4, 8, 8) for _ in range(3))
    t = torch.randint(0, 4, (2, 8, 8))
    t_ign = t.clone()
    t_ign[0, :2] = 255
    out = dict(seed=0)
    la, lb, lc = (x.clone().requires_grad_(True) for x in (a, b, c))
    pa, pb, pc = (torch.softmax(x, 1) for x in (la, lb, lc))
    ce = CrossEntropyLoss2d()(la, t)
    out["ce"] = ce
    out["ce_grad"] = torch.autograd.grad(ce, la, retain_graph=True)[0]
    cei = CrossEntropyLoss2d()(la, t_ign)
    out["ce_ignore"] = cei
    out["ce_ignore_grad"] = torch.autograd.grad(cei, la, retain_graph=True)[0]
    j2 = JSD_2D()([pa, pb])
    out["jsd2_map"] = j2
    g = torch.autograd.grad(j2.mean(), [la, lb], retain_graph=True)
    out["jsd2_grad_a"], out["jsd2_grad_b"] = g
    j3 = JSD_2D()([pa, pb, pc])
    out["jsd3_map"] = j3
    g = torch.autograd.grad(j3.mean(), [la, lb, lc], retain_graph=True)
    out["jsd3_grad_a"], out["jsd3_grad_b"], out["jsd3_grad_c"] = g
    kl = KL_Divergence_2D(reduce=True)(pa, pb.detach())
    out["kl"] = kl
    out["kl_grad_a"] = torch.autograd.grad(kl, la, retain_graph=True)[0]
    out["kl_map"] = KL_Divergence_2D(reduce=False)(pa, pb.detach())
    out["entropy_a"] = Entropy_2D()(pa)
    save("g1_losses", **out)


# ----------------------------------------------------------------------------- G2 schedulers
def g2_sched():
    out = {}
    for tag, args in {"cot": (0, 50, 0.5, -5), "adv": (20, 50, 0.05, -5)}.items():
        s = RampScheduler(*args)
        vals = []
        for _ in range(60):
            vals.append(float(s.value))
            s.step()
        out[tag + "_args"] = np.array(args, dtype=np.float64)
        out[tag] = np.array(vals, dtype=np.float64)
    save("g2_schedulers", **out)


# ----------------------------------------------------------------------------- G3/G4 nets
def _tensor_digest(t: torch.Tensor):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), t.norm().item(), t.abs().max().item()])


def g3_g4_nets():
    # UNet, eval(), 176 (minimum size) full logits + 256 digest; grads at 176
    for arch, C, seed, cases in (("unet", 4, 11, [(1, 176, "eval"), (1, 256, "eval")]),
                                 ("enet", 4, 12, [(2, 64, "train"), (2, 64, "eval"), (1, 256, "eval")])):
        out = dict(arch=arch, C=C, seed=seed)
        for (B, H, mode) in cases:
            net = ref_net(arch, C, seed)
            net.train() if mode == "train" else net.eval()
            torch.manual_seed(100 + H)
            x = torch.rand(B, 1, H, H)
            t = torch.randint(0, C, (B, H, H))
            tag = f"{mode}{H}"
            out[f"{tag}_shape"] = np.array([B, H])
            x.requires_grad_(True)
            y = net(x)
            out[f"{tag}_logits_digest"] = _tensor_digest(y)
            if H <= 176:
                out[f"{tag}_logits"] = y
                loss = CrossEntropyLoss2d()(y, t)
                loss.backward()
                out[f"{tag}_ce"] = loss
                out[f"{tag}_grad_x"] = x.grad
                names, norms = [], []
                for k, p in net.named_parameters():
                    names.append(k)
                    norms.append(p.grad.double().norm().item())
                out[f"{tag}_grad_names"] = np.array(names)
                out[f"{tag}_grad_norms"] = np.array(norms)
            if arch == "enet" and mode == "train":
                sd = net.state_dict()
                out[f"{tag}_bn_init_mean"] = sd["encoder.initial.batch_norm.running_mean"]
                out[f"{tag}_bn_init_var"] = sd["encoder.initial.batch_norm.running_var"]
                out[f"{tag}_bn_last_mean"] = sd["decoder.layers.4.block1x1_2.1.running_mean"]
                out[f"{tag}_bn_last_var"] = sd["decoder.layers.4.block1x1_2.1.running_var"]
        save(f"g3_{arch}", **out)


# ----------------------------------------------------------------------------- G5 full steps
class _FakeDataset:
    training = ModelMode.EVAL

    def set_mode(self, mode):
        self.training = mode


class _FakeLoader(list):
    """Just enough DataLoader surface for CoTrainer._train_loop and iterator_."""

    def __init__(self, batches, batch_size):
        super().__init__(batches)
        self.batch_size = batch_size
        self.dataset = _FakeDataset()


def _batches(seed, n, B, H, C):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        img = torch.rand(B, 1, H, H, generator=g)
        gt = torch.randint(0, C, (B, 1, H, H), generator=g)
        out.append([[img, gt], None, [f"s{seed}_{i}_{j}" for j in range(B)]])
    return out


class _Recorder(torch.nn.Module):
    def __init__(self, inner, log):
        super().__init__()
        self.inner, self.log = inner, log

    def forward(self, *a, **k):
        r = self.inner(*a, **k)
        self.log.append(r.detach().clone())
        return r


def g5_step(tag, arch, C, H, B, n_steps, train_adv, lam_cot=0.5, lam_adv=0.05, eps=0.03):
    seeds = (21, 22)
    segs = []
    for s in seeds:
        seg = Segmentator({"name": arch, "num_classes": C},
                          {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4},
                          {"name": "StepLR", "step_size": 90, "gamma": 0.1})
        seg.torchnet.load_state_dict(seeded_state(arch, C, s))
        for m in seg.torchnet.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        segs.append(seg)
    lab = [_FakeLoader(_batches(31 + i, n_steps, B, H, C), B) for i in range(2)]
    unl = _FakeLoader(_batches(41, n_steps, B, H, C), B)
    sup_log, jsd_log, adv_log = [], [], []
    crit = {"sup": _Recorder(get_loss_fn("cross_entropy"), sup_log),
            "jsd": _Recorder(get_loss_fn("jsd"), jsd_log),
            "adv": get_loss_fn("jsd")}
    tmp = tempfile.mkdtemp(prefix="golden_")
    tr = CoTrainer(segmentators=segs, labeled_dataloaders=lab, unlabeled_dataloader=unl, val_dataloader=unl,
                   criterions=crit, max_epoch=1, save_dir=tmp, device="cpu", axises=list(range(1, C)),
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": lam_cot},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": lam_adv},
                   adv_training_dict={"eplision": eps}, use_tqdm=False)
    orig_adv = tr._FSGM_adv_training

    def rec_adv(*a, **k):
        r = orig_adv(*a, **k)
        adv_log.append(r.detach().clone())
        return r

    tr._FSGM_adv_training = rec_adv
    out = dict(arch=arch, C=C, H=H, B=B, n_steps=n_steps, train_adv=int(train_adv), lam_cot=lam_cot,
               lam_adv=lam_adv, eps=eps, net_seeds=np.array(seeds), lab_seeds=np.array([31, 32]), unl_seed=41)
    np.random.seed(1234)
    snap_after = {}
    # run the reference loop body n_steps times by shadowing the hard-coded range(300) (:191,:203)
    counter = {"n": 0}

    def short_range(*a):
        if a == (300,):
            return range(n_steps)
        return range(*a)

    ref_trainer_mod.range = short_range
    # snapshot the weights after step 1: wrap second model's optimizer.step
    orig_steps = [seg.optimizer.step for seg in segs]

    def make_step(i):
        def stepper(*a, **k):
            r = orig_steps[i](*a, **k)
            if i == 1:
                counter["n"] += 1
                if counter["n"] == 1:
                    for j, seg in enumerate(segs):
                        snap_after[j] = {k2: v.detach().clone() for k2, v in seg.torchnet.state_dict().items()}
            return r
        return stepper

    for i, seg in enumerate(segs):
        seg.optimizer.step = make_step(i)
    try:
        dice_lab, dice_unl = tr._train_loop(lab, unl, epoch=0, mode=ModelMode.TRAIN, save=False,
                                            train_jsd=True, train_adv=train_adv)
    finally:
        del ref_trainer_mod.range
    # with FGSM the recorded criterion is also called inside FSGMGenerator: columns = (sup0, sup1, fgsm_ce)
    out["sup"] = torch.stack(sup_log).reshape(n_steps, 3 if train_adv else 2)
    out["jsd"] = torch.stack([j.mean() for j in jsd_log])
    if train_adv:
        out["adv"] = torch.stack(adv_log)
    out["dice_lab"] = dice_lab
    out["dice_unl"] = dice_unl
    for j, seg in enumerate(segs):
        sd = seg.torchnet.state_dict()
        names = [k for k in sd if sd[k].dtype.is_floating_point]
        out[f"m{j}_names"] = np.array(names)
        out[f"m{j}_digest_step1"] = np.stack([_tensor_digest(snap_after[j][k]) for k in names])
        out[f"m{j}_digest_final"] = np.stack([_tensor_digest(sd[k]) for k in names])
        # optimizer moments digest (exp_avg / exp_avg_sq) in parameter order
        st = seg.optimizer.state
        ps = list(seg.torchnet.parameters())
        out[f"m{j}_exp_avg_digest"] = np.stack([_tensor_digest(st[p]["exp_avg"]) for p in ps])
        out[f"m{j}_exp_avg_sq_digest"] = np.stack([_tensor_digest(st[p]["exp_avg_sq"]) for p in ps])
        # one small tensor in full for element-wise comparison
        small = "final.bias" if arch == "unet" else "decoder.layers.5.bias"
        out[f"m{j}_small_name"] = small
        out[f"m{j}_small_step1"] = snap_after[j][small]
        out[f"m{j}_small_final"] = sd[small]
    save(tag, **out)


def g5_fgsm(arch="enet", C=4, H=64):
    """FSGMGenerator alone (AEGenerator.py:16-51) on the labeled+unlabeled concat."""
    net = ref_net(arch, C, 23)
    net.train()
    g = torch.Generator().manual_seed(51)
    img = torch.rand(4, 1, H, H, generator=g)
    gt = torch.randint(0, C, (2, 1, H, H), generator=g)
    x_adv, noise, probs = FSGMGenerator(net, eplision=0.03)(img.clone(), gt, CrossEntropyLoss2d())
    save("g5_fgsm_enet", seed_net=23, seed_data=51, C=C, H=H, eps=0.03, x_adv=x_adv, noise=noise, probs=probs)


# ----------------------------------------------------------------------------- G6 dice
def g6_dice():
    torch.manual_seed(7)
    logits = torch.randn(3, 4, 16, 16)
    gt = torch.randint(0, 4, (3, 1, 16, 16))
    m2, m3 = DiceMeter(method="2d", report_axises=[1, 2, 3], C=4), DiceMeter(method="3d", report_axises=[1, 2, 3], C=4)
    m2.add(logits, gt)
    m3.add(logits, gt)
    logits2 = torch.randn(3, 4, 16, 16)
    gt2 = torch.randint(0, 4, (3, 1, 16, 16))
    m2.add(logits2, gt2)
    m3.add(logits2, gt2)
    (rm, rs), (means, stds) = m2.value()
    (rm3, rs3), (means3, stds3) = m3.value()
    save("g6_dice", seed=7, log2d=m2.log, log3d=m3.log, report_mean2d=rm, report_std2d=rs, means2d=means,
         stds2d=stds, report_mean3d=rm3, report_std3d=rs3, means3d=means3, stds3d=stds3)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g5", "g6"]
    if "g1" in which:
        g1_losses()
    if "g2" in which:
        g2_sched()
    if "g3" in which:
        g3_g4_nets()
    if "g5" in which:
        g5_fgsm()
        g5_step("g5_step_enet_jsd", "enet", 4, 64, 2, 3, train_adv=False)
        g5_step("g5_step_enet_adv", "enet", 4, 64, 2, 3, train_adv=True)
        g5_step("g5_step_unet_jsd", "unet", 4, 176, 1, 2, train_adv=False)
        g5_step("g5_step_unet_adv", "unet", 4, 176, 1, 2, train_adv=True)
    if "g6" in which:
        g6_dice()

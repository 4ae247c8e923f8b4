"""Host-side logic of the drop-in boundary on CPU: CoTrainer/_run_step op order, iterator
caching, schedulers, registry/CLI helpers, checkpoint format, flat parameter storage.
The arithmetic here comes from INJECTED oracle modules (test doubles) -- the product kernels are
HIP-only and are exercised by the -m gpu tests."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from helpers import FakeLoader, batches, digest


# ---- test doubles built on the oracle ------------------------------------------------------------
class OracleCE(nn.Module):
    ignore_index = 255

    def forward(self, outputs, targets):
        return oracle.cross_entropy_2d(outputs, targets)


class OracleJSD(nn.Module):
    def forward(self, probs):
        return oracle.jsd_2d(probs)


class OracleKL(nn.Module):
    def __init__(self, reduce=False, eps=1e-10):
        super().__init__()
        self.reduce, self.eps = reduce, eps

    def forward(self, p, y):
        return oracle.kl_divergence_2d(p, y, self.reduce, self.eps)


class OracleFGSM:
    def __init__(self, net, eplision=0.05):
        self.net, self.eps = net, eplision

    def __call__(self, img, gt, criterion):
        x_adv, noise, probs, _ = oracle.fgsm_generate(self.net, img, gt, self.eps)
        return x_adv, noise, probs


class OracleDice:
    def __init__(self, method='2d', report_axises='all', C=4):
        self.method, self.report_axis, self.C, self.diceLog = method, report_axises, C, []

    def add(self, pred, gt):
        d = oracle.dice_2d(pred, gt) if self.method == '2d' else oracle.dice_3d(pred, gt).unsqueeze(0)
        self.diceLog.append(d)

    def value(self):
        log = torch.cat(self.diceLog) if self.diceLog else torch.zeros(1, self.C)
        rm = log.mean(1) if self.report_axis == 'all' else log[:, self.report_axis].mean(1)
        return (rm.mean(), rm.std()), (log.mean(0), log.std(0))


def _make_trainer(tmp_path, monkeypatch, arch, C, H, B, n_steps, seeds=(21, 22), lam_cot=0.5, lam_adv=0.05, eps=0.03):
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    from dct_amd.trainer import cotraining_totalloss as mod
    monkeypatch.setattr(mod, "DiceMeter", OracleDice)
    segs = []
    for s in seeds:
        torch.manual_seed(s)
        net = oracle.build_net(arch, C, **({"dropout_p": 0.0} if arch == "unet" else {}))
        segs.append(Segmentator({"name": arch, "num_classes": C}, {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4},
                                {"name": "StepLR", "step_size": 90, "gamma": 0.1}, torchnet=net,
                                softmax_fn=oracle.softmax_channels))
    lab = [FakeLoader(batches(31 + i, n_steps, B, H, C), B) for i in range(len(seeds))]
    unl = FakeLoader(batches(41, n_steps, B, H, C), B)
    crit = {"sup": OracleCE(), "jsd": OracleJSD(), "adv": OracleJSD()}
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=str(tmp_path), device="cpu", axises=list(range(1, C)),
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": lam_cot},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": lam_adv},
                   adv_training_dict={"eplision": eps}, use_tqdm=False, steps_per_epoch=n_steps)
    tr._fsgm_cls = OracleFGSM
    tr._kl_override = OracleKL
    return tr, lab, unl


# ---- the step: host logic == reference's loop body -------------------------------------------------
@pytest.mark.parametrize("tag", ["g5_step_enet_jsd", "g5_step_enet_adv"])
def test_train_loop_host_logic_reproduces_reference_golden(golden, tmp_path, monkeypatch, tag):
    """CoTrainer._train_loop/_run_step (generic path) driven with oracle modules reproduces what the
    reference's own _train_loop produced (captured in tests/golden): same op order, zero_grad after the
    forwards, FGSM on this step's cached batches, one backward, all optimizers stepped."""
    from dct_amd import ModelMode
    g = golden(tag)
    n, adv = int(g["n_steps"]), bool(int(g["train_adv"]))
    tr, lab, unl = _make_trainer(tmp_path, monkeypatch, "enet", int(g["C"]), int(g["H"]), int(g["B"]), n)
    assert not tr._fused_ok()
    log = []
    orig = tr._run_step
    tr._run_step = lambda *a, **k: log.append(orig(*a, **k)) or log[-1]
    np.random.seed(1234)
    dl, du = tr._train_loop(lab, unl, epoch=0, mode=ModelMode.TRAIN, save=False, train_jsd=True, train_adv=adv)
    for k in range(n):
        np.testing.assert_allclose([s.item() for s in log[k]["sup"]], g["sup"][k][:2], rtol=2e-5)
        np.testing.assert_allclose(log[k]["jsd"].item(), g["jsd"][k], rtol=2e-4, atol=1e-7)
        if adv:
            np.testing.assert_allclose(log[k]["adv"].item(), g["adv"][k], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(dl.numpy(), g["dice_lab"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(du.numpy(), g["dice_unl"], rtol=1e-5, atol=1e-6)
    for j, seg in enumerate(tr.segmentators):
        sd = seg.torchnet.state_dict()
        names = list(g[f"m{j}_names"])
        df = np.stack([digest(sd[k]) for k in names])
        for col in (1, 2, 3):
            np.testing.assert_allclose(df[:, col], g[f"m{j}_digest_final"][:, col], rtol=1e-3, atol=1e-5)


def test_run_step_contract_and_rng(tmp_path, monkeypatch):
    tr, lab, unl = _make_trainer(tmp_path, monkeypatch, "enet", 2, 32, 2, 2)
    for s in tr.segmentators:
        s.train()
    lb = [(lab[i][0][0][0], lab[i][0][0][1]) for i in range(2)]
    ub = (unl[0][0][0], unl[0][0][1])
    np.random.seed(7)
    ref_state = np.random.RandomState(7)
    out = tr._run_step(lb, ub, True, True)          # adv_choice drawn inside: numpy RNG consumed exactly once
    ref_state.choice([0, 1], 2, replace=False)
    assert np.random.randint(1 << 30) == ref_state.randint(1 << 30)
    assert set(out) >= {"sup", "jsd", "adv", "preds", "unlab_probs"}
    assert len(out["sup"]) == 2 and out["preds"][0].shape == (2, 2, 32, 32) and out["unlab_probs"][1].shape == (2, 2, 32, 32)
    np.random.seed(7)
    st = np.random.get_state()[1].copy()
    tr._run_step(lb, ub, True, False)               # no adversarial term -> RNG untouched
    assert (np.random.get_state()[1] == st).all()
    # gradients were zeroed AFTER the forwards and every optimizer stepped once
    for s in tr.segmentators:
        assert all(int(v["step"]) == 2 for v in s.optimizer.state.values())


def test_fsgm_adv_training_reuses_cached_batches(tmp_path, monkeypatch):
    from dct_amd.utils import iterator_
    tr, lab, unl = _make_trainer(tmp_path, monkeypatch, "enet", 2, 32, 2, 3)
    its = [iterator_(l) for l in lab]
    uit = iterator_(unl)
    for it in its + [uit]:
        it.__next__()
    seen = {}

    class Spy(OracleFGSM):
        def __call__(self, img, gt, criterion):
            seen["img"], seen["gt"] = img.detach().clone(), gt.clone()
            return super().__call__(img, gt, criterion)

    tr._fsgm_cls = Spy
    loss = tr._FSGM_adv_training((tr.segmentators[0], tr.segmentators[1]), its, uit, eplision=0.03)
    assert loss.dim() == 0 and torch.isfinite(loss)
    assert torch.equal(seen["img"], torch.cat((lab[1][0][0][0], unl[0][0][0]), 0))   # labeled batch of model b + unlabeled
    assert torch.equal(seen["gt"], lab[1][0][0][1])
    assert torch.equal(its[1].__cache__()[0][0], lab[1][0][0][0])                    # iterators not advanced


def test_cotrainer_constructor_contract(tmp_path, monkeypatch):
    from dct_amd.trainer import CoTrainer
    tr, lab, unl = _make_trainer(tmp_path, monkeypatch, "enet", 2, 32, 2, 1)
    segs, crit = tr.segmentators, tr.criterions
    kw = dict(max_epoch=1, save_dir=str(tmp_path), device="cpu", use_tqdm=False,
              cot_scheduler_dict={"name": "RampScheduler", "begin_epoch": 0, "max_epoch": 50, "max_value": 0.5, "ramp_mult": -5},
              adv_scheduler_dict={"name": "RampScheduler", "begin_epoch": 20, "max_epoch": 50, "max_value": 0.05, "ramp_mult": -5})
    with pytest.raises(AssertionError):
        CoTrainer(segs, lab[:1], unl, unl, crit, **kw)                     # S models need S labeled loaders
    with pytest.raises(AssertionError):
        CoTrainer([segs[0], segs[0]], lab, unl, unl, crit, **kw)           # distinct instances
    with pytest.raises(AssertionError):
        CoTrainer(segs, lab, unl, unl, {"sup": crit["sup"], "jsd": crit["jsd"]}, **kw)   # keys == {sup,jsd,adv}
    ok = CoTrainer(segs, lab, unl, unl, crit, **kw)
    assert ok.C == 2 and abs(ok.cot_scheduler.value - 0.00336897) < 1e-8 and ok.adv_scheduler.value == 0.0
    ok.schedulerStep()
    assert ok.cot_scheduler.epoch == 1


def test_eval_loop_and_checkpoint_format(tmp_path, monkeypatch):
    from dct_amd import ModelMode
    tr, lab, unl = _make_trainer(tmp_path, monkeypatch, "enet", 2, 32, 2, 2)
    with torch.no_grad():
        d2, d3 = tr._eval_loop(unl, epoch=0, mode=ModelMode.EVAL)
    assert d2.shape == (2, 2, 2) and d3.shape == (2, 2, 2)
    tr.checkpoint(torch.tensor([0.5, 0.25]), epoch=3)
    ck = torch.load(os.path.join(str(tmp_path), "best_1.pth"), weights_only=False)
    assert set(ck) == {"segmentator", "best_score", "best_epoch"} and ck["best_epoch"] == 3
    assert set(ck["segmentator"]) == {"arch_dict", "optim_dict", "scheduler_dict", "net_state_dict",
                                      "optim_state_dict", "scheduler_state_dict"}
    tr.checkpoint(torch.tensor([0.4, 0.3]), epoch=4)            # only model 1 improved
    assert torch.load(os.path.join(str(tmp_path), "best_0.pth"), weights_only=False)["best_epoch"] == 3
    assert torch.load(os.path.join(str(tmp_path), "best_1.pth"), weights_only=False)["best_epoch"] == 4
    # DataParallel-style keys are accepted on load (segmentators.py:88-93)
    sd = copy.deepcopy(tr.segmentators[0].state_dict)
    sd["net_state_dict"] = {"module." + k: v for k, v in sd["net_state_dict"].items()}
    tr.segmentators[1].load_state_dict(sd)


# ---- small host utilities -------------------------------------------------------------------------
def test_iterator_wraps_and_caches():
    from dct_amd.utils import iterator_
    it = iterator_([1, 2, 3])
    assert [it.__next__() for _ in range(5)] == [1, 2, 3, 1, 2]
    assert it.__cache__() == 2
    fresh = iterator_([9])
    with pytest.warns(UserWarning):
        assert fresh.__cache__() == 9


def test_schedulers_match_reference_golden(golden):
    from dct_amd.scheduler import RampScheduler, ConstantScheduler, RampDownScheduler
    g = golden("g2_schedulers")
    for tag in ("cot", "adv"):
        s = RampScheduler(*g[tag + "_args"])
        vals = []
        for _ in range(60):
            vals.append(s.value)
            s.step()
        np.testing.assert_allclose(vals, g[tag], rtol=1e-12)
    c = ConstantScheduler(2, 0.3)
    assert [c.value, (c.step(), c.step(), c.value)[2]] == [0.0, 0.3]
    r = RampDownScheduler(10, 1.0, -5, 0.1, 5)
    assert r.value == 1.0
    st = r.state_dict()
    r2 = RampDownScheduler(10, 1.0, -5, 0.1, 5)
    r.step()
    r2.load_state_dict({**st, "epoch": 1})
    assert r.value == r2.value


def test_registry_and_dict_merge():
    from dct_amd.loss import get_loss_fn
    from dct_amd.utils import dict_merge
    with pytest.raises(ValueError):
        get_loss_fn("no_such_loss")
    assert type(get_loss_fn("jsd")).__name__ == "JSD_2D"
    # overrides as a command line delivers them (strings): a replaced leaf takes the type of the value it replaces
    args = {"Trainer": {"max_epoch": "3"}, "Arch": {"name": "unet"}, "StartTraining": {"train_jsd": "True"}, "New": {"k": 1}}
    cfg = {"Trainer": {"max_epoch": 300, "device": "cuda:0"}, "Arch": {"name": "enet", "num_classes": 4},
           "StartTraining": {"train_jsd": False}}
    out = dict_merge(cfg, args, True)
    assert out["Trainer"] == {"max_epoch": 3, "device": "cuda:0"} and out["Arch"] == {"name": "unet", "num_classes": 4}
    assert out["StartTraining"]["train_jsd"] is True and out["New"] == {"k": 1}
    assert dict_merge({"a": 1}, None, True) == {"a": 1} and dict_merge({"a": 1}, {"a": 2}) is None
    # the reference's leaf rule (utils/utils.py:345-349): a value that does not convert is kept AS GIVEN -- '0.5' against an int
    # default stays the string (int('0.5') raises), it does not become 0; bool defaults read 'false' / 'True' in any case
    out = dict_merge({"max_epoch": 300, "lr": 1e-3, "flag": True, "axes": [1, 2], "name": "enet", "none": None},
                     {"max_epoch": "0.5", "lr": "0.01", "flag": "false", "axes": "[1, 2, 3]", "name": 7, "none": "x"}, True)
    assert out == {"max_epoch": "0.5", "lr": 0.01, "flag": False, "axes": [1, 2, 3], "name": "7", "none": "x"}
    assert dict_merge({"flag": False, "n": 3}, {"flag": "TRUE", "n": 4.0}, True) == {"flag": True, "n": 4}
    assert dict_merge({"flag": False}, {"flag": "no_such_name"}, True) == {"flag": "no_such_name"}
    # the trainer's use: per-model report dictionaries merged leaf by leaf (cotraining_totalloss._report_dict)
    assert dict_merge({"S0": {"DSC1": 0.5}}, {"S0": {"DSC": 0.25}}, True) == {"S0": {"DSC1": 0.5, "DSC": 0.25}}


def test_tensor_predicates():
    from dct_amd.utils import class2one_hot, one_hot, probs2one_hot, simplex, sset, uniq
    p = torch.softmax(torch.randn(2, 3, 5, 4), 1)
    assert simplex(p) and not simplex(p * 2)
    oh = probs2one_hot(p)
    assert oh.shape == p.shape and one_hot(oh) and oh.dtype == torch.int32
    seg = torch.randint(0, 3, (5, 4))
    assert class2one_hot(seg, 3).shape == (1, 3, 5, 4) and uniq(seg) <= {0, 1, 2} and sset(seg, range(3))
    assert torch.equal(class2one_hot(seg, 3)[0].argmax(0), seg)


def test_arch_registry_names_match_reference_state_dict():
    from dct_amd.arch import get_arch
    net = get_arch("unet", {"num_classes": 4})
    o = oracle.build_net("unet", 4)
    assert list(net.state_dict()) == list(o.state_dict())
    assert all(net.state_dict()[k].shape == v.shape for k, v in o.state_dict().items())
    with pytest.raises(AssertionError):
        get_arch("fcn8", {"num_classes": 4})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.rand(1, 1, 176, 176))                      # product path never computes on the CPU


def test_flat_params_survive_reallocation():
    from dct_amd.arch import get_arch
    net = get_arch("unet", {"num_classes": 2})
    fp = net.flat_params
    assert fp.ensure() and fp.is_flat() and not fp.ensure()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    w = net.dec2.down.at(0).weight
    assert w.shape == (128, 64, 3, 3) and w.stride() == (576, 1, 192, 64)      # logical OIHW, physical [O][kh][kw][I]
    assert net.center.at(5).weight.shape == (1024, 512, 2, 2)                   # ConvTranspose2d layout
    assert fp.total % 64 == 0 and fp.total >= sum(p.numel() for p in net.parameters())
    net.double().float()                                                        # re-allocates every parameter
    assert not fp.is_flat() and fp.ensure()
    assert all(torch.equal(net.state_dict()[k], v) for k, v in before.items())
    assert fp.ensure_grads() and fp.grads_attached()
    assert w.grad.data_ptr() == fp.gflat.data_ptr() + 4 * fp.offsets[[id(p) for p in fp.params].index(id(w))]
    net.zero_grad()                                                             # set_to_none detaches
    assert not fp.grads_attached()
    sd = oracle.build_net("unet", 2).state_dict()
    net.load_state_dict(sd)
    assert fp.is_flat() and torch.equal(net.final.weight.detach(), sd["final.weight"])


def test_dropout_seed_follows_torch_seed_and_differs_between_models():
    """co-trained models must not share dropout masks; a user seed controls them."""
    from dct_amd.arch import get_arch
    torch.manual_seed(7)
    a, b = get_arch("unet", {"num_classes": 4}), get_arch("unet", {"num_classes": 4})
    torch.manual_seed(7)
    c = get_arch("unet", {"num_classes": 4})
    assert a.dropout_seed != b.dropout_seed
    assert a.dropout_seed == c.dropout_seed
    assert 0 <= a.dropout_seed < 2 ** 62


def test_fused_adam_load_state_dict_takes_the_loaded_step():
    """restoring a checkpoint into an optimizer that already stepped further must take the loaded step count
    (torch.optim.Adam does), not the larger one."""
    from dct_amd.arch import get_arch
    from dct_amd.optim import FusedAdam
    net = get_arch("enet", {"num_classes": 2})
    opt = FusedAdam(net.parameters(), flat=net.flat_params, lr=1e-3)
    opt._ensure_state()
    opt._steps = 5
    sd = copy.deepcopy(opt.state_dict())
    assert all(int(float(st["step"])) == 5 for st in sd["state"].values())
    opt._steps = 40                       # trained on ...
    opt._state_views()
    opt.load_state_dict(sd)               # ... then restored
    opt._ensure_state()
    assert opt._steps == 5
    assert all(int(float(st["step"])) == 5 for st in opt.state_dict()["state"].values())


def test_run_step_normalises_loader_dtypes(tmp_path, monkeypatch):
    """uint8 / int32 labels and double images from a loader are cast once at the step boundary."""
    tr, lab, unl = _make_trainer(tmp_path, monkeypatch, "enet", 2, 16, 2, 1)
    for s in tr.segmentators:
        s.train()
    lb = []
    for i in range(2):
        img, gt = lab[i][0][0]
        lb.append((img.double(), gt.to(torch.uint8)))
    seen = {}
    orig = tr._run_step_generic

    def spy(lab_, unl_, *a):
        seen["lab"] = [(x.dtype, y.dtype, x.is_contiguous()) for x, y in lab_]
        seen["unl"] = unl_[0].dtype
        return orig(lab_, unl_, *a)
    tr._run_step_generic = spy
    out = tr._run_step(lb, (unl[0][0][0].half(), unl[0][0][1].int()), True, False)
    assert seen["lab"] == [(torch.float32, torch.int64, True)] * 2 and seen["unl"] == torch.float32
    assert all(torch.isfinite(s) for s in out["sup"])


def test_step_scheduling_switches_on_cpu_modules(tmp_path, monkeypatch):
    """The scheduling switches are HIP-path decisions: with plain nn.Modules on the CPU the generic step runs, no layout is
    chosen and no stream is created, whatever the switches say (trainer/stream_sched.py is imported but idle)."""
    from dct_amd.trainer.stream_sched import EagerSchedule
    tr, lab, unl = _make_trainer(tmp_path, monkeypatch, "enet", 2, 32, 2, 2)
    for s in tr.segmentators:
        s.train()
    assert isinstance(tr._sched, EagerSchedule) and tr._sched.capturing is False
    assert tr.segmented_graphs is None and tr._use_segments() is False             # oracle modules do not ask for segments
    tr.segmented_graphs = True
    assert tr._use_segments() is True
    tr.segmented_graphs = None
    lb = [(lab[i][0][0][0], lab[i][0][0][1]) for i in range(2)]
    np.random.seed(3)
    tr._run_step(lb, (unl[0][0][0], unl[0][0][1]), True, True, (0, 1))
    assert tr._step_hint_adv_chain is False        # needs batch-independent HIP networks with gradient overwrite
    assert tr._streams() is None and tr._stream_dealer() is None and tr._step_graphs is None
    # the eager schedule's null stream is a no-op context
    with tr._sched.on(None):
        pass
    tr._sched.wait([])
    seen = []
    tr._sched.call(lambda: seen.append(1))
    assert seen == [1]


def test_refused_capture_filter_recognises_the_runtime_messages():
    """step_graph keeps a step shape on eager launches only for refused captures; the library's own launch errors propagate."""
    from dct_amd.trainer.step_graph import _is_refused_capture
    for msg in ("HIP error: operation not permitted when stream is capturing", "capturing stream has unjoined work",
                "operation would make the legacy stream depend on a capturing blocking stream", "hipGraphInstantiate failed"):
        assert _is_refused_capture(RuntimeError(msg)), msg
    for msg in ("dct_amd: dct_conv2d failed (DCT_ERR_BAD_ARG)", "shape mismatch", "dct_amd: dct_enet_conv failed (DCT_ERR_LAUNCH)"):
        assert not _is_refused_capture(RuntimeError(msg)), msg

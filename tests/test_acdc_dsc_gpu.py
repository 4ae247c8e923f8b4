"""BASELINE.json: "DSC within 0.2 of the reference on ACDC at equal steps" -- on the vendored ACDC subset.

The reference side is a golden (tests/golden/g9_acdc.npz): the UNMODIFIED reference CoTrainer (2 x Enet, CE + JSD, bs 4 + 4,
fp32 CPU) trained for one short epoch on real ACDC slices and validated per patient (tools/capture_golden.py::g9_acdc).  Here
the HIP bf16 trainer runs the same number of steps from the same initial weights, fed by the decoded-once device cache
(dct_amd.dataset.CachedLoader) -- which must serve exactly the batches the reference's DataLoaders served."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from conftest import GOLDEN  # noqa: E402

DEV = "cuda:0"
SUB = os.path.join(GOLDEN, "acdc_subset")
needs_acdc = pytest.mark.skipif(not os.path.isdir(os.path.join(SUB, "train", "img")), reason="tests/golden/acdc_subset is not in this checkout")
REGEX = r"(patient\d+_\d+)_\d+"


def _loaders(device):
    from torch.utils.data import DataLoader
    from dct_amd.dataset import MedicalImageDataset, PatientSampler, extract_patients, segment_transform, to_cached_loaders
    kw = dict(root_dir=SUB, subfolders=["img", "gt"], transform=segment_transform((256, 256)), augment="PILaugment",
              pin_memory=False, quite=True)
    train_set, val_set = MedicalImageDataset(mode="train", **kw), MedicalImageDataset(mode="val", **kw)
    base = DataLoader(train_set, batch_size=4, shuffle=True, drop_last=True, num_workers=0)
    labs = [extract_patients(base, ["1", "2"]) for _ in range(2)]
    unl = extract_patients(DataLoader(MedicalImageDataset(mode="train", **kw), batch_size=4, shuffle=True, drop_last=True, num_workers=0),
                           ["3", "4", "5"])
    val = DataLoader(val_set, batch_sampler=PatientSampler(val_set, REGEX, shuffle=False, quite=True))
    return to_cached_loaders(labs, unl, val, device=device)


@needs_acdc
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_acdc_subset_dsc_matches_reference_at_equal_steps(golden, tmp_path, dtype):
    from dct_amd import ModelMode
    from dct_amd.loss import get_loss_fn
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    from dct_amd.trainer import cotraining_totalloss as mod
    g = golden("g9_acdc")
    n, C = int(g["n_steps"]), int(g["C"])
    labs, unl, val = _loaders(DEV)
    assert labs[0].cache.img.is_cuda and labs[0].cache.img.dtype == torch.uint8
    segs = []
    for s in g["net_seeds"]:
        torch.manual_seed(int(s))
        sd = oracle.build_net("enet", C).state_dict()
        seg = Segmentator({"name": "enet", "num_classes": C, "compute_dtype": dtype}, {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4},
                          {"name": "StepLR", "step_size": 90, "gamma": 0.1})
        seg.torchnet.load_state_dict(sd)
        segs.append(seg)
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    tr = CoTrainer(segs, labs, unl, val, crit, max_epoch=1, save_dir=str(tmp_path), device=DEV, axises=[1, 2, 3],
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=n)
    names, sups = [], []
    orig_iter = mod.iterator_

    class rec_iter(orig_iter):
        def __next__(self):
            b = super().__next__()
            if isinstance(b, (list, tuple)) and len(b) == 3:     # (the progress-report toggle is an iterator_ over two strings)
                names.append(",".join(b[2]))
            return b
    orig_step = tr._run_step

    def rec_step(*a, **k):
        out = orig_step(*a, **k)
        sups.append(out["sup"])
        return out
    mod.iterator_, tr._run_step = rec_iter, rec_step
    np.random.seed(1234)
    try:
        tr._train_loop(labs, unl, epoch=0, mode=ModelMode.TRAIN, save=False, train_jsd=True, train_adv=False)
    finally:
        mod.iterator_ = orig_iter
    # 1. the data path served the reference's batches, in its order
    assert np.array(names).reshape(n, 3).tolist() == [[str(x) for x in row] for row in g["batch_names"]]
    # 2. the loss fell as the reference's did
    sup = np.array([[float(v) for v in s] for s in sups])
    ref = g["sup"]
    assert sup[-20:].mean() < 0.7 * sup[:10].mean()
    assert abs(sup[-20:].mean() - ref[-20:].mean()) <= 0.25 * ref[-20:].mean(), (sup[-20:].mean(), ref[-20:].mean())
    # 3. validation Dice per patient (3-D) and per slice (2-D), foreground mean per model: north_star allows 0.2
    with torch.no_grad():
        v2, v3 = tr._eval_loop(val, epoch=0, mode=ModelMode.EVAL, save=False)
    for mine, theirs, what in ((v3, g["val_dice3d"], "3-D"), (v2, g["val_dice2d"], "2-D")):
        a, b = mine[:, 1:, 0].mean(1).numpy(), theirs[:, 1:, 0].mean(1)
        print(what, "foreground DSC  HIP", a, " reference", b)
        assert np.all(np.abs(a - b) <= 0.2), (what, a, b)
        assert abs(a.mean() - b.mean()) <= 0.1


# ------------------------------------------------------------------------------------------------ a DSC check with teeth
def _loaders_all_labeled(device, bs):
    from torch.utils.data import DataLoader
    from dct_amd.dataset import MedicalImageDataset, PatientSampler, extract_patients, segment_transform, to_cached_loaders
    everyone = ["1", "2", "3", "4", "5"]
    kw = dict(root_dir=SUB, subfolders=["img", "gt"], transform=segment_transform((256, 256)), augment="PILaugment",
              pin_memory=False, quite=True)
    train_set, val_set = MedicalImageDataset(mode="train", **kw), MedicalImageDataset(mode="val", **kw)
    base = DataLoader(train_set, batch_size=bs, shuffle=True, drop_last=True, num_workers=0)
    labs = [extract_patients(base, everyone) for _ in range(2)]
    unl = extract_patients(DataLoader(MedicalImageDataset(mode="train", **kw), batch_size=bs, shuffle=True, drop_last=True, num_workers=0),
                           everyone)
    val = DataLoader(val_set, batch_sampler=PatientSampler(val_set, REGEX, shuffle=False, quite=True))
    return to_cached_loaders(labs, unl, val, device=device)


@needs_acdc
@pytest.mark.parametrize("arch", ["enet", "unet"])
def test_acdc_dsc_curve_matches_a_reference_that_learned(arch, tmp_path):
    """BASELINE.json: "DSC within 0.2 of the reference on ACDC at equal steps", against a reference run that actually segments
    the heart.  tests/golden/g10_acdc_<arch>.npz (tools/capture_golden.py::g10_acdc_dsc) is the UNMODIFIED reference CoTrainer,
    2 x Enet (5 epochs of 500 steps, bs 4 + 4) or 2 x UNet (the metric's network; 8 epochs of 250 steps, bs 2 + 2: the reference needs five of
    them to pass 0.5 and ends at 0.53 / 0.59), all five
    vendored training patients labeled for both models, CE + JSD, validated on patient 006 after every epoch.  The HIP bf16
    trainer runs the same epochs from the same initial weights on the same batches (digests compared step by step) and must
      * end within 0.2 of the reference's foreground DSC, per model, 3-D and 2-D,
      * reach a foreground DSC >= 0.3 itself (a net that predicts background scores ~0),
    and the fixture itself must show the reference at >= 0.5 on its best model (otherwise the comparison has no teeth)."""
    import hashlib
    from dct_amd import ModelMode
    from dct_amd.loss import get_loss_fn
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    from dct_amd.trainer import cotraining_totalloss as mod
    path = os.path.join(GOLDEN, f"g10_acdc_{arch}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not captured yet (tools/capture_golden.py g10_{arch})")
    g = np.load(path, allow_pickle=False)
    E, n, bs, C = int(g["epochs"]), int(g["steps_per_epoch"]), int(g["bs"]), int(g["C"])
    ref3 = g["val_dice3d"][:, :, 1:, 0].mean(2)          # [epoch, model] foreground mean
    ref2 = g["val_dice2d"][:, :, 1:, 0].mean(2)
    assert ref3[-1].max() >= 0.5, ("the reference never learned: the fixture cannot discriminate", ref3)
    labs, unl, val = _loaders_all_labeled(DEV, bs)
    segs = []
    for s in g["net_seeds"]:
        torch.manual_seed(int(s))
        sd = oracle.build_net(arch, C).state_dict()
        seg = Segmentator({"name": arch, "num_classes": C, "compute_dtype": torch.bfloat16}, {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4},
                          {"name": "StepLR", "step_size": 90, "gamma": 0.1})
        seg.torchnet.load_state_dict(sd)
        segs.append(seg)
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    tr = CoTrainer(segs, labs, unl, val, crit, max_epoch=E, save_dir=str(tmp_path), device=DEV, axises=[1, 2, 3],
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=n)
    names = []
    orig_iter = mod.iterator_

    # The reference fetches a loader's batch and runs that model's forward pass before it touches the next loader
    # (cotraining_totalloss.py:207-222), and its UNet's two nn.Dropout layers draw from the SAME global CPU generator the
    # shuffling samplers take their permutation seeds from (whenever a loader is exhausted and re-opened, twice per epoch
    # here).  So the reference's batch order depends on how many dropout calls its forward passes made.  The HIP nets draw
    # their masks on the device; to be served the reference's batches the test replays that consumption: after every fetch,
    # the dropout calls of the forward passes the reference would have run on it (one pass for a labeled batch, one per model
    # for the unlabeled batch; activation shapes of network.py:165,210 at 256 x 256).  Enet has no dropout: nothing to replay.
    S = len(segs)
    drop_shapes = [(bs, 512, 25, 25), (bs, 1024, 9, 9)] if arch == "unet" else []
    opened = [0]

    class rec_iter(orig_iter):
        def __init__(self, loader):
            super().__init__(loader)
            self.forwards = 1 if opened[0] < S else S        # _train_loop opens the S labeled loaders, then the unlabeled one
            opened[0] += 1

        def __next__(self):
            b = super().__next__()
            if isinstance(b, (list, tuple)) and len(b) == 3:
                names.append(",".join(b[2]))
                for _ in range(self.forwards):
                    for shp in drop_shapes:
                        torch.nn.functional.dropout(torch.zeros(shp), 0.5, True)
            return b
    np.random.seed(1234)
    torch.manual_seed(1234)
    mine3, mine2 = [], []
    for e in range(E):
        mod.iterator_, opened[0] = rec_iter, 0
        try:
            tr._train_loop(labs, unl, epoch=e, mode=ModelMode.TRAIN, save=False, train_jsd=True, train_adv=False)
        finally:
            mod.iterator_ = orig_iter
        with torch.no_grad():
            v2, v3 = tr._eval_loop(val, epoch=e, mode=ModelMode.EVAL, save=False)
        mine3.append(v3[:, 1:, 0].mean(1).numpy())
        mine2.append(v2[:, 1:, 0].mean(1).numpy())
        print(f"{arch} epoch {e}: 3-D foreground DSC HIP {mine3[-1]} reference {ref3[e]}   2-D HIP {mine2[-1]} reference {ref2[e]}", flush=True)
    # the data path served the reference's batches, step by step
    rows = np.array(names).reshape(E * n, 3)
    assert rows[:60].tolist() == [[str(x) for x in r] for r in g["batch_names_head"]]
    digest = np.array([hashlib.md5("|".join(r).encode()).hexdigest()[:12] for r in rows])
    assert (digest == g["batch_digest"]).all()
    mine3, mine2 = np.array(mine3), np.array(mine2)
    for mine, ref, what in ((mine3, ref3, "3-D"), (mine2, ref2, "2-D")):
        assert np.all(np.abs(mine[-1] - ref[-1]) <= 0.2), (what, mine[-1], ref[-1])
        assert mine[-1].max() >= 0.3, (what, mine[-1], ref[-1])

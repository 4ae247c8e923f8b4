"""Data path against goldens captured from the reference's dataset classes
(tools/capture_golden.py::g8_data): ACDC partition logic on the full tree's file names, decoded batches / patient grouping /
patient extraction on the vendored subset, and the decoded-once cache + rank sharding that replace per-step PNG decoding."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import digest

SUB = os.path.join(GOLDEN, "acdc_subset")
needs_acdc = pytest.mark.skipif(not os.path.isdir(os.path.join(SUB, "train", "img")), reason="tests/golden/acdc_subset is not in this checkout")
REGEX = r"(patient\d+_\d+)_\d+"


def _config(root, ratio, overlap, nm):
    from dct_amd.dataset import segment_transform
    return {"Dataset": {"root_dir": root, "subfolders": ["img", "gt"], "transform": segment_transform((256, 256)),
                        "augment": "PILaugment", "pin_memory": False},
            "Lab_Dataloader": {"pin_memory": False, "batch_size": 4, "num_workers": 0, "shuffle": True, "drop_last": True,
                               "batch_sampler": ["PatientSampler", {"grp_regex": REGEX, "shuffle": False}]},
            "Unlab_Dataloader": {"pin_memory": False, "batch_size": 4, "num_workers": 0, "shuffle": True, "drop_last": True},
            "Lab_Partitions": {"num_models": nm, "partition_sets": ratio, "partition_overlap": overlap}}


def test_acdc_partitions_match_reference(golden, tmp_path):
    """get_ACDC_split_dataloders on a tree of EMPTY files named like the whole ACDC-all set (names are all the partition logic
    reads): labeled file lists per model, unlabeled list, validation patient batches, numpy RNG consumption."""
    from dct_amd.dataset import get_ACDC_split_dataloders
    g = golden("g8_data")
    for mode, names in (("train", g["full_train_names"]), ("val", g["full_val_names"])):
        for sub in ("img", "gt"):
            d = tmp_path / mode / sub
            d.mkdir(parents=True)
            for n in names:
                (d / str(n)).touch()
    for tag in ("a", "b"):
        ratio, overlap, nm = g[f"{tag}_cfg"]
        np.random.seed(1234)
        labs, unl, val = get_ACDC_split_dataloders(_config(str(tmp_path), float(ratio), float(overlap), int(nm)))
        assert len(labs) == int(nm)
        for i, l in enumerate(labs):
            assert [os.path.basename(f) for f in l.dataset.filenames["img"]] == [str(x) for x in g[f"{tag}_lab{i}_names"]]
            assert l.batch_size == 4 and l.drop_last
        assert [os.path.basename(f) for f in unl.dataset.filenames["img"]] == [str(x) for x in g[f"{tag}_unl_names"]]
        assert sorted(len(b) for b in val.batch_sampler) == list(g[f"{tag}_val_batches"])
        assert np.random.randint(1 << 30) == int(g[f"{tag}_rng_after"])


def _subset(mode="train", pin=False):
    from dct_amd.dataset import MedicalImageDataset, segment_transform
    return MedicalImageDataset(root_dir=SUB, mode=mode, subfolders=["img", "gt"], transform=segment_transform((256, 256)),
                               augment="PILaugment", pin_memory=pin, quite=True)


@needs_acdc
def test_dataset_batches_match_reference(golden):
    from torch.utils.data import DataLoader
    from dct_amd.dataset import PatientSampler, extract_patients
    g = golden("g8_data")
    ds = _subset()
    assert len(ds) == 100
    torch.manual_seed(77)
    dl = DataLoader(ds, batch_size=4, shuffle=True, drop_last=True, num_workers=0)
    for k, ((img, gt), meta, fn) in enumerate(dl):
        assert list(fn) == [str(x) for x in g["sub_epoch_names"][k]]
        np.testing.assert_allclose(np.concatenate([digest(img), digest(gt.float())]), g["sub_epoch_digest"][k], rtol=1e-12)
        if k == 0:
            assert img.dtype == torch.float32 and gt.dtype == torch.int64 and img.shape == (4, 1, 256, 256)
            assert np.array_equal(img.numpy(), g["sub_first_img"]) and np.array_equal(gt.numpy(), g["sub_first_gt"])
    assert k + 1 == len(g["sub_epoch_names"])
    dv = _subset("val")
    groups = sorted(sorted(os.path.basename(dv.filenames["img"][i]) for i in b) for b in PatientSampler(dv, REGEX, quite=True))
    assert [",".join(x) for x in groups] == [str(x) for x in g["sub_val_groups"]]
    ext = extract_patients(DataLoader(ds, batch_size=2), ["2", "4"])
    assert [os.path.basename(f) for f in ext.dataset.filenames["img"]] == [str(x) for x in g["sub_extract_names"]]
    # pinned (pre-opened PIL images) datasets stay consistent after a patient extraction
    ext2 = extract_patients(DataLoader(_subset(pin=True), batch_size=2), ["2", "4"])
    a, b = ext.dataset[3], ext2.dataset[3]
    assert a[2] == b[2] and torch.equal(a[0][0], b[0][0]) and torch.equal(a[0][1], b[0][1])


@needs_acdc
def test_cached_loader_equals_dataloader_and_shards_over_ranks(golden):
    """DeviceSliceCache + CachedLoader: the same batches in the same order as DataLoader(shuffle, drop_last) for the same torch
    seed -- with one PNG decode per slice per run -- and disjoint equal shards for data-parallel ranks."""
    from dct_amd.dataset import CachedLoader, DeviceSliceCache, PatientSampler
    g = golden("g8_data")
    ds = _subset()
    cache = DeviceSliceCache(ds, "cpu")
    assert cache.img.dtype == torch.uint8 and cache.img.shape == (100, 1, 256, 256)
    torch.manual_seed(77)
    ld = CachedLoader(cache, 4, shuffle=True, drop_last=True, dataset=ds)
    assert len(ld) == 25 and ld.batch_size == 4
    for k, ((img, gt), meta, fn) in enumerate(ld):
        assert list(fn) == [str(x) for x in g["sub_epoch_names"][k]]
        np.testing.assert_allclose(np.concatenate([digest(img), digest(gt.float())]), g["sub_epoch_digest"][k], rtol=1e-12)
    assert k == 24
    # two ranks: same epoch permutation (same seed), every second batch each, nothing shared, equal counts
    seen = []
    for r in range(2):
        torch.manual_seed(5)
        names = [n for (_, _, fn) in CachedLoader(cache, 4, True, True, rank=r, world=2) for n in fn]
        assert len(names) == 12 * 4
        seen.append(set(names))
    assert not (seen[0] & seen[1])
    # validation: patient batches through the cache
    dv = _subset("val")
    cv = DeviceSliceCache(dv, "cpu")
    lv = CachedLoader(cv, batch_sampler=PatientSampler(dv, REGEX, quite=True), dataset=dv)
    sizes = [b[0][0].shape[0] for b in lv]
    assert sum(sizes) == len(dv) and len(sizes) == len(lv) == len(g["sub_val_groups"])
    lv.dataset.set_mode("eval")


@needs_acdc
def test_segment_transform_resizes_like_pil():
    """Resize(size) + ToTensor / NEAREST + ToLabel by their documented PIL semantics (torchvision is not in this image)."""
    from PIL import Image
    from dct_amd.dataset import segment_transform
    img = Image.open(os.path.join(SUB, "train", "img", "patient001_01_0_4.png"))
    gt = Image.open(os.path.join(SUB, "train", "gt", "patient001_01_0_4.png"))
    t = segment_transform((128, 192))
    a, b = t["img"](img), t["gt"](gt)
    assert a.shape == (1, 128, 192) and a.dtype == torch.float32 and 0.0 <= a.min() and a.max() <= 1.0
    assert b.shape == (1, 128, 192) and b.dtype == torch.int64 and set(b.unique().tolist()) <= {0, 1, 2, 3}
    ref = torch.from_numpy(np.array(img.resize((192, 128), Image.BILINEAR))).float() / 255
    assert torch.equal(a[0], ref)
    same = segment_transform((256, 256))
    assert torch.equal(same["img"](img)[0], torch.from_numpy(np.array(img)).float() / 255)


def test_gm_challenge_split_matches_reference(golden, tmp_path):
    """get_GMC_split_dataloders (dataset/GM_helper.py:34-101; BASELINE configs[3]'s data set) on a tree of empty files named like
    the whole GM_Challenge set: site-1 training acquisitions partitioned over the models with an overlap, sites 3 + 4 as
    validation, the unlabeled folder, numpy RNG consumption."""
    from dct_amd.dataset import get_GMC_split_dataloders, segment_transform
    g = golden("g8_data")
    for mode, names in (("train", g["gm_train_names"]), ("unlabeled", g["gm_unl_names"])):
        for sub in ("img", "gt"):
            d = tmp_path / mode / sub
            d.mkdir(parents=True)
            for n in names:
                (d / str(n)).touch()
    for tag in ("gma", "gmb"):
        overlap, nm = g[f"{tag}_cfg"]
        config = {"Dataset": {"root_dir": str(tmp_path), "subfolders": ["img", "gt"], "transform": segment_transform((200, 200)),
                              "augment": "PILaugment", "pin_memory": False},
                  "Unlab_Dataloader": {"pin_memory": False, "batch_size": 4, "num_workers": 0, "shuffle": True, "drop_last": True},
                  "Lab_Partitions": {"num_models": int(nm), "partition_overlap": float(overlap)}}
        np.random.seed(1234)
        labs, unl, val = get_GMC_split_dataloders(config)
        assert len(labs) == int(nm)
        for i, l in enumerate(labs):
            assert [os.path.basename(f) for f in l.dataset.filenames["img"]] == [str(x) for x in g[f"{tag}_lab{i}_names"]]
        assert len(unl.dataset) == int(g[f"{tag}_unl_n"])
        assert [os.path.basename(f) for f in val.dataset.filenames["img"]] == [str(x) for x in g[f"{tag}_val_names"]]
        assert np.random.randint(1 << 30) == int(g[f"{tag}_rng_after"])

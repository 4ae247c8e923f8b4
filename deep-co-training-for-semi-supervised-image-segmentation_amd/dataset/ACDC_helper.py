"""ACDC split helpers (reference: generalframework/dataset/ACDC_helper.py:27-141): ``PatientSampler`` (validation batches =
all slices of one patient scan), ``get_ACDC_dataloaders``, ``get_ACDC_split_dataloders`` (labeled partitions per model with a
configurable overlap, the unlabeled remainder, the validation loader) and ``extract_patients``.  Same config keys, same numpy
RNG consumption (``np.random.permutation`` then ``np.random.choice``), same file filtering; ``root_dir`` is taken from the
config instead of being forced to the reference checkout (:72)."""
from __future__ import annotations

import random
import re
from collections import defaultdict
from copy import deepcopy as dcopy
from pathlib import Path
from typing import Dict, List

import numpy as np
from torch.utils.data import DataLoader, Sampler

from .medicalDataLoader import CachedLoader, DeviceSliceCache, MedicalImageDataset


class PatientSampler(Sampler):
    """Batch sampler whose batches are the slice indices of one patient scan (reference :27-60).  ``grp_regex``'s first
    group names the scan a file stem belongs to; scans are served in first-seen file order (the reference iterates a
    ``set``, whose order is hash-dependent), each with its slice indices in file order; ``shuffle`` permutes the scans with
    ``random.sample`` -- one draw per iteration, as there."""

    def __init__(self, dataset: MedicalImageDataset, grp_regex, shuffle=False, quite=False) -> None:
        names: List[str] = dataset.filenames[dataset.subfolders[0]]
        self.grp_regex = grp_regex
        self.shuffle: bool = bool(shuffle)
        scan_of = re.compile(grp_regex)
        by_scan: Dict[str, List[int]] = defaultdict(list)        # insertion-ordered: scans in first-seen order
        for index, name in enumerate(names):
            by_scan[scan_of.match(Path(name).stem).group(1)].append(index)
        if not len(by_scan) < len(names):
            raise AssertionError("PatientSampler: the regex does not group the slices into scans")
        if not quite:
            print(f"Found {len(by_scan)} unique patients out of {len(names)} images")
        self.idx_map: Dict[str, List[int]] = dict(by_scan)

    def __len__(self):
        return len(self.idx_map)

    def __iter__(self):
        batches = list(self.idx_map.values())
        return iter(random.sample(batches, len(batches)) if self.shuffle else batches)


def get_ACDC_dataloaders(dataset_dict: dict, dataloader_dict: dict, quite=False, mode1='train', mode2='val'):
    dataset_dict = dict(dataset_dict)
    train_set = MedicalImageDataset(mode=mode1, quite=quite, **dataset_dict)
    val_set = MedicalImageDataset(mode=mode2, quite=quite, **dataset_dict)
    train_loader = DataLoader(train_set, **{**dataloader_dict, **{'batch_sampler': None}})
    if dataloader_dict.get('batch_sampler') is not None:
        name, kw = dataloader_dict.get('batch_sampler')
        assert name == 'PatientSampler', name
        val_sampler = PatientSampler(dataset=val_set, quite=quite, **kw)
        val_loader = DataLoader(val_set, batch_sampler=val_sampler)
    else:
        val_loader = DataLoader(val_set, **{**dataloader_dict, **{'shuffle': False, 'batch_size': 1}})
    return {'train': train_loader, 'val': val_loader}


def create_partitions(config):
    """-> (one array of labeled patient ids per model, list of unlabeled patient ids) (reference :86-103).

    ACDC's 100 training patients are cut at ``partition_sets``: ids 1..100*ratio are labeled, the rest unlabeled.  Every model
    sees the ``partition_overlap`` share of the labeled ids; the remaining ids are dealt to the models in equal consecutive
    runs (a remainder that does not divide is dropped, as in the reference).  The numpy global RNG is consumed in exactly this
    order -- it decides which patient lands where, and tests/golden/g8_data.npz pins it:
        1. ``np.random.permutation`` over the labeled ids,
        2. ``np.random.choice(..., replace=False)`` of the shared ids out of that permutation."""
    spec = config['Lab_Partitions']
    first_unlabeled = int(100 * spec['partition_sets'] + 1)
    labeled_ids = range(1, first_unlabeled)
    shuffled = np.random.permutation(labeled_ids)                                                     # RNG draw 1
    n_shared = int(float(spec['partition_overlap']) * len(labeled_ids))
    shared = np.random.choice(shuffled, size=n_shared, replace=False)                                 # RNG draw 2
    private = [pid for pid in shuffled if pid not in shared]       # keeps the permutation's order
    n_models = int(spec['num_models'])
    run = len(private) // n_models
    partitions = [np.hstack((shared, np.array(private[m * run:(m + 1) * run]))) for m in range(n_models)]
    return partitions, list(range(first_unlabeled, 101))


def get_ACDC_split_dataloders(config, quite=True):
    dataloders = get_ACDC_dataloaders(config['Dataset'], config['Lab_Dataloader'], quite=quite)
    lab_partitions, unlab = create_partitions(config)
    labeled_dataloaders = [extract_patients(dataloders['train'], [str(int(x)) for x in idx_lst]) for idx_lst in lab_partitions]
    unlab_dataloader = get_ACDC_dataloaders(config['Dataset'], config['Unlab_Dataloader'], quite=True)['train']
    unlab_dataloader = extract_patients(unlab_dataloader, [str(x) for x in unlab])
    return labeled_dataloaders, unlab_dataloader, dataloders['val']


def extract_patients(dataloader: DataLoader, patient_ids: List[str]):
    assert isinstance(patient_ids, list)
    patterns = re.compile('|'.join('patient%.3d' % int(d) for d in patient_ids))
    files = dcopy(dataloader.dataset.filenames)
    files = {k: sorted(s for s in file if re.search(patterns, s)) for k, file in files.items()}
    new_dataloader = dcopy(dataloader)
    if new_dataloader.dataset.pin_memory:       # keep the decoded images in step with the filtered names
        keep = {k: [i for i, s in enumerate(dataloader.dataset.filenames[k]) if re.search(patterns, s)] for k in files}
        new_dataloader.dataset.imgs = {k: [dataloader.dataset.imgs[k][i] for i in keep[k]] for k in files}
    else:
        new_dataloader.dataset.imgs = files
    new_dataloader.dataset.filenames = files
    return new_dataloader


def to_cached_loaders(labeled_dataloaders, unlab_dataloader, val_dataloader, device="cpu", rank=0, world=1):
    """The three kinds of loaders of a co-training run, re-served from decoded uint8 caches (one decode per slice per run
    instead of one per step): labeled / unlabeled loaders keep their batch size, shuffling and drop_last and are sharded over
    ``world`` ranks; the validation loader keeps its patient batches (every rank validates everything)."""
    def conv(dl, shard):
        cache = DeviceSliceCache(dl.dataset, device)
        if dl.batch_sampler is not None and isinstance(dl.batch_sampler, PatientSampler):
            return CachedLoader(cache, batch_sampler=dl.batch_sampler, dataset=dl.dataset)
        shuffle = dl.sampler.__class__.__name__ == "RandomSampler"
        return CachedLoader(cache, dl.batch_size, shuffle, dl.drop_last, rank=rank if shard else 0, world=world if shard else 1,
                            dataset=dl.dataset)
    return [conv(d, True) for d in labeled_dataloaders], conv(unlab_dataloader, True), conv(val_dataloader, False)

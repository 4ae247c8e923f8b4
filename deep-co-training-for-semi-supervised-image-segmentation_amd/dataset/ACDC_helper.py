"""ACDC split helpers (reference: generalframework/dataset/ACDC_helper.py:27-141): ``PatientSampler`` (validation batches =
all slices of one patient scan), ``get_ACDC_dataloaders``, ``get_ACDC_split_dataloders`` (labeled partitions per model with a
configurable overlap, the unlabeled remainder, the validation loader) and ``extract_patients``.  Same config keys, same numpy
RNG consumption (``np.random.permutation`` then ``np.random.choice``), same file filtering; ``root_dir`` is taken from the
config instead of being forced to the reference checkout (:72)."""
from __future__ import annotations

import random
import re
from copy import deepcopy as dcopy
from itertools import repeat
from pathlib import Path
from typing import Callable, Dict, List

import numpy as np
from torch.utils.data import DataLoader, Sampler

from .medicalDataLoader import CachedLoader, DeviceSliceCache, MedicalImageDataset


class PatientSampler(Sampler):
    def __init__(self, dataset: MedicalImageDataset, grp_regex, shuffle=False, quite=False) -> None:
        filenames: List[str] = dataset.filenames[dataset.subfolders[0]]
        self.grp_regex = grp_regex
        self.shuffle: bool = shuffle
        self.shuffle_fn: Callable = (lambda x: random.sample(x, len(x))) if self.shuffle else (lambda x: x)
        grouping_regex = re.compile(self.grp_regex)
        stems = [Path(filename).stem for filename in filenames]
        patients = [grouping_regex.match(s).group(1) for s in stems]
        unique_patients = list(dict.fromkeys(patients))     # first-seen order (the reference's set() order is hash-dependent)
        assert len(unique_patients) < len(filenames)
        if not quite:
            print(f"Found {len(unique_patients)} unique patients out of {len(filenames)} images")
        self.idx_map: Dict[str, List[int]] = dict(zip(unique_patients, repeat(None)))
        for i, patient in enumerate(patients):
            if not self.idx_map[patient]:
                self.idx_map[patient] = []
            self.idx_map[patient] += [i]
        assert sum(len(self.idx_map[k]) for k in unique_patients) == len(filenames)

    def __len__(self):
        return len(self.idx_map.keys())

    def __iter__(self):
        return iter(self.shuffle_fn(list(self.idx_map.values())))


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
    """Patient ids of the labeled partitions (one per model) and of the unlabeled set (:86-103)."""
    partition_ratio = config['Lab_Partitions']['partition_sets']
    lab_ids = [1, int(100 * partition_ratio + 1)]
    unlab_ids = [int(100 * partition_ratio + 1), 101]
    partition_overlap = config['Lab_Partitions']['partition_overlap']
    rd_idx = np.random.permutation(range(*lab_ids))
    overlap_idx = np.random.choice(rd_idx, size=int(float(partition_overlap) * len(range(*lab_ids))), replace=False)
    exclusive_idx = [x for x in rd_idx if x not in overlap_idx]
    n_splits = int(config['Lab_Partitions']['num_models'])
    exclusive_samples = int(len(exclusive_idx) / n_splits)
    excl_indx = [exclusive_idx[i * exclusive_samples: (i + 1) * exclusive_samples] for i in range(n_splits)]
    lab_partitions = [np.hstack((overlap_idx, np.array(excl_indx[idx]))) for idx in range(n_splits)]
    return lab_partitions, list(range(*unlab_ids))


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

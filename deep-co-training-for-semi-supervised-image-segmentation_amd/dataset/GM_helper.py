"""Spinal-cord grey-matter (GM challenge, 200 x 200, 2 classes: BASELINE configs[3]) split helpers (reference:
generalframework/dataset/GM_helper.py:13-101): ``train`` / ``unlabeled`` folders, validation = sites 3 and 4 of the labeled
folder, training = site 1, labeled partitions per model over the ``siteN-scNN`` acquisitions with a configurable overlap.
Same config keys, same numpy RNG consumption (one ``np.random.choice``), same file filtering."""
from __future__ import annotations

import re
from copy import deepcopy as dcopy
from functools import reduce
from pathlib import Path
from typing import Dict, List, Union

import numpy as np
from torch.utils.data import DataLoader

from .medicalDataLoader import MedicalImageDataset
from .augment import segment_transform, PILaugment  # noqa: F401  (config strings are eval'ed, as the reference does: :24)


def get_GM_dataloaders(dataset_dict: dict, dataloader_dict: dict, quite=False, mode1='train', mode2='unlabeled') -> Dict[str, DataLoader]:
    dataset_dict = {k: eval(v) if isinstance(v, str) and k not in ('root_dir', 'augment') else v for k, v in dataset_dict.items()}
    dataloader_dict = {k: eval(v) if isinstance(v, str) else v for k, v in dataloader_dict.items()}
    train_set = MedicalImageDataset(mode=mode1, quite=quite, **dataset_dict)
    unlabeled_set = MedicalImageDataset(mode=mode2, quite=quite, **dataset_dict)
    train_loader = DataLoader(train_set, **{**dataloader_dict, **{'batch_sampler': None}})
    unl_loader = DataLoader(unlabeled_set, **{**dataloader_dict, **{'batch_sampler': None}})
    return {'train': train_loader, 'unlabeled': unl_loader}


def extract_patients_gmc(dataloader: DataLoader, site_id: Union[List[int], None] = [1, 2], pattern=None) -> DataLoader:
    if pattern is not None:
        patterns = re.compile('|'.join(pattern))
    else:
        assert isinstance(site_id, list)
        patterns = re.compile('|'.join('site{:01d}'.format(int(site)) for site in site_id))
    files = dcopy(dataloader.dataset.filenames)
    keep = {k: [i for i, s in enumerate(v) if re.search(patterns, s)] for k, v in files.items()}
    new_dataloader = dcopy(dataloader)
    if new_dataloader.dataset.pin_memory:
        new_dataloader.dataset.imgs = {k: [dataloader.dataset.imgs[k][i] for i in keep[k]] for k in files}
    files = {k: sorted(files[k][i] for i in keep[k]) for k in files}
    if not new_dataloader.dataset.pin_memory:
        new_dataloader.dataset.imgs = files
    new_dataloader.dataset.filenames = files
    return new_dataloader


def get_GMC_split_dataloders(config, quite=True):
    def names(dl, key):
        return set(Path(x).name for x in dl.dataset.filenames[key])

    def no_overlap(d1, d2):
        assert names(d1, 'img') == names(d1, 'gt') and names(d2, 'img') == names(d2, 'gt')
        assert len(names(d1, 'img') & names(d2, 'img')) == 0
        return d2
    gm = get_GM_dataloaders(config['Dataset'], config['Unlab_Dataloader'], quite=True)
    lab_dataloader, unlabeled_dataloader = gm['train'], gm['unlabeled']
    val_dataloader = extract_patients_gmc(lab_dataloader, site_id=[3, 4])
    train_dataloader = extract_patients_gmc(lab_dataloader, site_id=[1])
    reduce(no_overlap, [lab_dataloader, unlabeled_dataloader, val_dataloader])
    u_pattern = re.compile(r'site\d-sc\d\d')
    u_samples = sorted(set(u_pattern.findall(x)[0] for x in train_dataloader.dataset.filenames['img']))
    num_model = int(config["Lab_Partitions"]["num_models"])
    common = list(np.random.choice(u_samples, int(len(u_samples) * float(config["Lab_Partitions"]["partition_overlap"])), replace=False))
    exclusive = [x for x in u_samples if x not in common]
    pattern_per_loader = [list(common) + exclusive[i::num_model] for i in range(num_model)]
    labeled = [extract_patients_gmc(dataloader=train_dataloader, pattern=pattern_per_loader[i]) for i in range(num_model)]
    return labeled, unlabeled_dataloader, val_dataloader

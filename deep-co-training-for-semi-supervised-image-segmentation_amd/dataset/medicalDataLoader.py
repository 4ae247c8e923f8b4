"""``MedicalImageDataset``: folder-of-PNG slices -> ``([img [1,H,W] float in [0,1], gt [1,H,W] int64], meta, stem)``
(reference: generalframework/dataset/medicalDataLoader.py:22-162), plus the MI355X-side cache that replaces per-step PNG
decoding: ``DeviceSliceCache`` decodes every slice ONCE into uint8 tensors (ACDC-all: 1674 x 256 x 256 x 2 = 220 MB, nothing
next to 288 GB of HBM) and ``CachedLoader`` serves ``[[img, gt], meta, names]`` batches from it with the DataLoader's own
shuffling arithmetic, optionally sharded over data-parallel ranks."""
from __future__ import annotations

import os
import re
from pathlib import Path
from typing import Dict, List, Optional, Union

import numpy as np
import torch
from PIL import Image, ImageOps
from torch import Tensor
from torch.utils.data import Dataset

from .. import ModelMode
from . import augment as augment_package
from .augment import segment_transform  # noqa: F401  (config strings are eval'ed here, as the reference does: :48-51)


class MedicalImageDataset(Dataset):
    dataset_modes = ['train', 'val', 'test', 'unlabeled']
    allow_extension = ['.jpg', '.png']

    def __init__(self, root_dir: str, mode: str, subfolders: List[str], transform=None, augment=None,
                 equalize: Union[List[str], str, None] = None, pin_memory=True, metainfo: str = None, quite=False) -> None:
        subfolders = [subfolders] if isinstance(subfolders, str) else subfolders
        assert isinstance(subfolders, list)
        assert len(subfolders) == len(set(subfolders)), f"subfolders must be unique, given {subfolders}."
        for s in subfolders:
            assert isinstance(s, str), f"subfolder element should be str, given {s}"
        self.name = '%s_dataset' % mode
        self.mode = mode
        self.root_dir = root_dir
        self.subfolders = subfolders
        if isinstance(transform, str):
            try:
                self.transform = getattr(augment_package, transform)
            except AttributeError:
                self.transform = eval(transform)        # e.g. "segment_transform((256,256))" from the YAML
        else:
            self.transform = transform
        self.pin_memory = pin_memory
        if not quite:
            print(f'->> Building {self.name}:\t')
        self.imgs, self.filenames = self.make_dataset(self.root_dir, self.mode, self.subfolders, self.pin_memory, quite=quite)
        self.augment = getattr(augment_package, augment) if isinstance(augment, str) else augment
        self.equalize = equalize
        self.training = ModelMode.TRAIN
        if metainfo:
            raise NotImplementedError("metainfo generators are not on the co-training path (config: metainfo unset)")
        self.metainfo_generator = None

    def __len__(self) -> int:
        return int(len(self.imgs[self.subfolders[0]]))

    def set_mode(self, mode) -> None:
        assert isinstance(mode, (str, ModelMode)), 'the type of mode should be str or ModelMode, given %s' % str(mode)
        self.training = ModelMode.from_str(mode) if isinstance(mode, str) else mode

    def load_pil(self, index) -> List[Image.Image]:
        if self.pin_memory:
            return [self.imgs[s][index] for s in self.subfolders]
        return [Image.open(self.imgs[s][index]) for s in self.subfolders]

    def __getitem__(self, index):
        img_list = self.load_pil(index)
        filename_list = [self.filenames[s][index] for s in self.subfolders]
        assert len(set(Path(x).stem for x in filename_list)) == 1, f"Check the filename list, given {filename_list}."
        filename = Path(filename_list[0]).stem
        if self.equalize:
            img_list = [ImageOps.equalize(img) if (b == self.equalize) or (b in self.equalize) else img
                        for b, img in zip(self.subfolders, img_list)]
        # medicalDataLoader.py:103: `if not self.augment and self.training == TRAIN` -- with an augmenter configured the
        # branch is never taken, so training runs on un-augmented slices; kept that way.
        img_T = [self.transform['img'](img) if b == 'img' else self.transform['gt'](img)
                 for b, img in zip(self.subfolders, img_list)]
        return img_T, [torch.Tensor([-1]), Tensor([1])], filename

    @classmethod
    def make_dataset(cls, root: str, mode: str, subfolders: List[str], pin_memory: bool, quite=False):
        def allowed(path: str) -> bool:
            try:
                return Path(path).suffixes[0] in cls.allow_extension
            except IndexError:
                return False
        assert mode in cls.dataset_modes
        for subfolder in subfolders:
            assert Path(os.path.join(root, mode, subfolder)).exists(), Path(os.path.join(root, mode, subfolder))
        items = [[x for x in os.listdir(os.path.join(root, mode, s)) if allowed(x)] for s in subfolders]
        assert len(set(len(i) for i in items)) == 1, [len(i) for i in items]
        imgs = {s: sorted(os.path.join(root, mode, s, x) for x in item) for s, item in zip(subfolders, items)}
        if not quite:
            for s in subfolders:
                print(f'found {len(imgs[s])} images in {s}\t')
        if pin_memory:
            return {k: [Image.open(i).convert('L') for i in v] for k, v in imgs.items()}, imgs
        return imgs, imgs


class DeviceSliceCache(object):
    """Every slice of a ``MedicalImageDataset`` decoded and transformed once: ``img`` uint8 ``[N,1,H,W]`` (the 8-bit grey
    levels; ``ToTensor`` is ``/ 255`` at batch time, bit-identical to transforming per item) and ``gt`` uint8 ``[N,1,H,W]``,
    on ``device`` (HBM) or in pinned host memory."""

    def __init__(self, dataset: MedicalImageDataset, device: Union[str, torch.device] = "cpu"):
        assert dataset.subfolders[:2] == ['img', 'gt'], dataset.subfolders
        self.names: List[str] = [Path(f).stem for f in dataset.filenames['img']]
        imgs, gts = [], []
        for i in range(len(dataset)):
            pil_img, pil_gt = dataset.load_pil(i)[:2]
            if dataset.equalize and ('img' == dataset.equalize or 'img' in dataset.equalize):
                pil_img = ImageOps.equalize(pil_img)
            t = dataset.transform['img'](pil_img)
            u8 = (t * 255.0).round().to(torch.uint8)
            assert torch.equal(u8.float() / 255.0, t), "the image transform must yield 8-bit grey levels / 255"
            imgs.append(u8)
            gts.append(dataset.transform['gt'](pil_gt).to(torch.uint8))
        device = torch.device(device)
        self.img = torch.stack(imgs).to(device)
        self.gt = torch.stack(gts).to(device)
        if device.type == "cpu" and torch.cuda.is_available():
            self.img, self.gt = self.img.pin_memory(), self.gt.pin_memory()

    def __len__(self):
        return len(self.names)

    def __deepcopy__(self, memo):
        return self         # immutable after construction: utils.iterator_ deep-copies its loader, the cache is shared

    def batch(self, idx: List[int]):
        i = torch.as_tensor(idx, dtype=torch.int64, device=self.img.device)
        return [[self.img[i].float().div_(255.0), self.gt[i].long()], [torch.full((len(idx), 1), -1.0), torch.ones(len(idx), 1)],
                [self.names[k] for k in idx]]


class CachedLoader(object):
    """Stand-in for ``DataLoader(dataset, batch_size, shuffle, drop_last)`` (or ``batch_sampler=PatientSampler``) over a
    ``DeviceSliceCache``: same batches, same order for the same torch seed (``RandomSampler`` draws one int64 seed from the
    global generator per epoch and permutes with it), no worker processes, no PNG decoding, no host->device copy per step.
    ``rank`` / ``world``: each data-parallel rank takes every ``world``-th batch of the epoch's order (equal counts)."""

    def __init__(self, cache: DeviceSliceCache, batch_size: int = 1, shuffle: bool = False, drop_last: bool = False,
                 batch_sampler=None, rank: int = 0, world: int = 1, dataset: Optional[MedicalImageDataset] = None):
        self.cache, self.batch_size, self.shuffle, self.drop_last = cache, batch_size, shuffle, drop_last
        self.batch_sampler, self.rank, self.world = batch_sampler, rank, world
        self.dataset = dataset if dataset is not None else _ModeHolder()

    def _batches(self) -> List[List[int]]:
        if self.batch_sampler is not None:
            out = [list(b) for b in self.batch_sampler]
        else:
            n = len(self.cache)
            if self.shuffle:
                seed = int(torch.empty((), dtype=torch.int64).random_().item())
                order = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).tolist()
            else:
                order = list(range(n))
            out = [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
            if self.drop_last and out and len(out[-1]) < self.batch_size:
                out.pop()
        if self.world > 1:
            usable = len(out) // self.world * self.world
            out = out[self.rank:usable:self.world]
        return out

    def __len__(self):
        if self.batch_sampler is not None:
            n = len(self.batch_sampler)
        else:
            n = len(self.cache) // self.batch_size if self.drop_last else -(-len(self.cache) // self.batch_size)
        return n // self.world if self.world > 1 else n

    def __iter__(self):
        return _CachedIter(self)


class _CachedIter(object):
    """Draws from the global torch generator exactly when a DataLoader iterator does: its `_base_seed` (worker seeding) when
    the iterator is CREATED, the sampler's permutation seed at the FIRST `next` -- so several loaders opened together
    (utils.iterator_ wraps each labeled loader, then the unlabeled one) see the same random stream as with DataLoaders."""

    def __init__(self, loader: "CachedLoader"):
        self.loader = loader
        torch.empty((), dtype=torch.int64).random_()
        self._it = None

    def __iter__(self):
        return self

    def __next__(self):
        if self._it is None:
            self._it = iter(self.loader._batches())
        return self.loader.cache.batch(next(self._it))


class _ModeHolder(object):
    training = ModelMode.EVAL

    def set_mode(self, mode):
        self.training = mode

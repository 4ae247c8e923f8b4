"""Slice transforms of the data path (reference: generalframework/dataset/augment.py:32-40,246-267,324-334).

``segment_transform(size)`` returns ``{'img': ..., 'gt': ...}`` exactly as the reference does, without torchvision:
``transforms.Resize(size)`` on a PIL image is ``img.resize((w, h), BILINEAR)`` (NEAREST for the mask) and
``transforms.ToTensor()`` on an 8-bit grey image is ``uint8 -> float32 / 255`` shaped ``[1, H, W]``; ``ToLabel`` is
``np.array(img)[None] -> int64``.  At a dataset's native size (ACDC 256 x 256) the resize is the identity.
``PILaugment`` (flip / mirror / rotate / crop with Python's ``random``) is restated too, although the reference never
applies it (the guard at medicalDataLoader.py:103 is ``if not self.augment and ...``)."""
from __future__ import annotations

import random

import numpy as np
import torch
from PIL import Image, ImageOps


class ToLabel(object):
    def __call__(self, img):
        return torch.from_numpy(np.array(img)[None, ...]).long()


class _ResizeToTensor(object):
    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        img = _resize(img, self.size, Image.BILINEAR)
        a = np.array(img, dtype=np.uint8)
        if a.ndim == 2:
            a = a[None, ...]
        else:
            a = a.transpose(2, 0, 1)
        return torch.from_numpy(a).float().div_(255.0)


class _ResizeToLabel(object):
    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        return ToLabel()(_resize(img, self.size, Image.NEAREST))


def _resize(img, size, resample):
    if isinstance(size, int):           # torchvision: the smaller edge is matched to `size`
        w, h = img.size
        if (w <= h and w == size) or (h <= w and h == size):
            return img
        if w < h:
            ow, oh = size, int(size * h / w)
        else:
            oh, ow = size, int(size * w / h)
        return img.resize((ow, oh), resample)
    h, w = size
    if img.size == (w, h):
        return img
    return img.resize((w, h), resample)


def segment_transform(size):
    return {'img': _ResizeToTensor(size), 'gt': _ResizeToLabel(size)}


def PILaugment(img_list):
    if random.random() > 0.5:
        img_list = [ImageOps.flip(img) for img in img_list]
    if random.random() > 0.5:
        img_list = [ImageOps.mirror(img) for img in img_list]
    if random.random() > 0.5:
        angle = random.random() * 90 - 45
        img_list = [img.rotate(angle, resample=Image.NEAREST) for img in img_list]
    if random.random() > 0.5:
        (w, h) = img_list[0].size
        crop = random.uniform(0.85, 0.95)
        W, H = int(crop * w), int(crop * h)
        x_pos, y_pos = int(random.uniform(0, w - W)), int(random.uniform(0, h - H))
        img_list = [img.crop((x_pos, y_pos, x_pos + W, y_pos + H)) for img in img_list]
    return img_list

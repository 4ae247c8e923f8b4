"""Host-side helpers the step touches (reference: generalframework/utils/utils.py).
``iterator_`` (:254-275) is kept verbatim in behaviour because ``_FSGM_adv_training`` relies on
its ``__cache__`` ("re-use this step's batches"); the tensor predicates (:142-207) are
debug-only here because each of them synchronises with the host."""
from __future__ import annotations

import argparse
import collections.abc
import os
import random
from copy import deepcopy as dcopy
from functools import partial, reduce
from typing import Any, Callable, Iterable, List, TypeVar, Union

import numpy as np
import torch
from torch import Tensor

A = TypeVar("A")
B = TypeVar("B")

try:
    from tqdm import tqdm
    tqdm_ = partial(tqdm, ncols=125, leave=False,
                    bar_format='{l_bar}{bar}| {n_fmt}/{total_fmt} [' '{rate_fmt}{postfix}]')
except Exception:  # pragma: no cover
    tqdm_ = None


def map_(fn: Callable[[A], B], iter: Iterable[A]) -> List[B]:
    return list(map(fn, iter))


def pred2class(pred: Tensor) -> Tensor:
    assert pred.shape.__len__() == 4, pred.shape
    return pred.max(1)[1]


def simplex(t: Tensor, axis=1) -> bool:
    _sum = t.sum(axis).type(torch.float32)
    return bool(torch.allclose(_sum, torch.ones_like(_sum, dtype=torch.float32)))


def uniq(a: Tensor) -> set:
    return set(torch.unique(a.cpu()).numpy())


def sset(a: Tensor, sub: Iterable) -> bool:
    return uniq(a).issubset(sub)


def one_hot(t: Tensor, axis=1) -> bool:
    return simplex(t, axis) and sset(t, [0, 1])


def probs2class(probs: Tensor) -> Tensor:
    return probs.argmax(dim=1)


def class2one_hot(seg: Tensor, C: int) -> Tensor:
    if len(seg.shape) == 2:
        seg = seg.unsqueeze(dim=0)
    return torch.stack([seg == c for c in range(C)], dim=1).type(torch.int32)


def probs2one_hot(probs: Tensor) -> Tensor:
    return class2one_hot(probs2class(probs), probs.shape[1])


class iterator_(object):
    """Infinite iterator over a loader that remembers the last batch (utils.py:254-275)."""

    def __init__(self, dataloader) -> None:
        super().__init__()
        self.dataloader = dcopy(dataloader)
        self.iter_dataloader = iter(dataloader)
        self.cache = None

    def __next__(self):
        try:
            self.cache = self.iter_dataloader.__next__()
        except StopIteration:
            self.iter_dataloader = iter(self.dataloader)
            self.cache = self.iter_dataloader.__next__()
        return self.cache

    def __cache__(self):
        if self.cache is not None:
            return self.cache
        import warnings
        warnings.warn('No cache found, iterator forward')
        return self.__next__()


# ---- "a.b=c" CLI overrides merged into the YAML config (utils.py:280-351) ----------------------
def yaml_parser() -> dict:
    parser = argparse.ArgumentParser('Augment parser for yaml config')
    parser.add_argument('strings', nargs='*', type=str, default=[''])
    args = parser.parse_args()
    return _parser(args.strings)


def _parser(strings: List[str]):
    assert isinstance(strings, list)
    assert len(set(s.split('=')[0] for s in strings)) == len(strings), 'Augment doubly input.'
    args = [_parser_(s) for s in strings]
    return reduce(lambda x, y: dict_merge(x, y, True), args)


def _parser_(input_string: str):
    if len(input_string) == 0:
        return None
    assert input_string.find('=') > 0, "Input args should include '=' to include the value"
    keys, value = input_string.split('=')[0].replace(' ', ''), input_string.split('=')[1].replace(' ', '')
    for k in reversed(keys.split('.')):
        value = {k: value}
    return dict(value)


def dict_merge(dct: dict, merge_dct: dict, re=False):
    """Recursive merge of ``merge_dct`` into ``dct``; values take the type already present in ``dct``
    (bool/list through eval), as the reference does (utils.py:325-351)."""
    if merge_dct is None:
        return dct if re else None
    for k, v in merge_dct.items():
        if k in dct and isinstance(dct[k], dict) and isinstance(v, collections.abc.Mapping):
            dict_merge(dct[k], v)
        else:
            try:
                dct[k] = type(dct[k])(eval(v)) if type(dct[k]) in (bool, list) else type(dct[k])(v)
            except Exception:
                dct[k] = v
    if re:
        return dcopy(dct)


def fix_all_seed(seed):
    random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)

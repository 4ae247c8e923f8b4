"""Host-side helpers of the training step.

Only what the hot path touches lives here: the endless loader iterator ``_FSGM_adv_training`` re-reads
batches from (reference behaviour: generalframework/utils/utils.py:254-275), the nested-dict merge the
report dictionaries use, seeding, and a few tensor predicates that are debug-only on this path (each
synchronises with the host).  The reference's command-line override parser (utils.py:280-351) is NOT
part of the path: a training script keeps using the reference's own ``yaml_parser``."""
from __future__ import annotations

import ast
import copy
import os
import random
import warnings
from functools import partial
from typing import Any, Callable, Iterable, List, Mapping, MutableMapping, Optional, TypeVar

import numpy as np
import torch
from torch import Tensor

A = TypeVar("A")
B = TypeVar("B")

try:
    from tqdm import tqdm
    tqdm_ = partial(tqdm, ncols=125, leave=False, bar_format='{l_bar}{bar}| {n_fmt}/{total_fmt} [{rate_fmt}{postfix}]')
except Exception:  # pragma: no cover
    tqdm_ = None


def map_(fn: Callable[[A], B], items: Iterable[A]) -> List[B]:
    return [fn(x) for x in items]


# ---- tensor predicates / conversions (debug asserts and the meters' inputs) ----------------------
def simplex(t: Tensor, axis: int = 1) -> bool:
    """every slice along ``axis`` sums to one"""
    total = t.sum(axis, dtype=torch.float32)
    return bool(torch.allclose(total, torch.ones((), dtype=torch.float32, device=total.device).expand_as(total)))


def uniq(a: Tensor) -> set:
    return set(a.detach().unique().cpu().tolist())


def sset(a: Tensor, allowed: Iterable) -> bool:
    return uniq(a) <= set(allowed)


def one_hot(t: Tensor, axis: int = 1) -> bool:
    return sset(t, (0, 1)) and simplex(t, axis)


def probs2class(probs: Tensor) -> Tensor:
    return probs.argmax(dim=1)


def pred2class(pred: Tensor) -> Tensor:
    assert pred.dim() == 4, pred.shape
    return pred.argmax(dim=1)


def class2one_hot(seg: Tensor, C: int) -> Tensor:
    if seg.dim() == 2:
        seg = seg[None]
    classes = torch.arange(C, device=seg.device).view(1, C, *([1] * (seg.dim() - 1)))
    return (seg.unsqueeze(1) == classes).to(torch.int32)


def probs2one_hot(probs: Tensor) -> Tensor:
    return class2one_hot(probs2class(probs), probs.shape[1])


# ---- the endless batch source of the training loop ------------------------------------------------
class iterator_:
    """Endless iterator over a loader that remembers the batch it returned last.

    ``next`` restarts the loader when it runs out (from a private copy, so restarting never disturbs the
    caller's loader object); ``__cache__()`` hands the remembered batch back -- the adversarial block of a
    step re-uses the batches the supervised block has just drawn (cotraining_totalloss.py:371-392) -- and
    draws one, with a warning, when nothing has been drawn yet."""

    def __init__(self, dataloader) -> None:
        self._source = copy.deepcopy(dataloader)
        self._it = iter(dataloader)
        self.cache: Optional[Any] = None

    def __iter__(self):
        return self

    def __next__(self):
        batch = next(self._it, _EXHAUSTED)
        if batch is _EXHAUSTED:
            self._it = iter(self._source)
            batch = next(self._it)
        self.cache = batch
        return batch

    def __cache__(self):
        if self.cache is None:
            warnings.warn('No cache found, iterator forward')
            return self.__next__()
        return self.cache


_EXHAUSTED = object()


# ---- nested dictionaries --------------------------------------------------------------------------
def _coerce_like(old: Any, new: Any) -> Any:
    """``new`` in the type ``old`` has, as the reference's leaf rule does it (utils/utils.py:345-349): bool / list defaults read a
    string override as a Python literal first (here without `eval`), every other type is the plain constructor call on the value
    AS GIVEN -- so 'max_epoch=0.5' against an int default stays the string '0.5' (int('0.5') raises) instead of silently
    becoming 0 -- and whatever fails to convert is kept as given.  One deliberate difference: 'true' / 'false' in any case are
    read as the booleans (the reference keeps the string 'false', which is truthy)."""
    kind = type(old)
    if kind in (bool, list):
        if not isinstance(new, str):
            return new                                   # (the reference's eval() of a non-string raises: kept as given)
        if kind is bool and new.strip().lower() in ("true", "false"):
            return new.strip().lower() == "true"
        try:
            return kind(ast.literal_eval(new))
        except (ValueError, SyntaxError, TypeError):
            return new
    if old is None:
        return new
    try:
        return kind(new)
    except (TypeError, ValueError):
        return new


def dict_merge(dct: MutableMapping, merge_dct: Optional[Mapping], re: bool = False):
    """Merge ``merge_dct`` into ``dct`` level by level (in place); a leaf that replaces an existing one takes its type.
    ``re=True`` returns a deep copy of the merged dictionary (the in-place form returns None, as the reference's does)."""
    stack = [(dct, merge_dct)] if merge_dct else []
    while stack:
        dst, src = stack.pop()
        for key, value in src.items():
            if isinstance(value, Mapping) and isinstance(dst.get(key), MutableMapping):
                stack.append((dst[key], value))
            elif key in dst:
                dst[key] = _coerce_like(dst[key], value)
            else:
                dst[key] = value
    return copy.deepcopy(dct) if re else None


def fix_all_seed(seed: int) -> None:
    os.environ['PYTHONHASHSEED'] = str(seed)
    for seeder in (random.seed, np.random.seed, torch.manual_seed, torch.cuda.manual_seed_all):
        seeder(seed)

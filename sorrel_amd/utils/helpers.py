"""Host helpers with the semantics of ``sorrel/utils/helpers.py`` (shift, one_hot_encode, set_seed)."""
from __future__ import annotations

import random
from typing import Any, Sequence

import numpy as np


def set_seed(seed: int) -> None:
    """Seed Python, numpy and torch (``sorrel/utils/helpers.py:22-32``).  The batched
    engine itself is keyed by its own ``seed`` (counter RNG), not by these streams."""
    import torch

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def random_seed() -> int:
    seed = random.randint(0, 10000)
    set_seed(seed)
    return seed


def shift(array: np.ndarray, shift: Sequence | np.ndarray, cval: Any = np.nan) -> np.ndarray:
    """Copy of ``array`` moved by ``shift`` along each axis, vacated cells = ``cval``
    (``sorrel/utils/helpers.py:48-77``).  On the device this is the window addressing
    of the observe kernel; this host form is kept for user code that imports it."""
    offs = [int(o) for o in np.atleast_1d(shift)]
    if len(offs) != array.ndim:
        raise AssertionError("shift needs one offset per array axis")
    out = np.full_like(array, cval)
    src = tuple(slice(max(-o, 0), array.shape[a] - max(o, 0)) for a, o in enumerate(offs))
    dst = tuple(slice(max(o, 0), array.shape[a] - max(-o, 0)) for a, o in enumerate(offs))
    if all(s.stop > s.start for s in src):
        out[dst] = array[src]
    return out


def one_hot_encode(value: int, num_classes: int) -> np.ndarray:
    """float64 one-hot vector (``sorrel/utils/helpers.py:130-150``)."""
    assert value <= num_classes - 1, f"The maximum value of `value` is {num_classes - 1}."
    v = np.zeros(num_classes)
    v[value] = 1
    return v


def nearest_2_power(n: int) -> int:
    """Smallest power of two >= n (``sorrel/utils/helpers.py:80-108``)."""
    return n if n and not (n & (n - 1)) else 1 << int(n).bit_length()


def clip(n, minimum, maximum):
    return minimum if n < minimum else maximum if n > maximum else n

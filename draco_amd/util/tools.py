"""Helpers of ``draco/util/tools.py`` that the path uses."""

from __future__ import annotations

import numpy as np


def invert_no_zero(x):
    """``1/x`` where ``x != 0`` else 0 (caput.algorithms.invert_no_zero [3P], ``tools.py:12``)."""
    x = np.asarray(x)
    if x.ndim == 0:
        return np.float64(0.0 if x == 0 else 1.0 / float(x))
    dt = x.dtype if x.dtype.kind in "fc" else np.float64
    out = np.zeros(x.shape, dtype=dt)
    nz = x != 0
    out[nz] = 1.0 / x[nz]
    return out


def find_keys(key_list, keys, require_match=False):
    """Indices of ``keys`` in ``key_list`` by exact match (``tools.py:95-127``)."""
    try:
        lookup = {tuple(k): i for i, k in enumerate(key_list)}
        index = [lookup.get(tuple(k)) for k in keys]
    except TypeError:
        lookup = {k: i for i, k in enumerate(key_list)}
        index = [lookup.get(k) for k in keys]
    if require_match and any(i is None for i in index):
        raise ValueError("Could not find all of the keys.")
    return index

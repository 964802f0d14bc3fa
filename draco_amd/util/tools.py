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


def _as_key(k):
    """A hashable form of one key: scalars as they are, sequences (rows of a 2-D key list) as tuples."""
    try:
        hash(k)
        return k
    except TypeError:
        return tuple(k)


def find_keys(key_list, keys, require_match=False):
    """Position of every entry of ``keys`` inside ``key_list`` (exact equality; ``tools.py:95-127``).

    Entries that do not occur give ``None``, or -- with ``require_match`` -- the reference's
    ``ValueError("Could not find all of the keys.")``, which ``BaseMapMaker.process`` relies on for data frequencies
    the beam transfers lack (``mapmaker.py:59``).  When a key occurs twice the last occurrence wins, as in a dict.
    """
    position = {}
    for i, k in enumerate(key_list):
        position[_as_key(k)] = i
    found = [position.get(_as_key(k)) for k in keys]
    if require_match and None in found:
        raise ValueError("Could not find all of the keys.")
    return found

"""Oracle: ExpandProducts.  TEST INFRASTRUCTURE ONLY.

Restates ``ExpandProducts.process`` (reference ``draco/synthesis/stream.py:193-246``) on plain arrays.
Pinned by ``tests/golden/stream_expand.npz`` (outputs of the reference class).
"""

from __future__ import annotations

import numpy as np


def expand_products(vis, feedmap, feedconj, ninput):
    """``vis [nfreq, nstack, nra]`` -> ``(vis [nfreq, nprod, nra], weight)`` over the full triangle of ``ninput``."""
    prod = [(fi, fj) for fi in range(ninput) for fj in range(fi, ninput)]
    out = np.zeros((vis.shape[0], len(prod), vis.shape[2]), dtype=vis.dtype)
    w = np.zeros(out.shape, dtype=np.float32)
    for pi, (fi, fj) in enumerate(prod):
        u = feedmap[fi, fj]
        if u < 0:
            continue
        out[:, pi] = vis[:, u].conj() if feedconj[fi, fj] else vis[:, u]
        w[:, pi] = 1.0
    return out, w

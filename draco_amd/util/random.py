"""Random draws used by the noise tasks (host side, NumPy ``Generator``).

Same draw order as ``draco/util/random.py`` so that a seeded ``np.random.Generator`` gives the
reference's stream: :func:`complex_normal` (``random.py:7-83``),
:func:`standard_complex_wishart` (``:106-137``, Bartlett factor), :func:`complex_wishart`
(``:140-166``).  Input generation only -- not a GPU target.
"""

from __future__ import annotations

import numpy as np


def complex_normal(loc=0.0, scale=1.0, size=None, dtype=None, rng=None, out=None):
    """Complex normal variates with total standard deviation ``scale`` (``random.py:7-83``)."""
    if size is None:
        size = (1,) if out is None else out.shape
    elif out is not None and tuple(out.shape) != tuple(size):
        raise ValueError(f"Shape of output array ({out.shape}) != size argument ({size}")
    if dtype is None:
        dtype = np.complex128 if out is None else out.dtype.type
    elif out is not None and out.dtype.type != dtype:
        raise ValueError(f"Dtype of output array ({out.dtype.type}) != dtype argument ({dtype}")
    real = {np.complex64: np.float32, np.complex128: np.float64}.get(dtype)
    if real is None:
        raise ValueError(f"Only dtype must be complex64 or complex128. Got dtype={dtype}.")
    if rng is None:
        rng = np.random.default_rng()
    if out is None:
        out = np.empty(size, dtype=dtype)
    # (re, im) interleaved draws straight into the output's real view
    rng.standard_normal((*size[:-1], 2 * size[-1]), dtype=real, out=out.view(real))
    out *= scale / 2**0.5
    if np.any(loc != 0.0):
        out += loc
    return out


def standard_complex_normal(shape, dtype=None, rng=None):
    return complex_normal(size=shape, dtype=dtype, rng=rng)


def standard_complex_wishart(m, n, rng=None):
    """Standard complex Wishart ``T T^H`` from the Bartlett factor ``T`` (``random.py:106-137``)."""
    if rng is None:
        rng = np.random.default_rng()
    nlow = m * (m - 1) // 2
    T = np.zeros((m, m), dtype=np.complex128)
    T[np.tril_indices(m, k=-1)] = (rng.standard_normal(nlow) + 1.0j * rng.standard_normal(nlow)) / 2**0.5
    for i in range(m):
        T[i, i] = rng.gamma(n - i) ** 0.5
    return T @ T.T.conj()


def complex_wishart(C, n, rng=None):
    """``L A L^H`` with ``C = L L^H`` and ``A`` standard Wishart (``random.py:140-166``)."""
    import scipy.linalg as la

    L = la.cholesky(C, lower=True)
    A = standard_complex_wishart(C.shape[0], n, rng=rng)
    return L @ (A @ L.T.conj())

"""Random draws behind the noise tasks of config 5 (host side, NumPy ``Generator``).

Input generation, not a GPU target.  What has to agree with
``draco/util/random.py`` is the *stream*: a ``Generator`` seeded alike must give the same numbers,
which fixes the order and the dtype of the underlying draws and nothing else --

* :func:`complex_normal`: ONE ``standard_normal`` call of the real dtype over the output viewed as
  interleaved (re, im) pairs (``random.py:7-83``);
* :func:`standard_complex_wishart`: the strict lower triangle of the Bartlett factor in row-major order,
  all real parts first and then all imaginary parts, then one ``gamma`` draw per diagonal entry from the
  top (``random.py:106-137``);
* :func:`complex_wishart`: colours a standard draw with the Cholesky factor of ``C`` (``random.py:140-166``).

``tests/golden/noise.npz`` (the reference functions run from source) enforces exactly that.
"""

from __future__ import annotations

import numpy as np

_REAL_OF = {np.dtype(np.complex64): np.float32, np.dtype(np.complex128): np.float64}


def _target(size, dtype, out):
    """Settle shape / dtype / destination of a draw from whichever of the three the caller gave."""
    if out is not None:
        if size is not None and tuple(np.shape(out)) != tuple(np.atleast_1d(size)):
            raise ValueError(f"`out` has shape {np.shape(out)} but size={size} was requested")
        if dtype is not None and np.dtype(dtype) != out.dtype:
            raise ValueError(f"`out` is {out.dtype} but dtype={np.dtype(dtype)} was requested")
        cdt = out.dtype
    else:
        cdt = np.dtype(np.complex128 if dtype is None else dtype)
    if cdt not in _REAL_OF:
        raise ValueError(f"complex draws exist for complex64 and complex128 only, not {cdt}")
    if out is None:
        shape = (1,) if size is None else tuple(int(n) for n in np.atleast_1d(size))
        out = np.empty(shape, dtype=cdt)
    return out, _REAL_OF[cdt]


def complex_normal(loc=0.0, scale=1.0, size=None, dtype=None, rng=None, out=None):
    """Circular complex Gaussian variates, mean ``loc`` and *total* standard deviation ``scale``.

    ``loc`` / ``scale`` broadcast against the result; ``out`` (a C-contiguous complex array) is filled in place
    when given.  Real and imaginary parts each carry ``scale**2 / 2`` of the variance.
    """
    out, real = _target(size, dtype, out)
    gen = np.random.default_rng() if rng is None else rng
    pairs = out.view(real)  # [..., 2 n]: re0 im0 re1 im1 ...
    gen.standard_normal(pairs.shape, dtype=real, out=pairs)
    # (a Python-float scale must stay a Python float: NumPy then multiplies in the output's own precision)
    out *= scale / 2**0.5
    if np.any(np.asarray(loc) != 0):
        out += loc
    return out


def standard_complex_normal(shape, dtype=None, rng=None):
    """Unit-variance, zero-mean :func:`complex_normal` of the given shape."""
    return complex_normal(size=shape, dtype=dtype, rng=rng)


def standard_complex_wishart(m, n, rng=None):
    """A draw from the standard complex Wishart distribution ``W_m(I, n)`` by Bartlett's decomposition.

    ``A = T T^H`` with ``T`` lower triangular, ``T_ii^2 ~ Gamma(n - i)`` and unit complex normals below the
    diagonal.
    """
    gen = np.random.default_rng() if rng is None else rng
    below = np.tril_indices(m, k=-1)
    count = below[0].size
    re = gen.standard_normal(count)
    im = gen.standard_normal(count)
    T = np.zeros((m, m), dtype=np.complex128)
    T[below] = (re + 1j * im) / 2**0.5
    T[np.diag_indices(m)] = [np.sqrt(gen.gamma(n - k)) for k in range(m)]
    return T @ T.conj().T


def complex_wishart(C, n, rng=None):
    """A draw from ``W(C, n)``: ``L A L^H`` for ``C = L L^H`` and ``A`` standard Wishart with ``n`` samples."""
    C = np.asarray(C)
    L = np.linalg.cholesky(C)
    A = standard_complex_wishart(C.shape[0], n, rng=rng)
    return L @ A @ L.conj().T

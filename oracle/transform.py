"""Oracle: sidereal stream <-> m-mode transform.  TEST INFRASTRUCTURE ONLY.

Restates ``draco/analysis/transform.py`` (reference), plain ndarray contracts:

* :func:`invert_no_zero`        caput.algorithms.invert_no_zero [3P] as used at
                                 ``transform.py:600-601,633,700``
* :func:`make_marray`           ``_make_marray``            ``transform.py:644-705``
* :func:`mmode_transform`       ``MModeTransform.process``  ``transform.py:573-641``
* :func:`unpack_marray`         ``_unpack_marray``          ``transform.py:820-851``
* :func:`make_ssarray`          ``_make_ssarray``           ``transform.py:814-817``
* :func:`mmode_inverse_transform`  ``MModeInverseTransform.process`` ``transform.py:733-792``

Pinned by ``tests/golden/transform_*.npz`` (outputs of the reference functions).
"""

from __future__ import annotations

import numpy as np


def invert_no_zero(x):
    """``1/x`` where ``x != 0`` else ``0`` (dtype preserving for floats)."""
    x = np.asarray(x)
    if x.ndim == 0:
        return 0.0 if x == 0 else 1.0 / float(x)
    dt = x.dtype if x.dtype.kind in "fc" else np.float64
    out = np.zeros(x.shape, dtype=dt)
    nz = x != 0
    out[nz] = (1.0 / x[nz]).astype(dt)
    return out


def mlimits(N: int, mmax: int) -> tuple[int, int]:
    """Largest filled +m and -m, ``transform.py:678-679``."""
    mlim = min(N // 2, mmax)
    mlim_neg = N // 2 - 1 + N % 2 if mmax >= N // 2 else mmax
    return mlim, mlim_neg


def make_marray(ts, mmodes=None, mmax=None, dtype=None):
    """FFT the last axis and pack +/-m; ``transform.py:644-705``.

    ``ts [..., N]`` complex -> ``mmodes [mmax+1, 2, ...]``.  The FFT runs in the
    precision of ``ts`` (complex64 for a SiderealStream) exactly like
    ``np.fft.fft`` at ``transform.py:689``; the result is multiplied by ``1/N``
    and written into ``mmodes`` (whatever its dtype), everything else untouched.
    """
    ts = np.asarray(ts)
    if dtype is None:
        dtype = np.complex64
    if mmodes is None and mmax is None:
        raise ValueError("One of `mmodes` or `mmax` must be set.")
    if mmodes is not None and mmax is not None:
        raise ValueError("If mmodes is set, mmax must be None.")
    if mmodes is not None and mmodes.shape[2:] != ts.shape[:-1]:
        raise ValueError(
            "ts and mmodes have incompatible shapes: "
            f"{mmodes.shape[2:]} != {ts.shape[:-1]}"
        )
    if mmodes is None:
        mmodes = np.zeros((mmax + 1, 2, *ts.shape[:-1]), dtype=dtype)
    if mmax is None:
        mmax = mmodes.shape[0] - 1

    N = ts.shape[-1]
    mlim, mlim_neg = mlimits(N, mmax)

    F = np.fft.fft(ts.reshape(-1, N), axis=-1).reshape(ts.shape)
    F = np.moveaxis(F, -1, 0)

    npos = mlim + 1
    nneg = mlim_neg + 1
    # caput's invert_no_zero [3P] hands back a NumPy float64 for an integer argument
    # (a "strong" type under NumPy>=2 promotion), so the reference's products at
    # transform.py:701,703 are formed in complex128 from the single-precision FFT
    # output; the stub used by gen_golden.py behaves the same way.
    norm = np.float64(invert_no_zero(N))
    mmodes[:npos, 0] = F[:npos] * norm
    mmodes[1:nneg, 1] = F[-1:-nneg:-1].conj() * norm
    return mmodes


def mmode_weight(weight):
    """``nra**2 * inz(sum_ra inz(w))``, ``transform.py:599-602`` -> ``[..., ]`` (ra axis reduced).

    Evaluated in the dtype of ``weight`` (float32 for a SiderealStream) like the
    reference, the caller stores it into a float64 dataset (``transform.py:627``).
    """
    weight = np.asarray(weight)
    nra = weight.shape[-1]
    return nra**2 * invert_no_zero(invert_no_zero(weight).sum(axis=-1))


def mmode_transform(vis, weight, mmax=None, remove_integration_window=False, vis_dtype=np.complex128, weight_dtype=np.float64):
    """``MModeTransform.process`` on plain arrays, ``transform.py:594-641``.

    ``vis [nfreq, nstack, nra]`` complex64, ``weight`` same shape float32 ->
    ``(mvis [mmax+1, 2, nfreq, nstack] complex128, mweight same float64)``.
    ``mmax=None`` means "no telescope given": ``nra // 2`` (``transform.py:604-607``).
    For a ``HybridVisStream`` (``transform.py:587``) ``vis`` is ``[pol, freq, ew, el, ra]`` and
    ``weight`` ``[pol, freq, ew, ra]``; the output datasets are complex64 / float32
    (``containers.py:1559-1574``): pass ``vis_dtype`` / ``weight_dtype``.
    """
    vis = np.asarray(vis)
    weight = np.asarray(weight)
    nra = weight.shape[-1]
    weight_sum = mmode_weight(weight)
    if mmax is None:
        mmax = vis.shape[-1] // 2

    mvis = np.zeros((mmax + 1, 2, *vis.shape[:-1]), dtype=vis_dtype)
    mweight = np.zeros((mmax + 1, 2, *weight.shape[:-1]), dtype=weight_dtype)
    make_marray(vis, mvis)
    mweight[:] = weight_sum[np.newaxis, np.newaxis]

    if remove_integration_window:
        m = np.arange(mmax + 1)
        w = np.sinc(m / nra)
        inv_w = invert_no_zero(w)
        # in-place like transform.py:636,639: NumPy forms the product in the wider type and casts back
        mvis *= inv_w[(slice(None),) + (np.newaxis,) * (mvis.ndim - 1)]
        mweight *= w[(slice(None),) + (np.newaxis,) * (mweight.ndim - 1)] ** 2
    return mvis, mweight


def unpack_marray(mmodes, n=None):
    """``[m, +/-, ...] -> [..., ntimes]`` FFT ordering, ``transform.py:820-851``."""
    mmodes = np.asarray(mmodes)
    shape = mmodes.shape[2:]
    mmax_plus = mmodes.shape[0] - 1
    if (mmodes[mmax_plus, 1, ...].flatten() == 0).all():
        mmax_minus = mmax_plus - 1
    else:
        mmax_minus = mmax_plus

    if n is None:
        ntimes = mmax_plus + mmax_minus + 1
    else:
        ntimes = n
        mmax_plus = min(ntimes // 2, mmax_plus)
        mmax_minus = min((ntimes - 1) // 2, mmax_minus)

    marray = np.zeros((*shape, ntimes), dtype=np.complex128)
    marray[..., 0] = mmodes[0, 0]
    for mi in range(1, mmax_minus + 1):
        marray[..., mi] = mmodes[mi, 0]
        marray[..., -mi] = mmodes[mi, 1].conj()
    if mmax_plus != mmax_minus:
        marray[..., mmax_plus] = mmodes[mmax_plus, 0]
    return marray


def make_ssarray(mmodes, n=None):
    """``ifft(unpack * ntimes)``, ``transform.py:814-817`` (complex128)."""
    marray = unpack_marray(mmodes, n=n)
    return np.fft.ifft(marray * marray.shape[-1], axis=-1)


def mmode_inverse_transform(mvis, mweight, oddra, nra=None, apply_integration_window=False):
    """``MModeInverseTransform.process`` on plain arrays, ``transform.py:752-792``.

    Returns ``(vis [nfreq, nstack, nra] complex64, weight float32)`` -- the dtypes
    of the SiderealStream datasets the reference assigns into (``containers.py:500-524``).
    Unlike the reference (warning at ``transform.py:713-714``) the inputs are not modified.
    """
    mvis = np.asarray(mvis)
    mweight = np.asarray(mweight)
    mmax = mvis.shape[0] - 1
    nra_cont = 2 * mmax + (1 if oddra else 0)
    nra = nra if nra is not None else nra_cont

    if apply_integration_window:
        m = np.arange(mmax + 1)
        w = np.sinc(m / nra)
        inv_w = invert_no_zero(w)
        sl = (slice(None),) + (np.newaxis,) * (mvis.ndim - 1)
        mvis = mvis * w[sl]
        mweight = mweight * inv_w[sl] ** 2

    ss = make_ssarray(mvis, n=nra)
    nra = ss.shape[-1]
    vis = ss.astype(np.complex64)
    weight = np.empty(vis.shape, dtype=np.float32)
    weight[:] = (mweight[0, 0][..., np.newaxis] / nra).astype(np.float32)
    return vis, weight

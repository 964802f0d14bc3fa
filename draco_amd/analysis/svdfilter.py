"""SVD filtering of m-modes, on the GPU.

Drop-in for ``draco/analysis/svdfilter.py``: ``SVDSpectrumEstimator`` (:11-57), ``SVDFilter``
(:60-149) and ``svd_em`` (:152-187), same config attributes (``niter``, ``global_threshold``,
``local_threshold``) and in-place semantics (``SVDFilter.process`` returns its input with ``vis``
rewritten).  Per m the matrix ``[freq, (msign, base)]`` is decomposed through the
eigen-decomposition of its frequency-side Gram matrix (f64 MFMA product + blocked Jacobi,
``dmm_mmode_svd``); everything the reference does with the factors -- the low-rank refill of
missing entries, the removal of the largest modes -- needs only the left vectors and the data.
Singular values therefore carry an absolute accuracy of ~1e-14 sigma_max^2 / sigma (the price of
the Gram route): exact to working precision for the bright modes the filter is about.

The decomposition couples all frequencies of one m, so with frequency-sharded ranks the m-modes
are exchanged to an m-distribution first (``parallel.freq_to_m`` / ``m_to_freq``: the one real
exchange step of this task, an all-to-all).
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib, parallel
from ..core import containers
from ..core.task import ContainerTask
from ..device import Context, ptr
from .transform import _dev_dataset


def _fill0(mvis, mweight):
    """First guess of the missing entries of every m: ``np.median`` of the present ones (:176), as a device
    ``[n_m]`` complex128 tensor (``dmm_mmode_fill0``: NumPy's complex order, real part first).  ``None`` if nothing
    is missing."""
    if not bool((mweight == 0.0).any()):
        return None
    ctx = Context.get()
    n_m = mvis.shape[0]
    out = ctx.empty((n_m,), np.complex128)
    _lib.check(_lib.lib.dmm_mmode_fill0(ctx.handle, ptr(mvis), ptr(mweight), int(n_m), int(mvis[0].numel()), ptr(out)))
    return out


def _decompose(ctx, mvis, mweight, niter, rank, mode, global_max=0.0, global_thr=0.0, local_thr=0.0, factors=False, fill0=None):
    """Run ``dmm_mmode_svd`` on device arrays ``[n_m, 2, nfreq, nbase]``.

    Returns the spectrum tensor, or ``(spectrum, u, uha)`` with ``factors=True`` (mode 0).
    """
    n_m, _, nfreq, nbase = mvis.shape
    nmode = min(2 * nbase, nfreq)
    spec = ctx.empty((n_m, nmode), np.float64)
    f0 = _fill0(mvis, mweight) if fill0 is None else fill0
    f0_d = None if f0 is None else (f0 if torch.is_tensor(f0) else ctx.to_device(f0, np.complex128))
    u = uha = None
    if factors:
        u = ctx.empty((n_m, nfreq, nmode), np.complex128)
        uha = ctx.empty((n_m, nmode, 2 * nbase), np.complex128)
    _lib.check(
        _lib.lib.dmm_mmode_svd(
            ctx.handle, ptr(mvis), ptr(mweight), int(n_m), int(nfreq), int(nbase), int(niter), int(rank), ptr(f0_d),
            int(mode), float(global_max), float(global_thr), float(local_thr), ptr(spec), ptr(u), ptr(uha),
        )
    )
    return (spec, u, uha) if factors else spec


class SVDSpectrumEstimator(ContainerTask):
    """Calculate the SVD spectrum of a set of m-modes (``svdfilter.py:11-57``).

    Attributes
    ----------
    niter : int
        Number of iterations of EM to perform.
    """

    niter = 5
    _config_names = ("niter",)

    def process(self, mmodes):
        ctx = Context.get()
        mvis = _dev_dataset(mmodes.vis, ctx, np.complex128)
        mweight = _dev_dataset(mmodes.weight, ctx, np.float64)
        mv, mw, lay = parallel.freq_to_m(mvis, mweight)  # identity on one rank
        spec_loc = _decompose(ctx, mv.clone(), mw, self.niter, 5, 0)
        spec_all = parallel.gather_m(spec_loc, lay)
        spec = containers.SVDSpectrum(singularvalue=spec_all.shape[1], axes_from=mmodes, comm=mmodes.comm, allocate=False)
        spec.attach("spectrum", spec_all)
        return spec


class SVDFilter(ContainerTask):
    """SVD filter the m-modes to remove the most correlated components (``svdfilter.py:60-149``).

    Attributes
    ----------
    niter : int
        Number of iterations of EM to perform.
    local_threshold : float
        Cut out modes with singular value higher than `local_threshold` times the largest mode on each m.
    global_threshold : float
        Remove modes with singular value higher than `global_threshold` times the largest mode on any m.
    """

    niter = 5
    global_threshold = 1e-3
    local_threshold = 1e-2
    _config_names = ("niter", "global_threshold", "local_threshold")

    def process(self, mmodes):
        ctx = Context.get()
        mvis = _dev_dataset(mmodes.vis, ctx, np.complex128)
        mweight = _dev_dataset(mmodes.weight, ctx, np.float64)
        mv, mw, lay = parallel.freq_to_m(mvis, mweight)
        # first pass: all singular values, for the largest one on any m of any rank (:101-113)
        spec = _decompose(ctx, mv.clone(), mw, self.niter, 5, 0)
        sv_max = float(spec[:, 0].max().item()) if spec.shape[0] else 0.0
        global_max = parallel.allreduce_max(sv_max)
        # second pass: remove the modes above the combined cut, in place (:122-147)
        mv = mv.contiguous()
        _decompose(ctx, mv, mw, self.niter, 5, 1, global_max, self.global_threshold, self.local_threshold)
        out = parallel.m_to_freq(mv, lay)
        mmodes.vis.set_device(out)
        return mmodes


def svd_em(A, mask, niter=5, rank=5, full_matrices=False):
    """SVD with missing entries by expectation-maximisation (``svdfilter.py:152-187``), on the GPU.

    Returns ``u, sig, vh`` of the matrix as refilled ``niter - 1`` times, like the reference.  The
    GPU path works on ``[freq, (msign, base)]`` matrices: an odd column count is padded with a
    zero column (dropped again from ``vh``).  Rows of ``vh`` whose singular value is below 1e-7 of
    the largest cannot be resolved through the Gram matrix and are returned as zeros;
    ``full_matrices=True`` is not supported.
    """
    if full_matrices:
        raise NotImplementedError("svd_em: full_matrices=True is not available on the GPU path")
    A = np.asarray(A, dtype=np.complex128)
    mask = np.asarray(mask, dtype=bool)
    nrow, ncol = A.shape
    A0, mask0 = A, mask
    if ncol % 2:  # a present, zero column changes neither the singular values nor the other columns of vh
        A = np.concatenate([A, np.zeros((nrow, 1), A.dtype)], axis=1)
        mask = np.concatenate([mask, np.zeros((nrow, 1), bool)], axis=1)
    nc2 = A.shape[1]
    ctx = Context.get()
    vis = np.ascontiguousarray(A.reshape(nrow, 2, nc2 // 2).transpose(1, 0, 2))[np.newaxis]
    w = np.ascontiguousarray((~mask).astype(np.float64).reshape(nrow, 2, nc2 // 2).transpose(1, 0, 2))[np.newaxis]
    mv, mw = ctx.to_device(vis, np.complex128), ctx.to_device(w, np.float64)
    f0 = np.array([np.median(A0[~mask0])]) if mask0.any() else None  # of the ORIGINAL present entries (:176)
    spec, u, uha = _decompose(ctx, mv, mw, niter, rank, 0, factors=True, fill0=f0)
    nmode = min(nrow, ncol)
    sig = spec[0].cpu().numpy()[:nmode]
    u = u[0].cpu().numpy()[:, :nmode]
    uha = uha[0].cpu().numpy()[:nmode, :ncol]
    vh = np.zeros((nmode, ncol), dtype=np.complex128)
    good = sig > 1e-7 * sig[0] if nmode else np.zeros(0, bool)
    vh[good] = uha[good] / sig[good, None]
    return u, sig, vh

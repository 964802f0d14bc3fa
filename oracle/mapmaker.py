"""Oracle: per-(m, freq) map-maker solves.  TEST INFRASTRUCTURE ONLY.

Restates ``draco/analysis/mapmaker.py`` (reference) on plain ndarrays.  ``bm`` is
what ``bt.beam_m(m, fi=f)`` returns: complex128 ``[2, npairs, npol, lmax+1]``
(driftscan [3P], layout inferred from ``mapmaker.py:162``).

* :func:`pinv_svd`        ``pinv_svd``                              ``mapmaker.py:287-300``
* :func:`dirty_solve`     ``DirtyMapMaker._solve_m``                ``mapmaker.py:156-168``
* :func:`ml_solve`        ``MaximumLikelihoodMapMaker._solve_m``    ``mapmaker.py:184-201``
* :func:`wiener_solve`    ``WienerMapMaker._solve_m``               ``mapmaker.py:235-284``
* :func:`find_keys`       ``tools.find_keys``                       ``util/tools.py:95-127``
* :func:`solve_alm`       the (m, freq) loop of ``BaseMapMaker.process`` ``mapmaker.py:50-109``

Pinned by ``tests/golden/mapmaker_*.npz`` (outputs of the reference functions).
"""

from __future__ import annotations

import numpy as np
import scipy.linalg as la


def pinv_svd(M, acond=1e-4, rcond=1e-3):
    """Pseudo-inverse by thin SVD with absolute+relative cut, ``mapmaker.py:287-300``."""
    u, sig, vh = la.svd(M, full_matrices=False)
    rank = np.sum(np.logical_and(sig > rcond * sig.max(), sig > acond))
    psigma_diag = 1.0 / sig[:rank]
    return np.transpose(np.conjugate(np.dot(u[:, :rank] * psigma_diag, vh[:rank])))


def pinv_svd_spectrum(M, acond=1e-4, rcond=1e-3):
    """What ``pinv_svd`` decides on: the singular values of ``M`` and the rank its rule keeps (``mapmaker.py:294-296``:
    ``rank = sum(sig > rcond * sig.max() and sig > acond)``).  Returns ``(rank, sig)``, ``sig`` descending."""
    sig = la.svd(M, compute_uv=False)
    rank = int(np.sum(np.logical_and(sig > rcond * sig.max(), sig > acond)))
    return rank, sig


def ml_spectrum(bm, Ni):
    """Rank and singular values of the matrix ``MaximumLikelihoodMapMaker._solve_m`` hands to ``pinv_svd``
    (``N^-1/2 B``, ``mapmaker.py:188-196``)."""
    ntel = bm.shape[0] * bm.shape[1]
    B = bm.reshape(ntel, -1)
    return pinv_svd_spectrum(B * (np.asarray(Ni).reshape(ntel) ** 0.5)[:, np.newaxis])


def dirty_solve(bm, v, Ni):
    """``a = B^H (Ni * v)``, ``mapmaker.py:156-168`` -> ``[npol, lmax+1]``."""
    npol, nl = bm.shape[-2:]
    ntel = bm.shape[0] * bm.shape[1]
    v = np.asarray(v).reshape(ntel)
    Ni = np.asarray(Ni).reshape(ntel)
    B = bm.reshape(ntel, npol * nl)
    a = np.dot(B.T.conj(), Ni * v)
    return a.reshape(npol, nl)


def dirty_solve_many(bm, vs, Nis):
    """D days against one tile: column d is ``dirty_solve(bm, vs[d], Nis[d])`` (``mapmaker.py:156-168`` called once per
    pipeline item with the same ``beam_m``) -- as ONE matrix product ``B^H [Ni_d * v_d]_d``, the form a CPU would use to
    share the read of B between the days.  Returns ``[D, npol, lmax+1]``."""
    npol, nl = bm.shape[-2:]
    ntel = bm.shape[0] * bm.shape[1]
    W = (np.asarray(Nis).reshape(-1, ntel) * np.asarray(vs).reshape(-1, ntel)).T  # [ntel, D]
    a = np.dot(bm.reshape(ntel, npol * nl).T.conj(), W)
    return a.T.reshape(-1, npol, nl)


def ml_solve(bm, v, Ni):
    """``a = pinv(N^-1/2 B) N^-1/2 v``, ``mapmaker.py:184-201``."""
    npol, nl = bm.shape[-2:]
    ntel = bm.shape[0] * bm.shape[1]
    v = np.asarray(v).reshape(ntel)
    Ni = np.asarray(Ni).reshape(ntel)
    B = bm.reshape(ntel, npol * nl)
    Nh = Ni**0.5
    ib = pinv_svd(B * Nh[:, np.newaxis])
    a = np.dot(ib, Nh * v)
    return a.reshape(npol, nl)


def ml_solve_with_spectrum(bm, v, Ni, acond=1e-4, rcond=1e-3):
    """``ml_solve`` and ``ml_spectrum`` from ONE decomposition (the SVD of a 758 x 2052 cfg-3 tile takes seconds): the
    solution exactly as ``pinv_svd`` forms it (``mapmaker.py:184-201, 287-300``), the rank its rule keeps and the singular
    values it decided on.  Returns ``(a [npol, lmax+1], rank, sig)``."""
    npol, nl = bm.shape[-2:]
    ntel = bm.shape[0] * bm.shape[1]
    v = np.asarray(v).reshape(ntel)
    Ni = np.asarray(Ni).reshape(ntel)
    Nh = Ni**0.5
    u, sig, vh = la.svd(bm.reshape(ntel, npol * nl) * Nh[:, np.newaxis], full_matrices=False)
    rank = int(np.sum(np.logical_and(sig > rcond * sig.max(), sig > acond)))
    ib = np.transpose(np.conjugate(np.dot(u[:, :rank] * (1.0 / sig[:rank]), vh[:rank])))
    return np.dot(ib, Nh * v).reshape(npol, nl), rank, sig


def wiener_prior(lmax, m, prior_amp=1.0, prior_tilt=0.5, npol=4):
    """Diagonal of S for ``l >= m``, ``mapmaker.py:260-264`` (the reference tiles it x4)."""
    l = np.arange(lmax + 1)
    l[0] = 1
    l = l[m:]
    cl_TT = prior_amp**2 * l ** (-prior_tilt)
    return np.concatenate([cl_TT] * npol)


def wiener_solve(bm, m, v, Ni, prior_amp=1.0, prior_tilt=0.5):
    """Wiener filter for one (m, f), ``mapmaker.py:235-284``.

    Both branches of the reference (``ntel > nsky`` at :267 and the block-inverse
    form at :275-278) are kept; they are algebraically the same estimator
    ``(S^-1 + B~^H B~)^-1 B~^H v~``.  ``sym_pos=True`` (removed from SciPy) is
    restated as ``assume_a="pos"`` -- the intended Hermitian-PD solve.
    The reference hard-codes 4 polarisations for S (:264); so does this.
    """
    npol, nl = bm.shape[-2:]
    ntel = bm.shape[0] * bm.shape[1]
    nsky = npol * nl
    lmax = nl - 1
    B = bm[..., m:].reshape(ntel, -1)
    v = np.asarray(v).reshape(ntel)
    Ni = np.asarray(Ni).reshape(ntel)
    Nh = Ni**0.5
    bmt = B * Nh[:, np.newaxis]
    bth = bmt.T.conj()
    vt = Nh * v
    S_diag = wiener_prior(lmax, m, prior_amp, prior_tilt, 4)

    if ntel > nsky:
        Ci = np.diag(1.0 / S_diag) + np.dot(bth, bmt)
        a_dirty = np.dot(bth, vt)
        a_wiener = la.solve(Ci, a_dirty, assume_a="pos")
    else:
        pCi = np.identity(ntel) + np.dot(bmt * S_diag[np.newaxis, :], bth)
        v_int = la.solve(pCi, vt, assume_a="pos")
        a_wiener = S_diag * np.dot(bth, v_int)

    a = np.zeros((npol, nl), dtype=np.result_type(v.dtype, np.complex128))
    a[:, m:] = a_wiener.reshape(npol, -1)
    return a


def find_keys(key_list, keys, require_match=False):
    """Exact-match index lookup, ``util/tools.py:95-127``."""
    try:
        dct = {tuple(kk): ii for ii, kk in enumerate(key_list)}
        index = [dct.get(tuple(key)) for key in keys]
    except TypeError:
        dct = {kk: ii for ii, kk in enumerate(key_list)}
        index = [dct.get(key) for key in keys]
    if require_match and any(ind is None for ind in index):
        raise ValueError("Could not find all of the keys.")
    return index


def solve_alm(kind, beam_m, mvis, mweight, lmax, tel_mmax, freq_ind, npol=4, **prior):
    """The (m, freq) loop of ``BaseMapMaker.process``, ``mapmaker.py:50-109``.

    ``mvis [n_m, 2, nfreq, nstack]`` c128, ``mweight`` f64, ``beam_m(m, f)`` callable,
    ``freq_ind[fi]`` = index into the beam-transfer frequencies (``mapmaker.py:59``).
    Returns the square ``alm [nfreq, 4, lmax+1, lmax+1]`` (``mapmaker.py:102-109``; the
    reference allocates 4 pols regardless of ``num_pol_sky``, :71).
    """
    n_m, _, nfreq, _ = mvis.shape
    mmax = min(tel_mmax, n_m - 1)
    alm = np.zeros((nfreq, 4, lmax + 1, lmax + 1), dtype=np.complex128)
    for m in range(mmax + 1):
        for fi in range(nfreq):
            bm = beam_m(m, freq_ind[fi])
            v = mvis[m, :, fi]
            Ni = mweight[m, :, fi]
            if kind == "dirty":
                a = dirty_solve(bm, v, Ni)
            elif kind == "ml":
                a = ml_solve(bm, v, Ni)
            elif kind == "wiener":
                a = wiener_solve(bm, m, v, Ni, **prior)
            else:
                raise ValueError(kind)
            alm[fi, :, :, m] = a  # [npol, lmax+1] into 4 slots: npol = 1 broadcasts (mapmaker.py:91-94; pinned by mapmaker_process.npz)
    return alm

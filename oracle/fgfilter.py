"""Oracle: SVDModeProject / KLModeProject.  TEST INFRASTRUCTURE ONLY.

Restates the task loops of ``draco/analysis/fgfilter.py:68-146,168-239`` on plain arrays.  The four driftscan calls
inside them [3P, absent] are restated as what their call sites imply -- per frequency (SVD) or per m (KL) a basis
matrix times the data vector: **parity of that arithmetic is unpinned**; the loops around them are pinned by
``tests/golden/fgfilter.npz`` (the reference classes run from source against duck-typed products).
"""

from __future__ import annotations

import numpy as np


def svd_forward(mvis, mweight, ut, ndofmax):
    """``mvis [n_m, 2, nfreq, npairs]`` -> ``(vis [n_m, ndofmax], weight, nmode)`` (``fgfilter.py:68-98``).
    ``ut(m, f) -> [n_f, ntel]``."""
    n_m, _, nfreq, npairs = mvis.shape
    vis = np.zeros((n_m, ndofmax), dtype=np.complex128)
    weight = np.zeros((n_m, ndofmax))
    nmode = np.zeros(n_m, dtype=np.int32)
    for mi in range(n_m):
        tm = mvis[mi].transpose((1, 0, 2)).reshape(nfreq, 2 * npairs)  # :86
        svdm = np.concatenate([np.asarray(ut(mi, f)) @ tm[f] for f in range(nfreq)])  # bt.project_vector_telescope_to_svd
        nmode[mi] = len(svdm)
        vis[mi, : nmode[mi]] = svdm
        weight[mi] = np.median(mweight[mi])  # :94
    return vis, weight, nmode


def svd_backward(svis, sweight, ut_inv, svd_len, nfreq, npairs):
    """``svis [n_m, ndofmax]`` -> ``(mvis [n_m, 2, nfreq, npairs], mweight)`` (``fgfilter.py:100-146``).
    ``ut_inv(m, f) -> [ntel, n_f]``, ``svd_len(m) -> int[nfreq]``."""
    n_m = svis.shape[0]
    mvis = np.zeros((n_m, 2, nfreq, npairs), dtype=np.complex128)
    mweight = np.zeros(mvis.shape)
    for mi in range(n_m):
        bounds = np.concatenate([[0], np.cumsum(svd_len(mi))])
        tm = np.stack([(np.asarray(ut_inv(mi, f)) @ svis[mi, bounds[f] : bounds[f + 1]]).reshape(2, npairs) for f in range(nfreq)])
        mvis[mi] = tm.transpose((1, 0, 2))  # :135
        mweight[mi] = np.median(sweight[mi])  # :141
    return mvis, mweight


def kl_apply(vis, weight, nmode, mats, nmax):
    """``out[m, :n] = mats(m) @ vis[m, :nmode[m]]`` with the median weight rule (``fgfilter.py:189-204,225-239``)."""
    n_m = vis.shape[0]
    out = np.zeros((n_m, nmax), dtype=np.complex128)
    wout = np.zeros((n_m, nmax))
    nout = np.zeros(n_m, dtype=np.int32)
    for mi in range(n_m):
        r = np.asarray(mats(mi)) @ vis[mi][: nmode[mi]]
        nout[mi] = len(r)
        out[mi, : nout[mi]] = r
        wout[mi] = np.median(weight[mi])
    return out, wout, nout

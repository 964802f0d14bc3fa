"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's m-mode SVD filter.

    reference: draco/analysis/svdfilter.py
      svd_em                        :152-187
      SVDSpectrumEstimator.process  :22-57
      SVDFilter.process             :79-149

Pinned by ``tests/golden/svdfilter.npz`` (outputs of the reference's own functions, see
``oracle/gen_golden.py --only-svd``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline may import this package.
"""

import numpy as np
import scipy.linalg as la


def svd_em(A, mask, niter=5, rank=5, full_matrices=False):
    """SVD with missing entries by expectation-maximisation (svdfilter.py:152-187).

    Returns the factors of the LAST decomposition, i.e. of the matrix before the final refill.
    """
    A = A.copy()
    A[mask] = np.median(A[~mask])  # :176 (complex median: NumPy's lexicographic order)
    for _ in range(niter):  # :181-185
        u, sig, vh = la.svd(A, full_matrices=full_matrices, overwrite_a=False)
        low_rank = np.dot(u[:, :rank] * sig[:rank], vh[:rank])
        A[mask] = low_rank[mask]
    return u, sig, vh


def _matrix_of_m(vis_m, weight_m):
    """[msign, freq, base] -> [freq, msign*base] and its missing-entry mask (:45-51)."""
    nfreq = vis_m.shape[1]
    a = vis_m.transpose((1, 0, 2)).reshape(nfreq, -1)
    w = weight_m.transpose((1, 0, 2)).reshape(nfreq, -1)
    return a, w == 0.0


def svd_spectrum(vis, weight, niter=5):
    """``spectrum [m, nmode]`` of MModes ``vis/weight [m, msign, freq, base]`` (:35-57)."""
    nm, _, nfreq, nbase = vis.shape
    nmode = min(2 * nbase, nfreq)
    spec = np.zeros((nm, nmode))
    for m in range(nm):
        a, mask = _matrix_of_m(vis[m], weight[m])
        spec[m] = svd_em(a, mask, niter=niter)[1]
    return spec


def svd_filter(vis, weight, niter=5, global_threshold=1e-3, local_threshold=1e-2, global_max=None):
    """Filtered copy of ``vis`` (:91-149).  ``global_max``: override of the all-rank maximum."""
    nm, _, nfreq, nbase = vis.shape
    out = vis.copy()
    if global_max is None:
        global_max = 0.0
        for m in range(nm):
            a, mask = _matrix_of_m(vis[m], weight[m])
            global_max = max(svd_em(a, mask, niter=niter)[1][0], global_max)
    for m in range(nm):
        a, mask = _matrix_of_m(vis[m], weight[m])
        u, sig, vh = svd_em(a, mask, niter=niter)
        global_cut = (sig > global_threshold * global_max).sum()
        local_cut = (sig > local_threshold * sig[0]).sum()
        cut = max(global_cut, local_cut)
        sig = sig.copy()
        sig[:cut] = 0.0  # :139 (the LARGEST modes are the ones removed)
        a = np.dot(u, sig[:, np.newaxis] * vh)
        out[m] = a.reshape(nfreq, 2, -1).transpose((1, 0, 2))
    return out

"""Oracle: SimulateSidereal forward model.  TEST INFRASTRUCTURE ONLY.

Restates ``SimulateSidereal.process`` (reference ``draco/synthesis/stream.py:48-178``) on
plain arrays, with the third-party pieces replaced by their oracle counterparts:
``hputil.sphtrans_sky`` -> :func:`oracle.sht.sphtrans_sky` (parity unpinned, see there; everything after
the SHT is pinned by the reference's own ``process`` run from source, ``tests/golden/stream_simulate.npz``),
``bt.project_vector_sky_to_telescope(m, a)`` -> ``B_m[f] @ a`` per frequency
(driftscan semantics inferred from the call site ``stream.py:109-112``).
"""

from __future__ import annotations

import numpy as np

from . import sht


def simulate_sidereal(skymap, beam_m, lmax, mmax, npairs, npol=4, niter=3):
    """``map [nfreq, npol, npix]`` -> ``vis [nfreq, npairs, 2*mmax+1]`` complex64 (``stream.py:64-175``)."""
    skymap = np.asarray(skymap, dtype=np.float64)
    row_alm = sht.sphtrans_sky(skymap[:, :npol], lmax, niter)  # [nfreq, npol, lmax+1, lmax+1], stream.py:85
    return simulate_from_alm(row_alm, beam_m, lmax, mmax, npairs, npol)


def simulate_from_alm(row_alm, beam_m, lmax, mmax, npairs, npol=4):
    """Everything of ``SimulateSidereal.process`` after the SHT (``stream.py:90-140,175``).

    ``row_alm [nfreq, npol, lmax+1, lmax+1]`` is what ``hputil.sphtrans_sky`` returned.  Pinned by
    ``tests/golden/stream_simulate.npz`` (the reference's ``process`` executed from source with a recorded a_lm).
    """
    nfreq = row_alm.shape[0]
    ntime = 2 * mmax + 1  # stream.py:76
    row_alm = row_alm[..., : mmax + 1]  # stream.py:90
    ntel = 2 * npairs
    vis_data = np.zeros((mmax + 1, nfreq, ntel), dtype=np.complex128)
    for mi in range(mmax + 1):
        for f in range(nfreq):
            B = beam_m(mi, f).reshape(ntel, npol * (lmax + 1))
            vis_data[mi, f] = B @ row_alm[f, :, :, mi].reshape(-1)  # stream.py:110
    tmp = vis_data.transpose(0, 2, 1).reshape(mmax + 1, 2, npairs, nfreq)  # stream.py:116-120
    col_vis = np.zeros((npairs, nfreq, ntime), dtype=np.complex128)
    col_vis[..., 0] = tmp[0, 0]
    for mi in range(1, mmax + 1):
        col_vis[..., mi] = tmp[mi, 0]
        col_vis[..., -mi] = tmp[mi, 1].conj()  # conjugate only, stream.py:131-133
    vis_stream = np.fft.ifft(col_vis, axis=-1) * ntime  # stream.py:138
    return vis_stream.transpose(1, 0, 2).astype(np.complex64)  # stream.py:139-140,175

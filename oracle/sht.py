"""Oracle: HEALPix spherical-harmonic transforms.  TEST INFRASTRUCTURE ONLY.

Stands in for ``cora.util.hputil.sphtrans_inv_sky`` (reference ``mapmaker.py:112``) and
``hputil.sphtrans_sky`` (``stream.py:85``) -- cora [3P] wraps healpy's ``alm2map`` /
``map2alm``; neither package is present in the build container and the reference holds no
fixture for them, so this stage is **parity unpinned**: it restates the published
HEALPix conventions (Gorski et al. 2005; the HEALPix primer) and is validated by
mathematical identities in ``tests/test_oracle_sht.py``:

* pixelisation: RING scheme ring geometry (``ring_info``) -- pixel centres have the exact
  HEALPix properties (equal areas, ring counts, 12*nside^2 pixels, z symmetric);
* scalar: ``Y_lm = lambda_lm(cos theta) e^{i m phi}`` with Condon-Shortley phase,
  real fields: ``a_{l,-m} = (-1)^m conj(a_lm)``, only ``m >= 0`` stored;
* polarisation (healpy ``pol=True``): ``(Q +- iU) = sum_lm a^{+-2}_lm  +-2Y_lm``,
  ``a^{+-2}_lm = -(E_lm +- i B_lm)``, spin-weighted harmonics by the Newman-Penrose
  edth construction; evaluated through the Kamionkowski-Kosowsky-Stebbins
  ``F_{1,lm}``, ``F_{2,lm}`` functions as HEALPix does; checked here against the edth
  definition and the closed forms for l = 2;
* alm packing of the reference: ``alm[..., l, m]`` square, ``m <= l`` used.

Two implementations: ``*_direct`` (definition level, O(npix * lmax^2), tiny sizes) and the
ring-based ones (Legendre recurrence per ring + Fourier sum per ring) used as the checker
for the HIP kernels.
"""

from __future__ import annotations

import numpy as np


# ------------------------------------------------------------------ pixelisation
def ring_info(nside):
    """Per-ring geometry of the RING scheme: ``(z, nphi, phi0, start)`` for rings 1..4*nside-1."""
    nside = int(nside)
    nring = 4 * nside - 1
    i = np.arange(1, nring + 1)
    z = np.empty(nring)
    nphi = np.empty(nring, dtype=np.int64)
    phi0 = np.empty(nring)
    start = np.empty(nring, dtype=np.int64)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    for k, ir in enumerate(i):
        if ir < nside:  # north cap
            z[k] = 1.0 - ir * ir / (3.0 * nside * nside)
            nphi[k] = 4 * ir
            phi0[k] = np.pi / (4.0 * ir)
            start[k] = 2 * ir * (ir - 1)
        elif ir <= 3 * nside:  # equatorial belt
            z[k] = (2.0 * nside - ir) * 2.0 / (3.0 * nside)
            nphi[k] = 4 * nside
            s = (ir - nside + 1) & 1
            phi0[k] = s * np.pi / (4.0 * nside)
            start[k] = ncap + (ir - nside) * 4 * nside
        else:  # south cap
            ip = 4 * nside - ir
            z[k] = -(1.0 - ip * ip / (3.0 * nside * nside))
            nphi[k] = 4 * ip
            phi0[k] = np.pi / (4.0 * ip)
            start[k] = npix - 2 * ip * (ip + 1)
    return z, nphi, phi0, start


def sin_theta(z, nside):
    """sin(theta) of the ring centres, formed without cancellation near the poles."""
    return np.sqrt((1.0 - z) * (1.0 + z))


def pix_angles(nside):
    """(theta, phi) of every pixel centre, RING order."""
    z, nphi, phi0, start = ring_info(nside)
    th = np.empty(12 * nside * nside)
    ph = np.empty_like(th)
    for zr, n, p0, s in zip(z, nphi, phi0, start):
        th[s : s + n] = np.arccos(zr)
        ph[s : s + n] = p0 + 2.0 * np.pi * np.arange(n) / n
    return th, ph


def ang2pix_ring(nside, theta, phi):
    """RING pixel index containing the direction(s) ``(theta, phi)`` -- the published HEALPix point-in-pixel rule
    (Gorski et al. 2005, section 4.1: pixel boundaries are lines of constant ``phi +/- f(z)``), written from the
    boundary equations, NOT from :func:`ring_info`.  Used only as an independent witness of the ring geometry
    (``tests/test_oracle_sht.py``); the reference's own healpy call sites use it with ``lonlat=True``
    (``draco/analysis/beamform.py:1708,1760``: ``ang2pix(nside, ra_deg, dec_deg, lonlat=True)``).
    """
    nside = int(nside)
    theta, phi = np.broadcast_arrays(np.asarray(theta, dtype=np.float64), np.asarray(phi, dtype=np.float64))
    z = np.cos(theta)
    za = np.abs(z)
    tt = np.mod(phi, 2.0 * np.pi) / (0.5 * np.pi)  # [0, 4)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    out = np.empty(z.shape, dtype=np.int64)
    eq = za <= 2.0 / 3.0
    # equatorial belt: the two families of boundary lines are straight in (z, phi)
    t1 = nside * (0.5 + tt[eq])
    t2 = nside * z[eq] * 0.75
    jp = np.floor(t1 - t2).astype(np.int64)  # ascending edge line index
    jm = np.floor(t1 + t2).astype(np.int64)  # descending edge line index
    ir = nside + 1 + jp - jm  # ring counted from z = 2/3, in 1 .. 2 nside + 1
    kshift = 1 - (ir & 1)
    ip = (jp + jm - nside + kshift + 1) // 2
    ip = np.mod(ip, 4 * nside)
    out[eq] = ncap + (ir - 1) * 4 * nside + ip
    # polar caps: boundaries are curves sqrt(3 (1 - |z|)) * {phi_t, 1 - phi_t} = integer / nside
    pc = ~eq
    tp = tt[pc] - np.floor(tt[pc])
    tmp = nside * np.sqrt(3.0 * (1.0 - za[pc]))
    jp = np.floor(tp * tmp).astype(np.int64)
    jm = np.floor((1.0 - tp) * tmp).astype(np.int64)
    ir = jp + jm + 1  # ring counted from the closest pole
    ip = np.mod(np.floor(tt[pc] * ir).astype(np.int64), 4 * ir)
    out[pc] = np.where(z[pc] > 0, 2 * ir * (ir - 1) + ip, npix - 2 * ir * (ir + 1) + ip)
    return out


def ang2pix_lonlat(nside, lon_deg, lat_deg):
    """``healpy.ang2pix(nside, lon, lat, lonlat=True)``: longitude = phi (RA), latitude = 90 deg - theta (dec)."""
    return ang2pix_ring(nside, np.radians(90.0 - np.asarray(lat_deg, dtype=np.float64)), np.radians(lon_deg))


# ------------------------------------------------------------- Legendre functions
def lambda_lm(lmax, m, x):
    """Normalised associated Legendre ``lambda_lm(x)``, ``l = m..lmax`` -> ``[lmax+1, len(x)]`` (rows l<m zero).

    ``lambda_lm = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) P_lm(x)`` with the Condon-Shortley phase.
    Extended-range start (log-scaled ``sin^m``) so that high m near the poles do not underflow
    into garbage: values below 1e-300 are flushed to exact zero until the recurrence has grown
    back into range.
    """
    x = np.asarray(x, dtype=np.float64)
    s = np.sqrt((1.0 - x) * (1.0 + x))
    out = np.zeros((lmax + 1, x.size))
    if m > lmax:
        return out
    k = np.arange(1, m + 1)
    log_pref = 0.5 * (np.log(2 * m + 1.0) - np.log(4 * np.pi) + np.sum(np.log((2 * k - 1.0) / (2.0 * k))))
    with np.errstate(divide="ignore"):
        log_mm = log_pref + m * np.log(s)
    # carry a per-ring exponent: lam = v * exp(e)  (sin(theta) = 0 only at the exact poles: lam_mm = 0 for m > 0)
    e = np.where(np.isfinite(log_mm), log_mm, -1e300) if m > 0 else np.full_like(x, log_pref)
    v_prev = np.zeros_like(x)
    v = np.full_like(x, (-1.0) ** m)
    with np.errstate(under="ignore"):
        out[m] = v * np.exp(e)

    def A(l):
        return np.sqrt((l * l - m * m) / (4.0 * l * l - 1.0))

    for l in range(m + 1, lmax + 1):
        v_new = (x * v - A(l - 1) * v_prev) / A(l)  # A(m) = 0 closes the two-term start
        v_prev, v = v, v_new
        big = np.abs(v) > 1e100
        if big.any():
            v[big] *= 1e-100
            v_prev[big] *= 1e-100
            e[big] += np.log(1e100)
        with np.errstate(over="ignore", under="ignore"):
            out[l] = v * np.exp(e)
    return out


def _spin_F(lmax, m, x, lam):
    """KKS ``F_1, F_2`` (``[lmax+1, nx]``) from ``lam = lambda_lm`` of the same m (rows l<2 zero)."""
    s2 = (1.0 - x) * (1.0 + x)
    F1 = np.zeros_like(lam)
    F2 = np.zeros_like(lam)
    for l in range(max(2, m), lmax + 1):
        c = 2.0 / np.sqrt((l - 1.0) * l * (l + 1.0) * (l + 2.0))
        lam_lm1 = lam[l - 1] if l - 1 >= m else 0.0
        d = np.sqrt((2 * l + 1.0) / (2 * l - 1.0) * (l * l - m * m))
        F1[l] = c * (-((l - m * m) / s2 + 0.5 * l * (l - 1)) * lam[l] + (x / s2) * d * lam_lm1)
        F2[l] = c * (m / s2) * (-(l - 1) * x * lam[l] + d * lam_lm1)
    return F1, F2


# ------------------------------------------------- spin-weighted harmonics by edth
def spin2_Y_edth(l, m, theta, h=1e-3):
    """``(+2Y_lm, -2Y_lm)`` theta-dependence by applying edth / edth-bar twice to ``lambda_lm``.

    For ``eta = f(theta) e^{i m phi}`` of spin s:
    ``edth eta = -(f' - s cot f - (m/sin) f) e^{i m phi}``,
    ``edthbar eta = -(f' + s cot f + (m/sin) f) e^{i m phi}``.
    Derivatives by an 8th-order central difference; definition-level check only.
    """
    co = np.array([1 / 280, -4 / 105, 1 / 5, -4 / 5, 0, 4 / 5, -1 / 5, 4 / 105, -1 / 280])

    def lam(t):
        return lambda_lm(l, abs(m), np.cos(t))[l] * ((-1.0) ** m if m < 0 else 1.0)

    def d(fun, t):
        return sum(c * fun(t + (k - 4) * h) for k, c in enumerate(co)) / h

    def edth(fun, s):
        return lambda t: -(d(fun, t) - s * fun(t) / np.tan(t) - m / np.sin(t) * fun(t))

    def edthbar(fun, s):
        return lambda t: -(d(fun, t) + s * fun(t) / np.tan(t) + m / np.sin(t) * fun(t))

    norm = np.sqrt(1.0 / ((l - 1.0) * l * (l + 1.0) * (l + 2.0)))
    p2 = edth(edth(lam, 0), 1)(theta) * norm
    m2 = edthbar(edthbar(lam, 0), -1)(theta) * norm  # (-1)^s with s = -2 is +1
    return p2, m2


# --------------------------------------------------------- definition-level synthesis
def alm2map_direct(alm, nside, spin_pair=None):
    """Definition-level synthesis at pixel centres.  ``alm [lmax+1, lmax+1]`` (l, m>=0).

    ``spin_pair=None``: scalar map.  Else ``alm`` is ``(almE, almB)`` and ``(Q, U)`` is returned.
    """
    th, ph = pix_angles(nside)
    x = np.cos(th)
    if spin_pair is None:
        lmax = alm.shape[0] - 1
        out = np.zeros(th.size)
        for m in range(lmax + 1):
            lam = lambda_lm(lmax, m, x)
            b = (alm[:, m, None] * lam).sum(axis=0)
            out += (1.0 if m == 0 else 2.0) * (b * np.exp(1j * m * ph)).real
        return out
    almE, almB = alm
    lmax = almE.shape[0] - 1
    Q = np.zeros(th.size)
    U = np.zeros(th.size)
    for m in range(lmax + 1):
        lam = lambda_lm(lmax, m, x)
        F1, F2 = _spin_F(lmax, m, x, lam)
        E, B = almE[:, m, None], almB[:, m, None]
        bQ = -(E * F1 + 1j * B * F2).sum(axis=0)
        bU = -(B * F1 - 1j * E * F2).sum(axis=0)
        ph_m = np.exp(1j * m * ph)
        fac = 1.0 if m == 0 else 2.0
        Q += fac * (bQ * ph_m).real
        U += fac * (bU * ph_m).real
    return Q, U


# ------------------------------------------------------------------- ring-based SHT
def _ring_pairs(nside):
    """Northern rings (incl. equator) and the index of their southern mirror (or -1)."""
    nring = 4 * nside - 1
    north = np.arange(0, 2 * nside)  # ring indices 0..2nside-1  (ring 2nside-1 is the equator)
    south = nring - 1 - north
    south[south == north] = -1
    return north, south


def legendre_synthesis(alms, nside, mmax=None):
    """``b[pol, ring, m]`` for pol in (I, Q, U, V) from ``alms [4 (T,E,B,V), lmax+1, lmax+1]``.

    ``npol = 1``: ``alms [1, ...]`` -> ``b[1, ring, m]``.
    """
    npol, nl, nm = alms.shape
    lmax = nl - 1
    mmax = nm - 1 if mmax is None else mmax
    z, nphi, phi0, start = ring_info(nside)
    nring = z.size
    b = np.zeros((npol, nring, mmax + 1), dtype=np.complex128)
    for m in range(mmax + 1):
        lam = lambda_lm(lmax, m, z)
        b[0, :, m] = (alms[0, :, m, None] * lam).sum(axis=0)
        if npol == 4:
            b[3, :, m] = (alms[3, :, m, None] * lam).sum(axis=0)
            F1, F2 = _spin_F(lmax, m, z, lam)
            E, B = alms[1, :, m, None], alms[2, :, m, None]
            b[1, :, m] = -(E * F1 + 1j * B * F2).sum(axis=0)
            b[2, :, m] = -(B * F1 - 1j * E * F2).sum(axis=0)
    return b


def ring_synthesis(b, nside):
    """``map[pol, pix] = Re sum_m c_m b[pol, ring, m] e^{i m phi}`` for every ring."""
    npol, nring, nm = b.shape
    z, nphi, phi0, start = ring_info(nside)
    out = np.zeros((npol, 12 * nside * nside))
    m = np.arange(nm)
    fac = np.where(m == 0, 1.0, 2.0)
    for r in range(nring):
        n = int(nphi[r])
        c = b[:, r, :] * (fac * np.exp(1j * m * phi0[r]))
        h = np.zeros((npol, n), dtype=np.complex128)
        np.add.at(h, (slice(None), m % n), c)
        out[:, start[r] : start[r] + n] = (np.fft.ifft(h, axis=-1) * n).real
    return out


def alm2map(alms, nside):
    """``alms [npol, lmax+1, lmax+1]`` (T,E,B,V or T) -> ``map [npol, npix]`` (I,Q,U,V or I)."""
    return ring_synthesis(legendre_synthesis(np.asarray(alms, dtype=np.complex128), nside), nside)


def ring_analysis(maps, nside, mmax):
    """``g[pol, ring, m] = (4 pi / npix) sum_j map_j e^{-i m phi_j}``."""
    npol = maps.shape[0]
    z, nphi, phi0, start = ring_info(nside)
    nring = z.size
    g = np.zeros((npol, nring, mmax + 1), dtype=np.complex128)
    m = np.arange(mmax + 1)
    w = 4.0 * np.pi / (12 * nside * nside)
    for r in range(nring):
        n = int(nphi[r])
        F = np.fft.fft(maps[:, start[r] : start[r] + n], axis=-1)
        g[:, r, :] = F[:, m % n] * np.exp(-1j * m * phi0[r]) * w
    return g


def legendre_analysis(g, nside, lmax):
    """Adjoint of :func:`legendre_synthesis`: ``alms [npol, lmax+1, lmax+1]`` from ring coefficients."""
    npol, nring, nm = g.shape
    mmax = nm - 1
    z, nphi, phi0, start = ring_info(nside)
    alms = np.zeros((npol, lmax + 1, lmax + 1), dtype=np.complex128)
    for m in range(min(mmax, lmax) + 1):
        lam = lambda_lm(lmax, m, z)
        alms[0, :, m] = (lam * g[0, None, :, m]).sum(axis=1)
        if npol == 4:
            alms[3, :, m] = (lam * g[3, None, :, m]).sum(axis=1)
            F1, F2 = _spin_F(lmax, m, z, lam)
            gQ, gU = g[1, None, :, m], g[2, None, :, m]
            # E = -int[Q X1* + i U X2*],  B = -int[U X1* - i Q X2*]   (X = F e^{i m phi}, F real)
            alms[1, :, m] = -(F1 * gQ + 1j * F2 * gU).sum(axis=1)
            alms[2, :, m] = -(F1 * gU - 1j * F2 * gQ).sum(axis=1)
    return alms


def map2alm(maps, lmax, niter=3):
    """Quadrature analysis with equal pixel weights + ``niter`` Jacobi refinements (healpy's ``iter``)."""
    maps = np.asarray(maps, dtype=np.float64)
    nside = int(round(np.sqrt(maps.shape[-1] // 12)))

    def analyse(mp):
        return legendre_analysis(ring_analysis(mp, nside, lmax), nside, lmax)

    alms = analyse(maps)
    for _ in range(niter):
        alms = alms + analyse(maps - alm2map(alms, nside))
    return alms


# ------------------------------------------------------------- cora-shaped wrappers
def sphtrans_inv_sky(alm, nside):
    """``alm [nfreq, npol, lmax+1, lmax+1] -> map [nfreq, npol, npix]`` (``mapmaker.py:112``)."""
    alm = np.asarray(alm)
    return np.stack([alm2map(alm[f], nside) for f in range(alm.shape[0])])


def sphtrans_sky(skymap, lmax, niter=3):
    """``map [nfreq, npol, npix] -> alm [nfreq, npol, lmax+1, lmax+1]`` (``stream.py:85``)."""
    skymap = np.asarray(skymap)
    return np.stack([map2alm(skymap[f], lmax, niter) for f in range(skymap.shape[0])])

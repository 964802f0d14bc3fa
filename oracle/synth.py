"""Oracle-side synthetic inputs.  TEST INFRASTRUCTURE ONLY.

* :func:`beam_tile` -- the counter-hash beam-transfer tile generator, restated
  independently of ``draco_amd.core.products.synth_beam_tile`` and of the HIP kernel
  ``csrc/synth.hip`` (all three must agree bit for bit in float64);
* :data:`CONFIGS` -- the concrete shapes of BASELINE.json's configs;
* :func:`sidereal_inputs` -- seeded ``vis`` / ``weight`` of a SiderealStream.
"""

from __future__ import annotations

import numpy as np

MASK = (1 << 64) - 1


def _mix64_int(z: int) -> int:
    z &= MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def tile_key(seed: int, m: int, f: int) -> int:
    seed, m, f = int(seed), int(m), int(f)
    a = _mix64_int(seed + 0x9E3779B97F4A7C15 * (m + 1))
    return _mix64_int(a ^ ((0xD1B54A32D192ED03 * (f + 1)) & MASK))


def beam_tile(seed, m, f, npairs, npol, lmax):
    """complex128 ``[2, npairs, npol, lmax+1]``, uniform real/imag parts, variance 1/ntel, zero for l<m."""
    m = int(m)
    ntel = 2 * npairs
    scale = np.sqrt(3.0 / (2.0 * ntel))
    key = np.uint64(tile_key(seed, m, f))
    ctr = np.arange(ntel * npol * (lmax + 1), dtype=np.uint64)
    with np.errstate(over="ignore"):
        h1 = _mix64(key + np.uint64(2) * ctr)
        h2 = _mix64(key + np.uint64(2) * ctr + np.uint64(1))
    re = ((h1 >> np.uint64(11)).astype(np.float64) * 2.0**-53 * 2.0 - 1.0) * scale
    im = ((h2 >> np.uint64(11)).astype(np.float64) * 2.0**-53 * 2.0 - 1.0) * scale
    b = (re + 1j * im).reshape(2, npairs, npol, lmax + 1)
    b[..., :m] = 0.0
    return b


def beam_row(seed, m, f, row, npairs, npol, lmax):
    """Row ``row`` (0 .. 2*npairs-1: sign-major, pair-minor) of :func:`beam_tile` as ``[npol, lmax+1]`` without
    generating the rest of the tile (config-sized read-back checks touch every (m, f) but one row of each)."""
    m = int(m)
    ntel = 2 * npairs
    scale = np.sqrt(3.0 / (2.0 * ntel))
    key = np.uint64(tile_key(seed, m, f))
    n = npol * (lmax + 1)
    ctr = np.arange(int(row) * n, (int(row) + 1) * n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h1 = _mix64(key + np.uint64(2) * ctr)
        h2 = _mix64(key + np.uint64(2) * ctr + np.uint64(1))
    re = ((h1 >> np.uint64(11)).astype(np.float64) * 2.0**-53 * 2.0 - 1.0) * scale
    im = ((h2 >> np.uint64(11)).astype(np.float64) * 2.0**-53 * 2.0 - 1.0) * scale
    b = (re + 1j * im).reshape(npol, lmax + 1)
    b[:, :m] = 0.0
    return b


def sky_alm(cfg_seed, nfreq, lmax, npol=4):
    """Band-limited Gaussian sky of SURVEY 8d: ``a_lm [nfreq, npol, lmax+1, lmax+1]`` with ``C_l = (l+1)^-2``,
    seed ``4000 + cfg``; real m = 0, E/B start at l = 2."""
    rng = np.random.default_rng(4000 + cfg_seed)
    amp = (np.arange(lmax + 1) + 1.0) ** -1.0
    a = (rng.standard_normal((nfreq, npol, lmax + 1, lmax + 1)) + 1j * rng.standard_normal((nfreq, npol, lmax + 1, lmax + 1))) * amp[None, None, :, None]
    a *= np.tril(np.ones((lmax + 1, lmax + 1)))[None, None]
    a[..., 0] = a[..., 0].real
    if npol == 4:
        a[:, 1:3, :2] = 0
    return a


def npairs_of(ncyl, nfeed_cyl):
    """Unique baselines (autos included) of ncyl x nfeed_cyl x 2-pol regular grid."""
    return 4 * (nfeed_cyl + (ncyl - 1) * (2 * nfeed_cyl - 1)) - 1


# cfg -> shapes of BASELINE.json's configs as made concrete in SURVEY.md section 8d
CONFIGS = {
    1: dict(ncyl=1, nfeed_cyl=8, nfreq=4, nra=127, lmax=63, nside=32),
    2: dict(ncyl=2, nfeed_cyl=16, nfreq=64, nra=512, lmax=256, nside=128),
    3: dict(ncyl=2, nfeed_cyl=32, nfreq=256, nra=1024, lmax=512, nside=256),
    4: dict(ncyl=2, nfeed_cyl=64, nfreq=512, nra=2048, lmax=1024, nside=512),
    5: dict(ncyl=2, nfeed_cyl=64, nfreq=1024, nra=2047, lmax=1023, nside=512),
}


def frequencies(nfreq):
    return np.linspace(400.0, 800.0, nfreq, endpoint=False)


def sidereal_inputs(cfg_seed, nfreq, npairs, nra, zero_frac=0.01):
    """``vis ~ CN(0,1)`` complex64 and ``weight ~ U(0.5,1.5)`` float32 with exact zeros."""
    rng = np.random.default_rng(1000 + cfg_seed)
    vis = (rng.standard_normal((nfreq, npairs, nra), dtype=np.float32) + 1j * rng.standard_normal((nfreq, npairs, nra), dtype=np.float32)).astype(np.complex64)
    rng = np.random.default_rng(2000 + cfg_seed)
    w = rng.uniform(0.5, 1.5, (nfreq, npairs, nra)).astype(np.float32)
    w[rng.uniform(size=w.shape) < zero_frac] = 0.0
    return vis, w


# ------------------------------------------------------------------ physically structured tiles ("beam screens")
# Twin of csrc/beamscreen.hip (BeamScreenProvider): same model, NumPy arithmetic, the oracle's own SHT.  Not a
# restatement of reference code -- driftscan's beam transfers are third-party inputs to the path; this is
# the checker of the library's structured INPUT generator.
def screen_coeffs(seed):
    """The 16 plane waves of the two polarisation types' gain / leakage screens: arrays ``[2, 2, 4]``."""
    ka = np.zeros((2, 2, 4), dtype=np.int64)
    kb = np.zeros((2, 2, 4), dtype=np.int64)
    cr = np.zeros((2, 2, 4))
    ci = np.zeros((2, 2, 4))
    for t in range(2):
        for w in range(2):
            for k in range(4):
                key = _mix64_int((seed + 0x9E3779B97F4A7C15 * (1 + k + 4 * (w + 2 * t))) & 0xFFFFFFFFFFFFFFFF)
                ka[t, w, k] = _mix64_int((key + 1) & 0xFFFFFFFFFFFFFFFF) % 7 - 3
                kb[t, w, k] = _mix64_int((key + 2) & 0xFFFFFFFFFFFFFFFF) % 7 - 3
                cr[t, w, k] = (2.0 * ((_mix64_int((key + 3) & 0xFFFFFFFFFFFFFFFF) >> 11) * 2.0**-53) - 1.0) / 4
                ci[t, w, k] = (2.0 * ((_mix64_int((key + 4) & 0xFFFFFFFFFFFFFFFF) >> 11) * 2.0**-53) - 1.0) / 4
    return ka, kb, cr, ci


def screen_jones(model, wavelength, sigma_e):
    """Per pixel: ``up`` mask, direction cosines (east, north) and the Jones vectors ``e [2 types, 2 comps, npix]``."""
    from . import sht

    th, ph = sht.pix_angles(model["nside"])
    nx, ny, nz = np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)
    lat = np.deg2rad(model["latitude"])
    cz = nx * np.cos(lat) + nz * np.sin(lat)
    ce = ny
    cn = -nx * np.sin(lat) + nz * np.cos(lat)
    up = cz > 0
    env = np.where(up, np.sqrt(np.where(up, cz, 0.0)) * np.exp(-0.5 * ce**2 / sigma_e**2) * np.exp(-0.5 * cn**2 / model["sigma_n"] ** 2), 0.0)
    ka, kb, cr, ci = screen_coeffs(model["seed"])
    e = np.zeros((2, 2, th.size), dtype=np.complex128)
    for t in range(2):
        scr = [sum((cr[t, w, k] + 1j * ci[t, w, k]) * np.exp(1j * np.pi * (ka[t, w, k] * ce + kb[t, w, k] * cn)) for k in range(4)) for w in range(2)]
        g = env * (1.0 + model["eps_gain"] * scr[0])
        e[t, t] = g
        e[t, 1 - t] = g * (model["eps_leak"] * scr[1])
    return up, ce, cn, e


def screen_response(model, wavelength, sigma_e, s, npol=4):
    """Complex response maps ``[npol, npix]`` (A_I, A_Q, A_U, A_V) of unique pair ``s``."""
    up, ce, cn, e = screen_jones(model, wavelength, sigma_e)
    a, b = int(model["pol_a"][s]), int(model["pol_b"][s])
    ph = np.exp(2j * np.pi * (model["sep_e"][s] * ce + model["sep_n"][s] * cn) / wavelength)
    c = lambda i, j: np.conj(e[a, i]) * e[b, j] * ph  # noqa: E731
    A = np.stack([0.5 * (c(0, 0) + c(1, 1)), 0.5 * (c(0, 0) - c(1, 1)), 0.5 * (c(0, 1) + c(1, 0)), 0.5j * (c(1, 0) - c(0, 1))])
    return np.where(up, A, 0.0)[:npol]


def screen_tile(model, freq_mhz, m, lmax, npol=4, sigma_e_600=None):
    """Tile ``[2, npairs, npol, lmax+1]`` of the beam-screen model at one frequency (MHz) and m."""
    from . import sht

    wavelength = 299.792458 / float(freq_mhz)
    sigma_e = (model["sigma_e"] if sigma_e_600 is None else sigma_e_600) * wavelength / (299.792458 / 600.0)
    npairs = len(model["sep_e"])
    out = np.zeros((2, npairs, npol, lmax + 1), dtype=np.complex128)
    for s in range(npairs):
        A = screen_response(model, wavelength, sigma_e, s, npol)
        ar = sht.map2alm(A.real, lmax, niter=0)  # [npol, l, m]
        ai = sht.map2alm(A.imag, lmax, niter=0)
        out[0, s] = np.conj(ar[:, :, m]) + 1j * np.conj(ai[:, :, m])
        if m > 0:
            out[1, s] = np.conj(ar[:, :, m]) - 1j * np.conj(ai[:, :, m])
    out[..., :m] = 0.0
    return out

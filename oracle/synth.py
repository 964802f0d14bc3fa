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

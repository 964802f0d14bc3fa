"""Oracle SHT pinned by mathematical identities (healpy/cora are absent: parity unpinned upstream)."""

import numpy as np
import pytest

from oracle import sht


def _rand_alm(rng, lmax, npol=4, mmax=None):
    mmax = lmax if mmax is None else mmax
    a = np.zeros((npol, lmax + 1, lmax + 1), dtype=np.complex128)
    for l in range(lmax + 1):
        for m in range(min(l, mmax) + 1):
            a[:, l, m] = rng.standard_normal(npol) + (1j * rng.standard_normal(npol) if m > 0 else 0)
    if npol == 4:
        a[1:3, :2] = 0  # E, B start at l = 2
    return a


def test_ring_geometry():
    for nside in (1, 2, 4, 8, 16):
        z, nphi, phi0, start = sht.ring_info(nside)
        assert z.size == 4 * nside - 1 and nphi.sum() == 12 * nside**2
        assert np.all(np.diff(z) < 0) and np.allclose(z, -z[::-1], atol=1e-15)
        assert np.array_equal(start, np.concatenate([[0], np.cumsum(nphi)[:-1]]))
        assert nphi[0] == 4 and nphi[nside - 1] == 4 * nside and nphi[-1] == 4
        # equal-area: the mean of z and of z^2 over pixel centres of the sphere are 0 and ~1/3
        zz = np.repeat(z, nphi)
        assert abs(zz.mean()) < 1e-15 and abs((zz**2).mean() - 1 / 3) < 0.35 / nside**2 + 1e-12


def test_lambda_against_scipy():
    from scipy.special import sph_harm_y

    th = np.linspace(0.05, 3.1, 11)
    for l, m in ((0, 0), (1, 0), (1, 1), (7, 3), (64, 64), (120, 5), (300, 200)):
        lam = sht.lambda_lm(l, m, np.cos(th))[l]
        ref = sph_harm_y(l, m, th, 0.0).real
        np.testing.assert_allclose(lam, ref, rtol=0, atol=1e-12 * np.abs(ref).max())  # recurrence depth ~l: O(l eps)


def test_lambda_high_m_near_pole_no_garbage():
    z, *_ = sht.ring_info(512)
    lam = sht.lambda_lm(1024, 1000, z[:3])
    assert np.all(np.isfinite(lam)) and np.abs(lam).max() < 1e-200


def test_spin2_convention_against_edth_and_closed_forms():
    th = np.linspace(0.2, 2.9, 9)
    c, s = np.cos(th), np.sin(th)
    p2, m2 = sht.spin2_Y_edth(2, 2, th)
    np.testing.assert_allclose(p2, np.sqrt(5 / np.pi) / 8 * (1 - c) ** 2, atol=1e-9)
    np.testing.assert_allclose(m2, np.sqrt(5 / np.pi) / 8 * (1 + c) ** 2, atol=1e-9)
    p2, m2 = sht.spin2_Y_edth(2, 1, th)
    # odd m carries the Condon-Shortley sign of Y_lm through edth
    np.testing.assert_allclose(p2, -np.sqrt(5 / np.pi) / 4 * s * (1 - c), atol=1e-9)
    p2, m2 = sht.spin2_Y_edth(2, 0, th)
    np.testing.assert_allclose(p2, 0.75 * np.sqrt(5 / (6 * np.pi)) * s**2, atol=1e-9)
    for l, m in ((2, 0), (2, 1), (3, 1), (5, 4), (9, 9), (8, 0)):
        p2, m2 = sht.spin2_Y_edth(l, m, th)
        lam = sht.lambda_lm(l, m, c)
        F1, F2 = sht._spin_F(l, m, c, lam)
        np.testing.assert_allclose(F1[l], (p2 + m2) / 2, atol=2e-9)
        np.testing.assert_allclose(F2[l], (p2 - m2) / 2, atol=2e-9)


def test_monopole_and_single_modes():
    nside = 4
    a = np.zeros((1, 4, 4), complex)
    a[0, 0, 0] = np.sqrt(4 * np.pi)
    np.testing.assert_allclose(sht.alm2map(a, nside)[0], 1.0, atol=1e-14)
    th, ph = sht.pix_angles(nside)
    a[:] = 0
    a[0, 1, 1] = 1.0  # real field: a_11 Y_11 + a_1-1 Y_1-1 = 2 Re(Y_11) = -2 sqrt(3/8pi) sin cos(phi)
    np.testing.assert_allclose(sht.alm2map(a, nside)[0], -2 * np.sqrt(3 / (8 * np.pi)) * np.sin(th) * np.cos(ph), atol=1e-14)
    # pure E_20: Q = -E * (3/4) sqrt(5/6pi) sin^2, U = 0
    a4 = np.zeros((4, 4, 4), complex)
    a4[1, 2, 0] = 1.0
    mp = sht.alm2map(a4, nside)
    np.testing.assert_allclose(mp[1], -0.75 * np.sqrt(5 / (6 * np.pi)) * np.sin(th) ** 2, atol=1e-14)
    np.testing.assert_allclose(mp[[0, 2, 3]], 0.0, atol=1e-15)
    # pure B_20 -> pure U with the same pattern
    a4[:] = 0
    a4[2, 2, 0] = 1.0
    mp = sht.alm2map(a4, nside)
    np.testing.assert_allclose(mp[2], -0.75 * np.sqrt(5 / (6 * np.pi)) * np.sin(th) ** 2, atol=1e-14)
    np.testing.assert_allclose(mp[1], 0.0, atol=1e-15)


@pytest.mark.parametrize("nside,lmax", [(2, 5), (4, 9), (8, 12)])
def test_ring_synthesis_equals_definition(nside, lmax):
    rng = np.random.default_rng(lmax)
    a = _rand_alm(rng, lmax)
    mp = sht.alm2map(a, nside)
    np.testing.assert_allclose(mp[0], sht.alm2map_direct(a[0], nside), atol=1e-12)
    np.testing.assert_allclose(mp[3], sht.alm2map_direct(a[3], nside), atol=1e-12)
    Q, U = sht.alm2map_direct((a[1], a[2]), nside, spin_pair=True)
    np.testing.assert_allclose(mp[1], Q, atol=1e-12)
    np.testing.assert_allclose(mp[2], U, atol=1e-12)


def test_roundtrip_bandlimited():
    rng = np.random.default_rng(3)
    nside, lmax = 8, 12  # nside >= lmax/2: the quadrature converges under iteration
    a = _rand_alm(rng, lmax)
    mp = sht.alm2map(a, nside)
    err0 = np.abs(sht.map2alm(mp, lmax, niter=0) - a).max()
    err3 = np.abs(sht.map2alm(mp, lmax, niter=3) - a).max()
    assert err0 < 0.1 and err3 < 2e-4 and err3 < err0 / 50
    back = sht.map2alm(mp, lmax, niter=12)
    assert np.abs(back - a).max() < 1e-8


def test_analysis_is_adjoint_of_synthesis():
    """<y, S a> = <A y, a> up to the pixel weight (real inner products): pins map2alm to alm2map."""
    rng = np.random.default_rng(4)
    nside, lmax = 4, 7
    a = _rand_alm(rng, lmax)
    y = rng.standard_normal((4, 12 * nside**2))
    Sa = sht.alm2map(a, nside)
    Ay = sht.legendre_analysis(sht.ring_analysis(y, nside, lmax), nside, lmax)
    w = 4 * np.pi / (12 * nside**2)
    m = np.arange(lmax + 1)
    fac = np.where(m == 0, 1.0, 2.0)
    lhs = (y * Sa).sum() * w
    rhs = (fac * (np.conj(Ay) * a).real).sum()
    assert abs(lhs - rhs) < 1e-12 * abs(lhs)


def test_cora_shaped_wrappers():
    rng = np.random.default_rng(5)
    a = np.stack([_rand_alm(rng, 6), _rand_alm(rng, 6)])
    mp = sht.sphtrans_inv_sky(a, 4)
    assert mp.shape == (2, 4, 192)
    back = sht.sphtrans_sky(mp, 6, niter=10)
    assert back.shape == a.shape and np.abs(back - a).max() < 1e-6


# ----------------------------------------------------------------------------------------------------------------
# Independent witnesses of the conventions that cora/healpy would fix (both absent: a10 stays "parity unpinned").
# Each test names the convention it pins.


def test_pixel_centres_literal_tables_nside_1_2():
    """RING pixel centres for nside 1 and 2 written out as literal (z, phi/pi) rationals -- the published HEALPix
    values (Gorski et al. 2005 eqs. 4-9 with the library's ring start: belt rings with odd ``i + nside`` start at
    phi = 0, the others half a pixel in).  Pins: ring order north -> south, pixel order west -> east inside a ring,
    ring latitudes, the half-pixel stagger."""
    from fractions import Fraction as F

    tables = {
        1: [(F(2, 3), [F(1, 4), F(3, 4), F(5, 4), F(7, 4)]), (F(0), [F(0), F(1, 2), F(1), F(3, 2)]), (F(-2, 3), [F(1, 4), F(3, 4), F(5, 4), F(7, 4)])],
        2: [
            (F(11, 12), [F(2 * k + 1, 4) for k in range(4)]),
            (F(2, 3), [F(2 * k + 1, 8) for k in range(8)]),
            (F(1, 3), [F(k, 4) for k in range(8)]),
            (F(0), [F(2 * k + 1, 8) for k in range(8)]),
            (F(-1, 3), [F(k, 4) for k in range(8)]),
            (F(-2, 3), [F(2 * k + 1, 8) for k in range(8)]),
            (F(-11, 12), [F(2 * k + 1, 4) for k in range(4)]),
        ],
    }
    for nside, rings in tables.items():
        zs = np.array([float(z) for z, phis in rings for _ in phis])
        ph = np.array([float(p) * np.pi for _, phis in rings for p in phis])
        assert zs.size == 12 * nside**2
        th, phi = sht.pix_angles(nside)
        np.testing.assert_allclose(np.cos(th), zs, atol=1e-15)
        np.testing.assert_allclose(phi, ph, atol=1e-15)
    # the values quoted in healpy's documentation of pix2ang(nside=1/2): theta of the first rings
    th1, _ = sht.pix_angles(1)
    np.testing.assert_allclose(th1[[0, 4, 8]], [0.84106867, 1.57079633, 2.30052398], atol=5e-9)
    th2, _ = sht.pix_angles(2)
    np.testing.assert_allclose(th2[[0, 4, 12, 20]], [0.41113786, 0.84106867, 1.23095942, 1.57079633], atol=5e-9)


@pytest.mark.parametrize("nside", [1, 2, 3, 4, 8, 16, 32])
def test_point_in_pixel_rule_agrees_with_ring_tables(nside):
    """The boundary-equation ``ang2pix`` (independent of ``ring_info``) maps every tabulated centre to its own
    index, and random directions fall into equal-area pixels.  Pins the pixel numbering against the published
    point-in-pixel rule; ``lonlat=True`` is the form the reference calls (beamform.py:1708,1760)."""
    th, ph = sht.pix_angles(nside)
    npix = 12 * nside**2
    np.testing.assert_array_equal(sht.ang2pix_ring(nside, th, ph), np.arange(npix))
    np.testing.assert_array_equal(sht.ang2pix_lonlat(nside, np.degrees(ph), 90.0 - np.degrees(th)), np.arange(npix))
    if nside <= 8:
        rng = np.random.default_rng(nside)
        n = 400 * npix
        z = rng.uniform(-1, 1, n)
        cnt = np.bincount(sht.ang2pix_ring(nside, np.arccos(z), rng.uniform(0, 2 * np.pi, n)), minlength=npix)
        assert cnt.size == npix and np.abs(cnt - 400).max() < 6 * 20  # Poisson sigma = 20


@pytest.mark.parametrize("nside,lmax", [(1, 2), (2, 5), (4, 8), (8, 16)])
def test_scalar_synthesis_whole_map_against_scipy(nside, lmax):
    """The whole scalar map from SciPy's Y_lm at the pixel centres: T = sum_l a_l0 Y_l0 + 2 Re sum_{m>0} a_lm Y_lm.
    Pins, independently of the oracle's recurrences: the a_lm index order [l, m], the Condon-Shortley phase, the
    real-field (m >= 0 only) convention and the orthonormal Y_lm normalisation healpy documents."""
    from scipy.special import sph_harm_y

    rng = np.random.default_rng(100 * nside + lmax)
    a = _rand_alm(rng, lmax, npol=1)
    th, ph = sht.pix_angles(nside)
    ref = np.zeros(th.size)
    for l in range(lmax + 1):
        ref += (a[0, l, 0] * sph_harm_y(l, 0, th, ph)).real
        for m in range(1, l + 1):
            ref += 2.0 * (a[0, l, m] * sph_harm_y(l, m, th, ph)).real
    np.testing.assert_allclose(sht.alm2map(a, nside)[0], ref, atol=1e-12 * np.abs(ref).max())
    # V (slot 3 of a four-polarisation transform) is the same scalar transform
    a4 = _rand_alm(rng, lmax)
    a4[3] = a[0]
    np.testing.assert_allclose(sht.alm2map(a4, nside)[3], ref, atol=1e-12 * np.abs(ref).max())


def test_spin2_whole_map_against_scipy_derivatives():
    """Q, U of a whole map from second derivatives of SciPy's Y_lm: with P = Q + iU = sum_lm -(E + iB)_lm (+2Y_lm) and
    +2Y_lm = edth edth Y_lm / sqrt((l+2)!/(l-2)!) (Newman-Penrose; Zaldarriaga & Seljak 1997 eq. 6 -- the convention
    HEALPix documents).  edth is applied by finite differences in theta to SciPy's functions, so nothing of the oracle's
    Legendre/KKS code is involved.  Pins the (E, B) -> (Q, U) signs and the HEALPix polarisation convention."""
    from scipy.special import sph_harm_y

    nside, lmax = 2, 4
    rng = np.random.default_rng(77)
    a = _rand_alm(rng, lmax)
    th, ph = sht.pix_angles(nside)

    P = np.zeros(th.size, complex)
    h = 1e-3
    for l in range(2, lmax + 1):
        norm = np.sqrt(float((l - 1) * l * (l + 1) * (l + 2)))
        for m in range(-l, l + 1):
            am = abs(m)
            E = a[1, l, am] if m >= 0 else (-1) ** am * np.conj(a[1, l, am])
            B = a[2, l, am] if m >= 0 else (-1) ** am * np.conj(a[2, l, am])

            def Y(t):
                return sph_harm_y(l, m, t, 0.0)

            def spin1(t):  # edth Y (spin 0 -> 1): -(dY/dtheta - m/sin Y)   [i d/dphi -> -m]
                return -((Y(t + h) - Y(t - h)) / (2 * h) - m / np.sin(t) * Y(t))

            def spin2(t):  # edth (spin 1 -> 2): -(d/dtheta - m/sin - cot) f
                f = spin1
                return -((f(t + h) - f(t - h)) / (2 * h) - m / np.sin(t) * f(t) - np.cos(t) / np.sin(t) * f(t))

            P += -(E + 1j * B) * spin2(th) / norm * np.exp(1j * m * ph)
    mp = sht.alm2map(a, nside)
    scale = np.abs(P).max()
    np.testing.assert_allclose(mp[1], P.real, atol=3e-5 * scale)  # finite differences: O(h^2)
    np.testing.assert_allclose(mp[2], P.imag, atol=3e-5 * scale)

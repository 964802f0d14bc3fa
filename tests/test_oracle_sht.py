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

"""BASELINE-sized checks on the GPU: direct oracle comparisons where the oracle finishes in seconds,
size-independent properties otherwise (SURVEY.md 8d shapes: cfg 2 = 64 feeds/lmax 256, cfg 3 = 128 feeds/
lmax 512/nside 256, cfg 4 = 256 feeds/lmax 1024)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import sht as osht
from oracle import synth as osyn


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _tel(cfg, nfreq):
    from draco_amd.core.products import TransitTelescope

    c = osyn.CONFIGS[cfg]
    return TransitTelescope(osyn.frequencies(nfreq), lmax=c["lmax"], ncyl=c["ncyl"], nfeed_cyl=c["nfeed_cyl"])


def test_cfg4_tile_dirty_and_project_adjointness():
    """cfg-4 tile (1526 x 4100, 100 MB): <B a, v> == <a, B^H v> ties k_project to k_dirty at full size,
    and a unit-vector solve reads a B row back exactly."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(4, 1)
    assert tel.npairs == 763 and tel.lmax == 1024
    bt = SyntheticProvider(tel, seed=4004)
    # only m = 0..2 (three 100 MB tiles) to keep the pool small
    tel.mmax = 2
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, cache=True)
    gen = torch.Generator(device=ctx.device).manual_seed(1)
    v = torch.randn((3, 2, 1, 763), dtype=torch.complex128, device=ctx.device, generator=gen)
    ones = torch.ones((3, 2, 1, 763), dtype=torch.float64, device=ctx.device)
    a = torch.randn((1, 4, 3, 1025), dtype=torch.complex128, device=ctx.device, generator=gen)
    for m in range(3):
        a[:, :, m, :m] = 0
    Bhv = eng.solve("dirty", v, ones, [0], 2)  # [1, 4, 3, 1025]
    Ba = eng.project(a, [0], 2)                # [3, 2, 1, 763]
    lhs = (Ba.conj() * v).sum()
    rhs = (a.conj() * Bhv).sum()
    assert abs((lhs - rhs).item()) < 1e-12 * abs(lhs.item())
    e = torch.zeros_like(v)
    e[1, 1, 0, 700] = 1.0
    row = eng.solve("dirty", e, ones, [0], 2)[0, :, 1, :].cpu().numpy()
    ref = np.conj(osyn.beam_tile(4004, 1, 0, 763, 4, 1024).reshape(1526, 4, 1025)[763 + 700])
    assert np.array_equal(row, ref)


def test_cfg3_tile_wiener_normal_equations_and_ml_vs_oracle_cfg2():
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.core.products import SyntheticProvider

    rng = np.random.default_rng(9)
    # Wiener at the cfg-3 tile size 758 x 2052 (and a trimmed high-m tile)
    tel = _tel(3, 1)
    bt = SyntheticProvider(tel, seed=33)
    w = WienerMapMaker(prior_amp=1.5, prior_tilt=0.75)
    w.setup(bt)
    for m in (0, 400):
        v = rng.standard_normal((2, 379)) + 1j * rng.standard_normal((2, 379))
        Ni = rng.uniform(0.5, 1.5, (2, 379)) * 20
        Ni[rng.uniform(size=Ni.shape) < 0.05] = 0
        a = w._solve_m(m, 0, v, Ni)
        B = osyn.beam_tile(33, m, 0, 379, 4, 512)[..., m:].reshape(758, -1)
        S = omm.wiener_prior(512, m, 1.5, 0.75)
        x = a[:, m:].reshape(-1)
        lhs = x / S + B.conj().T @ (Ni.reshape(-1) * (B @ x))
        rhs = B.conj().T @ (Ni.reshape(-1) * v.reshape(-1))
        assert _rel(lhs, rhs) < 1e-10, m
        assert np.all(a[:, :m] == 0)
    # ML at the cfg-2 tile size 374 x 1028 against the oracle's SVD
    tel2 = _tel(2, 1)
    bt2 = SyntheticProvider(tel2, seed=22)
    ml = MaximumLikelihoodMapMaker()
    ml.setup(bt2)
    for m in (0, 200):
        v = rng.standard_normal((2, 187)) + 1j * rng.standard_normal((2, 187))
        Ni = rng.uniform(0.5, 1.5, (2, 187)) * 20
        Ni[rng.uniform(size=Ni.shape) < 0.05] = 0
        a = ml._solve_m(m, 0, v, Ni)
        ref = omm.ml_solve(osyn.beam_tile(22, m, 0, 187, 4, 256), v, Ni)
        assert _rel(a, ref) < 1e-8, m


def test_cfg3_alm2map_against_oracle():
    """lmax = mmax = 512, nside = 256 (cfg 3's SHT), one frequency, all four polarisations."""
    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    lmax, nside = 512, 256
    rng = np.random.default_rng(5)
    alm = np.zeros((1, 4, lmax + 1, lmax + 1), dtype=np.complex128)
    l = np.arange(lmax + 1)
    amp = (l + 1.0) ** -1.0
    re = rng.standard_normal(alm.shape) * amp[None, None, :, None]
    im = rng.standard_normal(alm.shape) * amp[None, None, :, None]
    alm[:] = np.tril(np.ones((lmax + 1, lmax + 1)))[None, None] * (re + 1j * im)
    alm[..., 0] = alm[..., 0].real
    alm[:, 1:3, :2] = 0
    a_dev = ctx.to_device(np.ascontiguousarray(alm.transpose(0, 1, 3, 2)), np.complex128)
    out = ctx.empty((1, 4, 12 * nside * nside), np.float64)
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(a_dev), 1, 4, lmax, lmax, nside, ptr(out)))
    ref = osht.sphtrans_inv_sky(alm, nside)
    assert _rel(out.cpu().numpy(), ref) < 1e-10

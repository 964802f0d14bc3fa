"""`BaseMapMaker.process_many`: D sidereal days against ONE pass over the beam transfers.

The reference's pipeline calls ``process`` once per item (``doc/tutorial.rst:110-120``) and its loop reads every
``beam_m`` again each time (``mapmaker.py:79-94,160-162``).  Here a slab of B is made resident once for all D days
(one PCIe crossing for host-fed providers) and ``DirtyMapMaker`` shares every tile READ between up to eight days
(``dmm_dirty_run_multi``: eight accumulators per column).  Checked: every day's a_lm and map equal its own
single-day ``process`` BIT FOR BIT -- all groupings of the day count (8 / 4 / 2 / 1 kernels), both storage types of B,
full and packed layouts through the C ABI, a host-streamed provider with many slabs --, against the oracle, and that
the beam transfers crossed PCIe once.
"""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import synth as osyn


def _days(D, nfreq=5, lmax=37, nfeed_cyl=6, seed=91):
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider, TransitTelescope

    tel = TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, ncyl=1, nfeed_cyl=nfeed_cyl)
    bt = SyntheticProvider(tel, seed=seed)
    shape = (lmax + 1, 2, nfreq, tel.npairs)
    days = []
    for d in range(D):
        rng = np.random.default_rng(seed + 17 * d)
        mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
        mm.vis[:] = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        w = rng.uniform(0.5, 1.5, shape)
        w[rng.uniform(size=shape) < 0.05] = 0
        mm.weight[:] = w
        days.append(mm)
    return tel, bt, days


@pytest.mark.parametrize("b_dtype", ["complex128", "complex64"])
@pytest.mark.parametrize("D", [1, 2, 3, 7, 8, 13, 16])
def test_dirty_days_share_the_tile_reads_bit_identically(D, b_dtype):
    from draco_amd.analysis.mapmaker import DirtyMapMaker

    tel, bt, days = _days(D)
    t = DirtyMapMaker(nside=16, b_dtype=b_dtype)
    t.setup(bt)
    single = [t.process(mm) for mm in days]
    many = t.process_many(days)
    assert len(many) == D
    for d in range(D):
        np.testing.assert_array_equal(many[d].map[:], single[d].map[:])
    alms = t.make_alm_many(days)
    for d in (0, D - 1):
        np.testing.assert_array_equal(alms[d].cpu().numpy(), t.make_alm(days[d]).cpu().numpy())
    if b_dtype == "complex128":  # and the oracle, on one day
        d = D - 1
        ref = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(91, m, f, tel.npairs, 4, tel.lmax), days[d].vis[:], days[d].weight[:], tel.lmax, tel.mmax, list(range(tel.nfreq)))
        got = t.alm_square(alms[d])
        assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()


def test_host_streamed_b_crosses_pcie_once_for_all_days():
    """Pinned store, two buffers of ~1.4 frequencies each (many slabs): the D-day pass uploads exactly the bytes ONE
    day uploads, and every day still equals its single-day pass."""
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker, WienerMapMaker
    from draco_amd.core.products import PackedStoreProvider
    from draco_amd.device import Context

    D = 5
    tel, bt, days = _days(D, nfreq=6)
    store = PackedStoreProvider.from_provider(bt, Context.get(), np.complex128, pin=True)
    per_freq_bytes = store.per_freq * 16
    for cls, kw in ((DirtyMapMaker, {}), (WienerMapMaker, {"prior_amp": 1.5})):
        t = cls(nside=16, pool_bytes=int(2.8 * per_freq_bytes), **kw)
        t.setup(store)
        ref = [t.make_alm(mm).cpu().numpy() for mm in days]
        eng = t._get_engine()
        _solve.release_pools()
        f0 = eng.fills
        t.make_alm(days[0])
        one_day_bytes, one_day_fills = eng.last_b_bytes, eng.fills - f0
        assert one_day_fills >= 5
        _solve.release_pools()
        f0 = eng.fills
        got = t.make_alm_many(days)
        assert eng.fills - f0 == one_day_fills and eng.last_b_bytes == one_day_bytes  # one pass over B, not D
        for d in range(D):
            np.testing.assert_array_equal(got[d].cpu().numpy(), ref[d])
    _solve.release_pools()


def test_ml_days_and_argument_checks():
    from draco_amd.analysis.mapmaker import DirtyMapMaker, MaximumLikelihoodMapMaker
    from draco_amd.core import containers

    tel, bt, days = _days(3, nfreq=2, lmax=20, nfeed_cyl=4)
    t = MaximumLikelihoodMapMaker(nside=8)
    t.setup(bt)
    got = t.make_alm_many(days)
    for d in range(3):
        np.testing.assert_array_equal(got[d].cpu().numpy(), t.make_alm(days[d]).cpu().numpy())
    assert DirtyMapMaker().process_many([]) == []
    other = containers.MModes(mmax=tel.lmax, freq=tel.frequencies[:1], stack=tel.npairs)
    dm = DirtyMapMaker(nside=8)
    dm.setup(bt)
    with pytest.raises(ValueError, match="same frequencies"):
        dm.process_many([days[0], other])


@pytest.mark.parametrize("layout", ["packed", "full"])
def test_abi_multi_day_launch_and_its_argument_errors(layout):
    """`dmm_dirty_run_multi` straight through the C ABI on both tile layouts; shared or NULL day arrays are refused."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import Slab
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    lmax, nfreq, D = 21, 2, 6
    tel = TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, ncyl=1, nfeed_cyl=5)
    bt = SyntheticProvider(tel, seed=5)
    lay = _lib.DMM_B_PACKED if layout == "packed" else _lib.DMM_B_FULL
    ms = np.tile(np.arange(lmax + 1, dtype=np.int32), nfreq)
    fs = np.repeat(np.arange(nfreq, dtype=np.int32), lmax + 1)
    slab = Slab(ctx, bt, ms, fs, fs, _lib.DMM_C128, lay, nfreq, lmax + 1)
    gen = torch.Generator(device=ctx.device).manual_seed(3)
    shape = (lmax + 1, 2, nfreq, tel.npairs)
    mv = [torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen) for _ in range(D)]
    mw = [torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) for _ in range(D)]
    al = [torch.empty((nfreq, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device) for _ in range(D)]
    PA = C.c_void_p * D
    pv, pw, pa = PA(*[ptr(x) for x in mv]), PA(*[ptr(x) for x in mw]), PA(*[ptr(x) for x in al])
    _lib.check(_lib.lib.dmm_dirty_run_multi(slab.plan, ptr(slab.pool), pv, pw, pa, D))
    ref = torch.empty_like(al[0])
    for d in range(D):
        _lib.check(_lib.lib.dmm_dirty_run(slab.plan, ptr(slab.pool), ptr(mv[d]), ptr(mw[d]), ptr(ref)))
        assert torch.equal(al[d], ref)
    bad = PA(*[ptr(al[0])] * D)
    assert _lib.lib.dmm_dirty_run_multi(slab.plan, ptr(slab.pool), pv, pw, bad, D) == _lib.DMM_E_ARG
    assert b"share their alm" in _lib.lib.dmm_last_error()
    assert _lib.lib.dmm_dirty_run_multi(slab.plan, ptr(slab.pool), pv, pw, pa, 0) == _lib.DMM_E_ARG
    nul = PA(*([ptr(mv[0])] + [C.c_void_p(0)] * (D - 1)))
    assert _lib.lib.dmm_dirty_run_multi(slab.plan, ptr(slab.pool), nul, pw, pa, D) == _lib.DMM_E_ARG
    slab.close()


def test_many_baselines_take_smaller_day_groups_and_one_polarisation_falls_back():
    """cfg-4 baseline count (763 -> 1526 telescope rows): eight days' weights no longer fit the LDS side by side, the library
    groups the days by four; a telescope with ONE sky polarisation goes through the per-day path (the reference broadcasts
    its single solve into four slots, mapmaker.py:94).  Both: every day equals its own pass."""
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider, TransitTelescope

    for kw, D in (({"npairs": 763}, 9), ({"num_pol_sky": 1}, 3)):
        lmax, nfreq = 24, 2
        tel = TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, ncyl=1, nfeed_cyl=3, **kw)
        bt = SyntheticProvider(tel, seed=17)
        shape = (lmax + 1, 2, nfreq, tel.npairs)
        days = []
        for d in range(D):
            rng = np.random.default_rng(100 + d)
            mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
            mm.vis[:] = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
            mm.weight[:] = rng.uniform(0.5, 1.5, shape)
            days.append(mm)
        t = DirtyMapMaker(nside=8)
        t.setup(bt)
        many = t.process_many(days)
        for d in range(D):
            np.testing.assert_array_equal(many[d].map[:], t.process(days[d]).map[:])


def test_cfg3_tile_size_days_share_the_reads_bit_identically():
    """BASELINE cfg-3 tile size (379 baselines, lmax 512; 2 frequencies = 12.8 GB of tiles), 11 days = groups of 8 + 2 + 1:
    every day's a_lm equals the single-day launch, and the adjointness <B a, v> = <a, B^H v> ties day 10 to k_project."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context

    ctx = Context.get()
    c = osyn.CONFIGS[3]
    nfreq, D = 2, 11
    tel = TransitTelescope(osyn.frequencies(nfreq), lmax=c["lmax"], ncyl=c["ncyl"], nfeed_cyl=c["nfeed_cyl"])
    assert tel.npairs == 379 and tel.lmax == 512
    eng = SolveEngine(SyntheticProvider(tel, seed=3003), ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, cache=True)
    gen = torch.Generator(device=ctx.device).manual_seed(4)
    shape = (tel.lmax + 1, 2, nfreq, tel.npairs)
    mv = [torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen) for _ in range(D)]
    mw = [torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) for _ in range(D)]
    many = eng.solve_many("dirty", mv, mw, list(range(nfreq)), tel.lmax)
    for d in (0, 7, 8, 9, 10):
        assert torch.equal(many[d], eng.solve("dirty", mv[d], mw[d], list(range(nfreq)), tel.lmax)), d
    a = torch.randn(many[10].shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    for m in range(tel.lmax + 1):
        a[:, :, m, :m] = 0
    ones = torch.ones(shape, dtype=torch.float64, device=ctx.device)
    bhv = eng.solve_many("dirty", [mv[10]] * 2, [ones] * 2, list(range(nfreq)), tel.lmax)[1]
    ba = eng.project(a, list(range(nfreq)), tel.lmax)
    lhs, rhs = (ba.conj() * mv[10]).sum(), (a.conj() * bhv).sum()
    assert abs((lhs - rhs).item()) < 1e-12 * abs(lhs.item())
    del eng
    from draco_amd.analysis import _solve

    _solve.release_pools()

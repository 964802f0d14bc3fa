"""GPU parity: HEALPix SHT kernels and the tasks built on them vs the oracle.

float64 throughout; the kernels and the oracle share the algorithm (Legendre recurrence
per ring + Fourier sums per ring) but not the evaluation order: synthesis agrees to
~1e-13 of the map's scale, asserted 1e-11; analysis (quadrature + Jacobi refinements)
1e-10.  The oracle itself is pinned by identities only (healpy absent: parity unpinned).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import sht as osht
from oracle import stream as ostream
from oracle import synth as osyn


def _rand_alm(rng, nfreq, npol, lmax, mmax=None):
    mmax = lmax if mmax is None else mmax
    a = np.zeros((nfreq, npol, lmax + 1, lmax + 1), dtype=np.complex128)
    for l in range(lmax + 1):
        for m in range(min(l, mmax) + 1):
            a[:, :, l, m] = rng.standard_normal((nfreq, npol)) + (1j * rng.standard_normal((nfreq, npol)) if m > 0 else 0)
    if npol == 4:
        a[:, 1:3, :2] = 0
    return a


def _alm2map_gpu(alm_sq, nside, mmax=None):
    """alm_sq [nfreq, npol, lmax+1, lmax+1] (l, m) -> map via dmm_alm2map."""
    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    nfreq, npol, nl, _ = alm_sq.shape
    lmax = nl - 1
    mmax = lmax if mmax is None else mmax
    a_dev = ctx.to_device(np.ascontiguousarray(alm_sq[..., : mmax + 1].transpose(0, 1, 3, 2)), np.complex128)  # m-major
    out = ctx.empty((nfreq, npol, 12 * nside * nside), np.float64)
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(a_dev), nfreq, npol, lmax, mmax, nside, ptr(out)))
    return out.cpu().numpy()


def _map2alm_gpu(maps, lmax, mmax, niter):
    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    nfreq, npol, npix = maps.shape
    nside = int(round((npix // 12) ** 0.5))
    m_dev = ctx.to_device(maps, np.float64)
    out = ctx.empty((nfreq, npol, mmax + 1, lmax + 1), np.complex128)
    _lib.check(_lib.lib.dmm_map2alm(ctx.handle, ptr(m_dev), nfreq, npol, lmax, mmax, nside, niter, ptr(out)))
    a = out.cpu().numpy().transpose(0, 1, 3, 2)  # -> [f, pol, l, m]
    sq = np.zeros((nfreq, npol, lmax + 1, lmax + 1), np.complex128)
    sq[..., : mmax + 1] = a
    return sq


@pytest.mark.parametrize("nside,lmax,npol,mmax", [(1, 2, 4, None), (2, 5, 4, None), (4, 9, 1, None), (8, 20, 4, None), (16, 40, 4, 25), (32, 70, 4, None)])
def test_alm2map_vs_oracle(nside, lmax, npol, mmax):
    rng = np.random.default_rng(nside * 100 + lmax)
    alm = _rand_alm(rng, 2, npol, lmax, mmax)
    ref = osht.sphtrans_inv_sky(alm, nside)
    out = _alm2map_gpu(alm, nside, mmax)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()


def test_alm2map_definition_level():
    """Against the O(npix*lmax^2) definition, not just the ring-based oracle."""
    rng = np.random.default_rng(9)
    alm = _rand_alm(rng, 1, 4, 7)
    out = _alm2map_gpu(alm, 4)[0]
    np.testing.assert_allclose(out[0], osht.alm2map_direct(alm[0, 0], 4), atol=1e-12)
    Q, U = osht.alm2map_direct((alm[0, 1], alm[0, 2]), 4, spin_pair=True)
    np.testing.assert_allclose(out[1], Q, atol=1e-12)
    np.testing.assert_allclose(out[2], U, atol=1e-12)
    np.testing.assert_allclose(out[3], osht.alm2map_direct(alm[0, 3], 4), atol=1e-12)


def test_alm2map_extended_range_near_poles():
    """lmax, m large vs the ring's sin(theta): the scaled start and the ring skip must not lose power."""
    rng = np.random.default_rng(11)
    nside, lmax = 128, 300
    alm = _rand_alm(rng, 1, 4, lmax)
    alm[:, :, :200] *= 1e-3  # put the power at high l
    ref = osht.sphtrans_inv_sky(alm, nside)
    out = _alm2map_gpu(alm, nside)
    assert np.all(np.isfinite(out))
    assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("nside,lmax,npol,niter", [(4, 6, 4, 0), (8, 12, 4, 3), (8, 12, 1, 2), (16, 20, 4, 3)])
def test_map2alm_vs_oracle(nside, lmax, npol, niter):
    rng = np.random.default_rng(lmax)
    maps = rng.standard_normal((2, npol, 12 * nside * nside))
    ref = osht.sphtrans_sky(maps, lmax, niter)
    out = _map2alm_gpu(maps, lmax, lmax, niter)
    assert np.abs(out - ref).max() < 1e-10 * np.abs(ref).max()


def test_sht_roundtrip_larger():
    """Size-independent property at nside 128 / lmax 128: map2alm(alm2map(a)) -> a."""
    rng = np.random.default_rng(2)
    nside, lmax = 128, 128
    alm = _rand_alm(rng, 1, 4, lmax)
    mp = _alm2map_gpu(alm, nside)
    back = _map2alm_gpu(mp, lmax, lmax, 3)
    assert np.abs(back - alm).max() < 1e-6 * np.abs(alm).max()


@pytest.mark.parametrize("nside,lmax,npol", [(8, 23, 4), (64, 128, 4), (128, 200, 1), (256, 300, 4)])
def test_ring_fft_vs_direct_sums(nside, lmax, npol):
    """The FFT / Bluestein ring stages against the direct per-ring sums (sht_variant bit 2).

    Covers every Bluestein class (M = 256 ... 2048 at nside 256), rings shorter than mmax
    (folding modulo nphi) and both transforms; both evaluate the same sums, so 1e-12.
    """
    from draco_amd import _lib
    from draco_amd.device import Context

    ctx = Context.get()
    rng = np.random.default_rng(nside)
    alm = _rand_alm(rng, 2, npol, lmax)
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"sht_variant", 4))
        m_direct = _alm2map_gpu(alm, nside)
        a_direct = _map2alm_gpu(m_direct, lmax, lmax, 0)
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"sht_variant", 0))
    m_fft = _alm2map_gpu(alm, nside)
    a_fft = _map2alm_gpu(m_direct, lmax, lmax, 0)
    assert np.abs(m_fft - m_direct).max() < 1e-12 * np.abs(m_direct).max()
    assert np.abs(a_fft - a_direct).max() < 1e-12 * np.abs(a_direct).max()


@pytest.mark.parametrize("nside,lmax,nf", [(16, 40, 3), (64, 150, 5), (128, 256, 9), (256, 512, 2)])
def test_legendre_synthesis_kernels_agree(nside, lmax, nf):
    """The three MFMA forms of the Legendre synthesis (sht_variant bits 6 / 7: the first form of rounds 1-4; the pipelined
    kernel with 4 or 8 frequencies per block) evaluate the same sums -- 1e-12 of the map's scale apart (the pipelined form
    builds F1 / F2 with another association and reads lambda_{l-1} as 0 at the step a ring's scale reaches 1: < 2^-60 of
    the ring's values) -- with ragged frequency groups, chunks and ring tiles."""
    from draco_amd import _lib
    from draco_amd.device import Context

    ctx = Context.get()
    rng = np.random.default_rng(nside + nf)
    alm = _rand_alm(rng, nf, 4, lmax)
    maps = {}
    try:
        for variant in (0, 128, 64):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"sht_variant", variant))
            maps[variant] = _alm2map_gpu(alm, nside)
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"sht_variant", 0))
    ref = maps[64]
    scale = np.abs(ref).max()
    for variant in (0, 128, 64):
        assert np.abs(maps[variant] - ref).max() < 1e-12 * scale, (variant, np.abs(maps[variant] - ref).max() / scale)
    if nside <= 16:
        assert np.abs(ref - osht.sphtrans_inv_sky(alm, nside)).max() < 1e-10 * scale


def _tel(nfreq, lmax, ncyl=1, nfeed_cyl=3):
    from draco_amd.core.products import TransitTelescope

    return TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, ncyl=ncyl, nfeed_cyl=nfeed_cyl)


def test_dirty_mapmaker_process_to_map():
    """Full task: MModes -> Map, against oracle solve + oracle inverse SHT (mapmaker.py:35-118)."""
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    nfreq, lmax, nside = 3, 16, 8
    tel = _tel(nfreq, lmax)
    bt = SyntheticProvider(tel, seed=21)
    rng = np.random.default_rng(21)
    mv = rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape)
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    mm.attrs["tag"] = 1
    task = DirtyMapMaker(nside=nside)
    task.setup(bt)
    out = task.process(mm)
    assert isinstance(out, containers.Map) and out.map.shape == (nfreq, 4, 12 * nside**2) and out.map.dtype == np.float64
    assert np.array_equal(out.index_map["freq"]["centre"], tel.frequencies)
    alm = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(21, m, f, tel.npairs, 4, lmax), mv, mw, lmax, tel.mmax, list(range(nfreq)))
    ref = osht.sphtrans_inv_sky(alm, nside)
    rms = np.sqrt(((out.map[:] - ref) ** 2).mean() / (ref**2).mean())
    assert rms < 1e-11  # north star: maps within 1e-5 relative RMS
    # the inverse SHT beside the solves (default: side stream, the map carries the stream wait) or between them: same bits
    seq = DirtyMapMaker(nside=nside, overlap_sht=False)
    seq.setup(bt)
    assert np.array_equal(seq.process(mm).map[:], out.map[:])


def test_simulate_sidereal_vs_oracle():
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.synthesis.stream import SimulateSidereal

    nfreq, lmax, nside = 2, 10, 8
    tel = _tel(nfreq, lmax)
    bt = SyntheticProvider(tel, seed=31)
    rng = np.random.default_rng(31)
    alm = _rand_alm(rng, nfreq, 4, lmax)
    sky = osht.sphtrans_inv_sky(alm, nside)
    mp = containers.Map(nside=nside, freq=tel.frequencies)
    mp.map[:] = sky
    task = SimulateSidereal()
    task.setup(bt)
    ss = task.process(mp)
    assert isinstance(ss, containers.SiderealStream)
    assert ss.vis.shape == (nfreq, tel.npairs, 2 * lmax + 1) and ss.vis.dtype == np.complex64
    assert np.all(ss.weight[:] == 1.0) and ss.weight.dtype == np.float32
    assert len(ss.index_map["prod"]) == tel.nfeed * (tel.nfeed + 1) // 2 and len(ss.index_map["stack"]) == tel.npairs
    ref = ostream.simulate_sidereal(sky, lambda m, f: osyn.beam_tile(31, m, f, tel.npairs, 4, lmax), lmax, lmax, tel.npairs)
    assert np.abs(ss.vis[:] - ref).max() < 3e-7 * np.abs(ref).max()  # complex64 output
    bad = containers.Map(nside=nside, freq=tel.frequencies + 1.0)
    with pytest.raises(ValueError, match="Frequencies in map do not match"):
        task.process(bad)


def test_sim_to_map_chain_on_device():
    """SimulateSidereal -> MModeTransform -> DirtyMapMaker without leaving the GPU; checks against the oracle chain."""
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.synthesis.stream import SimulateSidereal
    from oracle import transform as otr

    nfreq, lmax, nside = 2, 12, 8
    tel = _tel(nfreq, lmax)
    bt = SyntheticProvider(tel, seed=41)
    rng = np.random.default_rng(41)
    sky = osht.sphtrans_inv_sky(_rand_alm(rng, nfreq, 4, lmax), nside)
    mp = containers.Map(nside=nside, freq=tel.frequencies)
    mp.map[:] = sky
    sim = SimulateSidereal()
    sim.setup(bt)
    tr = MModeTransform()
    tr.setup(bt)
    dm = DirtyMapMaker(nside=nside)
    dm.setup(bt)
    ss = sim.process(mp)
    mm = tr.process(ss)
    out = dm.process(mm)
    assert ss.vis.on_device and mm.vis.on_device and out.map.on_device
    beam = lambda m, f: osyn.beam_tile(41, m, f, tel.npairs, 4, lmax)  # noqa: E731
    vis = ostream.simulate_sidereal(sky, beam, lmax, lmax, tel.npairs)
    mv, mw = otr.mmode_transform(vis, np.ones(vis.shape, np.float32), mmax=lmax)
    alm = omm.solve_alm("dirty", beam, mv, mw, lmax, lmax, list(range(nfreq)))
    ref = osht.sphtrans_inv_sky(alm, nside)
    rms = np.sqrt(((out.map[:] - ref) ** 2).mean() / (ref**2).mean())
    assert rms < 1e-5  # the north star's tolerance; single-precision stream in the middle


def test_round_trip_recovers_the_sky():
    """Physics-level round trip: a band-limited sky -> SimulateSidereal -> MModeTransform -> maximum-likelihood
    map.  With more telescope degrees of freedom than sky modes at every m (43 baselines >= 4 (lmax + 1 - m)) and
    no noise, the pseudo-inverse is an exact inverse and the input map must come back (the stream in the
    middle is complex64, the beam matrices are random with condition ~1e2: asserted 1e-4 relative RMS)."""
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import ArrayProvider
    from draco_amd.synthesis.stream import SimulateSidereal

    nfreq, lmax, nside = 2, 9, 8
    tel = _tel(nfreq, lmax, ncyl=2, nfeed_cyl=4)
    assert tel.npairs >= 4 * (lmax + 1)  # at m = 0 only the +m half of the telescope carries data

    def beam(m, f):
        # like a real telescope's products, nothing lives in the "-0" half: the simulated stream drops it
        # (stream.py:131-133 keeps +m only at m = 0), so a beam with content there could not be inverted
        b = osyn.beam_tile(77, m, f, tel.npairs, 4, lmax).copy()
        if m == 0:
            b[1] = 0.0
        return b

    bt = ArrayProvider(tel, beam)
    rng = np.random.default_rng(77)
    sky = osht.sphtrans_inv_sky(_rand_alm(rng, nfreq, 4, lmax), nside)
    mp = containers.Map(nside=nside, freq=tel.frequencies)
    mp.map[:] = sky
    sim = SimulateSidereal()
    sim.setup(bt)
    tr = MModeTransform()
    tr.setup(bt)
    mm = tr.process(sim.process(mp))
    ml = MaximumLikelihoodMapMaker(nside=nside)
    ml.setup(bt)
    out = ml.process(mm).map[:]
    rms = np.sqrt(((out - sky) ** 2).mean() / (sky**2).mean())
    assert rms < 1e-4, rms
    # a Wiener filter with an uninformative prior converges to the same map
    wf = WienerMapMaker(nside=nside, prior_amp=1e6, prior_tilt=0.0)
    wf.setup(bt)
    out_w = wf.process(mm).map[:]
    rms_w = np.sqrt(((out_w - sky) ** 2).mean() / (sky**2).mean())
    assert rms_w < 1e-4, rms_w


def test_back_to_back_days_without_synchronisation():
    """Several `process` calls issued back to back, no synchronisation in between, the maps read only at the end: every
    day's last alm2map is still running on the side stream when the next day's transform and solves are enqueued (the
    map carries its own stream wait), allocations are recycled through the caching allocator meanwhile.  Every map must
    equal the one of a synchronised run."""
    import torch

    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    nfreq, lmax, nside = 6, 24, 16
    tel = _tel(nfreq, lmax)
    bt = SyntheticProvider(tel, seed=23)
    rng = np.random.default_rng(23)
    days = []
    for _ in range(4):
        mv = rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs))
        mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
        mm.vis[:] = mv
        mm.weight[:] = rng.uniform(0.5, 1.5, mv.shape)
        days.append(mm)
    per_f = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16
    task = DirtyMapMaker(nside=nside, pool_bytes=int(2.2 * per_f))  # two frequencies per slab: three slabs per day
    task.setup(bt)
    ref = []
    for mm in days:
        ref.append(task.process(mm).map[:].copy())
        torch.cuda.synchronize()
    for _ in range(3):
        outs = [task.process(mm) for mm in days]  # nothing reads a map, nothing synchronises
        for o, r in zip(outs, ref):
            assert np.array_equal(o.map[:], r)


def test_host_run_ahead_is_bounded_and_the_allocator_reaches_a_steady_state():
    """`process` never waits for its own day, but it may not get more than `days_in_flight` days ahead of the GPU: each
    day in flight holds its own a_lm and maps, and a pipeline that only issues work would otherwise fill the HBM until
    the caching allocator synchronises and frees its cache (380 instead of 256 ms per day over 20 days).
    After the first days nothing may be allocated from the device any more."""
    import torch

    from draco_amd.analysis import mapmaker
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    nfreq, lmax, nside = 8, 48, 64
    tel = _tel(nfreq, lmax)
    bt = SyntheticProvider(tel, seed=29)
    rng = np.random.default_rng(29)
    mv = rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs))
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = rng.uniform(0.5, 1.5, mv.shape)
    per_f = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16
    task = DirtyMapMaker(nside=nside, pool_bytes=int(2.2 * per_f))
    task.setup(bt)
    ref = task.process(mm).map[:].copy()
    for _ in range(8):
        task.process(mm)
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    last = None
    for _ in range(40):
        last = task.process(mm)
        q = mapmaker._IN_FLIGHT[0]
        assert len(q) <= task.days_in_flight
        # everything older than the days still listed has finished: the host is at most days_in_flight days ahead
    s1 = torch.cuda.memory_stats()
    assert s1["num_alloc_retries"] == s0["num_alloc_retries"]
    # steady state: what the 40 days reserved on top of the warm-up is less than ONE more day's a_lm + maps
    day_bytes = nfreq * 4 * ((lmax + 1) ** 2 * 16 + 12 * nside**2 * 8)
    assert s1["reserved_bytes.all.current"] - s0["reserved_bytes.all.current"] < day_bytes, (s0["reserved_bytes.all.current"], s1["reserved_bytes.all.current"])
    assert np.array_equal(last.map[:], ref)

"""BASELINE.json configurations run as *workloads* through the task classes on the GPU.

cfg 1 (16 feeds, 4 freq, 127 RA, lmax 63, nside 32): the whole SimulateSidereal -> MModeTransform -> DirtyMapMaker
chain against the full oracle chain, stage by stage and on the final maps (north star: maps within 1e-5 relative RMS).
cfg 2 (64 feeds, 64 freq, 512 RA, lmax 256, nside 128): MModeTransform + DirtyMapMaker through the task classes; the
m-modes against the oracle on the whole stream, a stratified sample of (m, f) solves against the oracle's
``dirty_solve``, EVERY (m, f) through an exact B-row read-back, one frequency's map against the oracle SHT.
cfg 3 tile size (758 x 2052): the ML solve's least-squares / minimum-norm property and a sampled oracle SVD.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import sht as osht
from oracle import stream as ostream
from oracle import synth as osyn
from oracle import transform as otr


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _rel_rms(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.sqrt((np.abs(a - b) ** 2).mean() / (np.abs(b) ** 2).mean())


def _tel(cfg, nfreq=None):
    from draco_amd.core.products import TransitTelescope

    c = osyn.CONFIGS[cfg]
    return TransitTelescope(osyn.frequencies(c["nfreq"] if nfreq is None else nfreq), lmax=c["lmax"], ncyl=c["ncyl"], nfeed_cyl=c["nfeed_cyl"])


def test_cfg1_full_chain_against_oracle_chain():
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.synthesis.stream import SimulateSidereal

    c = osyn.CONFIGS[1]
    tel = _tel(1)
    assert (tel.nfeed, tel.npairs, tel.nfreq, c["nra"], tel.lmax, c["nside"]) == (16, 31, 4, 127, 63, 32)
    seed = 3001
    bt = SyntheticProvider(tel, seed=seed)

    def beam(m, f):
        return osyn.beam_tile(seed, m, f, tel.npairs, 4, tel.lmax)

    # the sky of SURVEY 8d (band-limited Gaussian, C_l = (l+1)^-2, seed 4001) as a HEALPix map
    sky = osht.sphtrans_inv_sky(osyn.sky_alm(1, tel.nfreq, tel.lmax), c["nside"])
    map_ = containers.Map(nside=c["nside"], freq=tel.frequencies)
    map_.map[:] = sky

    sim = SimulateSidereal()
    sim.setup(bt)
    ss = sim.process(map_)
    assert ss.vis.shape == (4, 31, 127) and ss.vis.dtype == np.complex64
    ss_ref = ostream.simulate_sidereal(sky, beam, tel.lmax, tel.mmax, tel.npairs)
    assert _rel(ss.vis[:], ss_ref) < 1e-6  # complex64 stream; the map2alm Jacobi iterations are float64 on both sides

    tr = MModeTransform()
    tr.setup(bt)
    mm = tr.process(ss)
    assert mm.vis.shape == (64, 2, 4, 31) and bool(mm.attrs["oddra"])
    mv_ref, mw_ref = otr.mmode_transform(ss_ref, np.ones(ss_ref.shape, np.float32), mmax=tel.mmax)
    assert _rel(mm.vis[:], mv_ref) < 3e-6  # single-precision FFT of length 127 (Bluestein on the GPU, pocketfft in the oracle)
    np.testing.assert_allclose(mm.weight[:], mw_ref, rtol=2e-6)

    dm = DirtyMapMaker(nside=c["nside"])
    dm.setup(bt)
    out = dm.process(mm)
    assert out.map.shape == (4, 4, 12 * 32 * 32)
    alm_ref = omm.solve_alm("dirty", beam, mv_ref, mw_ref, tel.lmax, tel.mmax, list(range(tel.nfreq)))
    map_ref = osht.sphtrans_inv_sky(alm_ref, c["nside"])
    rms = _rel_rms(out.map[:], map_ref)
    assert rms < 1e-5, rms  # the north star's acceptance figure
    assert _rel(out.map[:], map_ref) < 1e-5
    # and with the oracle's m-modes as input the solve + SHT stage alone is at float64 level
    mm2 = containers.MModes(mmax=tel.mmax, freq=tel.frequencies, stack=tel.npairs, oddra=True)
    mm2.vis[:] = mv_ref
    mm2.weight[:] = mw_ref
    assert _rel(dm.process(mm2).map[:], map_ref) < 1e-11


def test_cfg2_workload_through_the_task_classes():
    import torch

    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    c = osyn.CONFIGS[2]
    tel = _tel(2)
    nfreq, npairs, nra, lmax = tel.nfreq, tel.npairs, c["nra"], tel.lmax
    assert (tel.nfeed, npairs, nfreq, nra, lmax, c["nside"]) == (64, 187, 64, 512, 256, 128)
    seed = 3002
    bt = SyntheticProvider(tel, seed=seed)
    vis, w = osyn.sidereal_inputs(2, nfreq, npairs, nra)
    ss = containers.SiderealStream(freq=tel.frequencies, ra=nra, stack=npairs)
    ss.vis[:] = vis
    ss.weight[:] = w
    tr = MModeTransform()
    tr.setup(bt)
    mm = tr.process(ss)
    mv_ref, mw_ref = otr.mmode_transform(vis, w, mmax=tel.mmax)
    mv = mm.vis[:]
    assert mv.shape == (257, 2, 64, 187)
    assert _rel(mv, mv_ref) < 2e-6
    np.testing.assert_allclose(mm.weight[:], mw_ref, rtol=2e-6)
    assert np.array_equal(mm.weight[:] == 0, mw_ref == 0)

    dm = DirtyMapMaker(nside=c["nside"])
    dm.setup(bt)
    alm = dm.alm_square(dm.make_alm(mm))  # [f, 4, l, m]
    assert alm.shape == (64, 4, 257, 257)
    # (a) stratified sample of (m, f): every m-decile x 4 frequency quartiles, against the oracle's dirty_solve on the
    #     m-modes the kernel consumed
    rng = np.random.default_rng(2)
    mw_dev = mm.weight[:]
    for dec in range(10):
        for q in range(4):
            m = int(rng.integers(dec * 257 // 10, (dec + 1) * 257 // 10))
            f = int(rng.integers(q * 16, (q + 1) * 16))
            a_ref = omm.dirty_solve(osyn.beam_tile(seed, m, f, npairs, 4, lmax), mv[m, :, f], mw_dev[m, :, f])
            assert _rel(alm[f, :, :, m], a_ref) < 1e-12, (m, f)
    iu = np.triu_indices(257, 1)
    assert not alm[:, :, iu[0], iu[1]].any()  # l < m stays exactly zero
    # (b) EVERY (m, f): a unit data vector on a row that differs per (m, f) reads that row of B back exactly
    rows = (np.arange(257)[:, None] * 31 + np.arange(64)[None, :] * 17) % (2 * npairs)  # [m, f]
    e = np.zeros((257, 2, 64, npairs), np.complex128)
    mi, fi = np.meshgrid(np.arange(257), np.arange(64), indexing="ij")
    e[mi, rows // npairs, fi, rows % npairs] = 1.0
    mm_e = containers.MModes(mmax=256, freq=tel.frequencies, stack=npairs)
    mm_e.vis[:] = e
    mm_e.weight[:] = 1.0
    back = dm.alm_square(dm.make_alm(mm_e))
    for m in range(257):
        for f in range(64):
            ref = np.conj(osyn.beam_row(seed, m, f, rows[m, f], npairs, 4, lmax))
            assert np.array_equal(back[f, :, :, m], ref), (m, f)
    # (c) the whole process() to maps; one frequency against the oracle SHT of the alm just checked
    out = dm.process(mm)
    assert out.map.shape == (64, 4, 12 * 128 * 128)
    f = 37
    ref_map = osht.sphtrans_inv_sky(alm[f : f + 1], c["nside"])
    assert _rel(out.map[f], ref_map[0]) < 1e-10
    del out, back
    torch.cuda.empty_cache()


def test_cfg3_tile_ml_pseudo_inverse_property_and_sampled_oracle_svd():
    """BASELINE config 3 names ML at 758 x 2052.  For each sampled tile: the defining property of the pseudo-inverse
    solution with the reference's cut removing nothing (exact fit + minimum norm when nsky_m >= ntel, normal equations
    otherwise), and the oracle's SVD solve (``pinv_svd`` restated, mapmaker.py:287-300) on the same tile."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    rng = np.random.default_rng(303)
    tel = _tel(3, 1)
    assert tel.npairs == 379 and tel.lmax == 512
    lmax = 512
    bt = SyntheticProvider(tel, seed=3003)
    ml = MaximumLikelihoodMapMaker()
    ml.setup(bt)
    for m in (0, 150, 320, 330, 480):  # telescope side, the near-square tiles around nsky_m ~ ntel (m ~ 323), sky side
        v = rng.standard_normal((2, 379)) + 1j * rng.standard_normal((2, 379))
        Ni = rng.uniform(0.5, 1.5, (2, 379)) * 20
        Ni[rng.uniform(size=Ni.shape) < 0.03] = 0
        full = osyn.beam_tile(3003, m, 0, 379, 4, lmax)
        B = full[..., m:].reshape(758, -1)
        nv, vv = Ni.reshape(-1), v.reshape(-1)
        a = ml._solve_m(m, 0, v, Ni)
        assert np.all(a[:, :m] == 0)
        x = a[:, m:].reshape(-1)
        Bt = np.sqrt(nv)[:, None] * B
        dv = np.sqrt(nv) * vv
        keep = nv > 0
        sv = np.linalg.svd(Bt[keep], compute_uv=False)
        nothing_cut = sv.min() > max(1e-3 * sv.max(), 1e-4)
        a_ref = omm.ml_solve(full, v, Ni)  # the oracle's SVD with the reference's cut
        assert _rel(a, a_ref) < 1e-8, (m, _rel(a, a_ref))
        if nothing_cut:
            r = Bt @ x - dv
            if B.shape[1] >= keep.sum():
                assert np.abs(r).max() < 1e-9 * np.abs(dv).max(), m
                y = np.linalg.lstsq(Bt[keep].conj().T, x, rcond=None)[0]
                assert np.abs(Bt[keep].conj().T @ y - x).max() < 1e-9 * np.abs(x).max(), m
            else:
                assert np.abs(Bt.conj().T @ r).max() < 1e-9 * np.abs(Bt.conj().T @ dv).max(), m
        # the same tile through the eigen path (certificate off)
        try:
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
            a_eig = ml._solve_m(m, 0, v, Ni)
        finally:
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        assert _rel(a_eig, a_ref) < 1e-8, (m, _rel(a_eig, a_ref))


def test_cfg4_slice_through_the_task_classes_and_frequency_shards():
    """BASELINE config 4 (256 feeds -> 763 baselines, 2048 RA, lmax 1024, nside 512; 512 frequencies over 8 GPUs) at its
    real tile / stream / map sizes on a 4-frequency slice: MModeTransform + DirtyMapMaker through the task classes,
    oracle on the whole stream and on sampled (m, f), exact read-back of EVERY (m, f), and the two halves of the slice
    processed as two ranks' frequency shards (``parallel.shard_freq``) giving bit-identical a_lm and maps -- what makes
    the 8-GPU form of this config correct by construction (no collective before the final gather)."""
    import torch

    from draco_amd import parallel
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    _solve.release_pools()
    c = osyn.CONFIGS[4]
    nfreq = 4
    tel = _tel(4, nfreq)
    npairs, nra, lmax = tel.npairs, c["nra"], tel.lmax
    assert (tel.nfeed, npairs, nra, lmax, c["nside"]) == (256, 763, 2048, 1024, 512)
    seed = 3004
    bt = SyntheticProvider(tel, seed=seed)
    vis, w = osyn.sidereal_inputs(4, nfreq, npairs, nra)
    ss = containers.SiderealStream(freq=tel.frequencies, ra=nra, stack=npairs)
    ss.vis[:] = vis
    ss.weight[:] = w
    tr = MModeTransform()
    tr.setup(bt)
    mm = tr.process(ss)
    mv_ref, mw_ref = otr.mmode_transform(vis, w, mmax=tel.mmax)
    mv = mm.vis[:]
    assert mv.shape == (1025, 2, 4, 763)
    assert _rel(mv, mv_ref) < 2e-6
    np.testing.assert_allclose(mm.weight[:], mw_ref, rtol=2e-6)

    dm = DirtyMapMaker(nside=c["nside"])
    dm.setup(bt)
    out = dm.process(mm)  # 2 slabs (one frequency's tiles are 51 GB)
    assert out.map.shape == (4, 4, 12 * 512 * 512)
    alm_d = dm.make_alm(mm)
    alm = alm_d.cpu().numpy()  # [f, 4, m, l]
    rng = np.random.default_rng(4)
    mw_dev = mm.weight[:]
    for m, f in ((0, 0), (3, 2), (500, 1), (1000, 3), (1024, 0)):
        a_ref = omm.dirty_solve(osyn.beam_tile(seed, m, f, npairs, 4, lmax), mv[m, :, f], mw_dev[m, :, f])
        assert _rel(alm[f, :, m, :], a_ref) < 1e-12, (m, f)
    # every (m, f): unit data vector on a row that differs per (m, f)
    rows = (np.arange(1025)[:, None] * 37 + np.arange(nfreq)[None, :] * 211) % (2 * npairs)
    e = np.zeros((1025, 2, nfreq, npairs), np.complex128)
    mi, fi = np.meshgrid(np.arange(1025), np.arange(nfreq), indexing="ij")
    e[mi, rows // npairs, fi, rows % npairs] = 1.0
    mm_e = containers.MModes(mmax=1024, freq=tel.frequencies, stack=npairs)
    mm_e.vis[:] = e
    mm_e.weight[:] = 1.0
    back = dm.make_alm(mm_e).cpu().numpy()
    for m in range(1025):
        for f in range(nfreq):
            assert np.array_equal(back[f, :, m, :], np.conj(osyn.beam_row(seed, m, f, rows[m, f], npairs, 4, lmax))), (m, f)
    # two ranks' shards of the same data: each rank's slab alone gives exactly its rows of the full result
    full_map = out.map[:]
    for rank in range(2):
        ss_r = parallel.shard_freq(ss, rank=rank, world=2)
        assert len(ss_r.index_map["freq"]) == 2
        mm_r = tr.process(ss_r)
        dm_r = DirtyMapMaker(nside=c["nside"])
        dm_r.setup(bt)  # the provider knows all frequencies; find_keys maps the shard's two
        out_r = dm_r.process(mm_r)
        assert np.array_equal(out_r.map[:], full_map[2 * rank : 2 * rank + 2])
        del out_r, mm_r, dm_r
    del out, alm_d, back
    torch.cuda.empty_cache()
    _solve.release_pools()


def test_cfg5_slice_simulate_noise_transform_wiener_round_trip():
    """BASELINE config 5 (CHIME-pathfinder scale: 763 baselines, lmax 1023, odd nra 2047, nside 512; 1024 frequencies
    over 8 GPUs) at its real sizes on ONE frequency: SimulateSidereal -> GaussianNoise -> MModeTransform ->
    WienerMapMaker through the task classes.  Checked on sampled m against oracle-generated tiles: the simulated
    m-modes are B_m a_m of the sky's a_lm, the Wiener a_lm satisfy the normal equations of mapmaker.py:260-272 with the
    noisy data and their weights, and the recovered map correlates with the input sky.
    (The Wishart task SampleNoise needs a positive-definite expectation per (freq, RA), which a random synthetic B does
    not give; it is covered on its own reference vectors in tests/test_noise.py.)"""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import WienerMapMaker
    from draco_amd.analysis.transform import MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context, ptr
    from draco_amd.synthesis.noise import GaussianNoise
    from draco_amd.synthesis.stream import SimulateSidereal

    _solve.release_pools()
    ctx = Context.get()
    c = osyn.CONFIGS[5]
    tel = _tel(5, 1)
    npairs, lmax, nside = tel.npairs, tel.lmax, c["nside"]
    assert (tel.nfeed, npairs, lmax, c["nra"], nside) == (256, 763, 1023, 2047, 512) and 2 * tel.mmax + 1 == c["nra"]
    seed = 3005
    bt = SyntheticProvider(tel, seed=seed)
    # sky of SURVEY 8d: band-limited Gaussian, C_l = (l+1)^-2, seed 4005, made on the device through alm2map
    gen = torch.Generator(device=ctx.device).manual_seed(4005)
    alm = torch.randn((1, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)  # [f, pol, m, l]
    ll = torch.arange(lmax + 1, device=ctx.device)
    alm = alm * ((ll.double() + 1.0) ** -1.0)[None, None, None, :] * (ll[None, :] >= ll[:, None]).to(alm.dtype)[None, None]
    alm[:, :, 0] = alm[:, :, 0].real.to(alm.dtype)
    alm[:, 1:3, :, :2] = 0
    alm = alm.contiguous()
    sky = ctx.empty((1, 4, 12 * nside * nside), np.float64)
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), 1, 4, lmax, lmax, nside, ptr(sky)))
    mp = containers.Map(nside=nside, freq=tel.frequencies, allocate=False)
    mp.attach("map", sky)

    sim = SimulateSidereal()
    sim.setup(bt)
    ss = sim.process(mp)
    assert ss.vis.shape == (1, npairs, 2047) and ss.vis.dtype == np.complex64
    tr = MModeTransform()
    tr.setup(bt)
    mm_clean = tr.process(ss).vis[:]  # [1024, 2, 1, 763]
    a_sky = alm.cpu().numpy()[0]  # [pol, m, l]: band-limited at lmax = 2 nside, map2alm (3 iterations) recovers it to ~1e-4
    for m in (1, 400, 1023):
        B = osyn.beam_tile(seed, m, 0, npairs, 4, lmax).reshape(2 * npairs, -1)
        v_ref = (B @ a_sky[:, m, :].reshape(-1)).reshape(2, npairs)
        assert _rel(mm_clean[m, :, 0], v_ref) < 2e-3, (m, _rel(mm_clean[m, :, 0], v_ref))  # the forward SHT's quadrature error at lmax = 2 nside

    gn = GaussianNoise(seed=5, ndays=733.0, recv_temp=50.0)
    gn.setup(tel)
    ss = gn.process(ss)
    mm = tr.process(ss)
    mv, mw = mm.vis[:], mm.weight[:]
    assert np.all(mw[:, :, 0] > 0) and mm.attrs["oddra"]
    wm = WienerMapMaker(nside=nside, prior_amp=1.0, prior_tilt=0.5)
    wm.setup(bt)
    a_w = wm.make_alm(mm).cpu().numpy()[0]  # [pol, m, l]
    assert np.all(np.isfinite(a_w))
    for m in (0, 400, 1023):
        B = osyn.beam_tile(seed, m, 0, npairs, 4, lmax)[..., m:].reshape(2 * npairs, -1)
        nv, vv = mw[m, :, 0].reshape(-1), mv[m, :, 0].reshape(-1)
        x = a_w[:, m, m:].reshape(-1)
        S = omm.wiener_prior(lmax, m, 1.0, 0.5)
        lhs = x / S + B.conj().T @ (nv * (B @ x))
        rhs = B.conj().T @ (nv * vv)
        assert _rel(lhs, rhs) < 1e-9, (m, _rel(lhs, rhs))
        assert np.all(a_w[:, m, :m] == 0)
    out = wm.process(mm)
    rec, inp = out.map[:][0, 0], sky.cpu().numpy()[0, 0]
    assert out.map.shape == (1, 4, 12 * 512 * 512) and np.all(np.isfinite(rec))
    corr = np.corrcoef(rec, inp)[0, 1]
    assert corr > 0.5, corr  # a Wiener-filtered estimate of the sky that went in
    del out, mp, sky
    torch.cuda.empty_cache()
    _solve.release_pools()

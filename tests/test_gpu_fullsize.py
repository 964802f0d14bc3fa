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


def _check_rows_against_oracle_svd(out, seed, mv, mw, f_bt, f_data, ms, tol):
    """Rows (m) of a BATCHED pass against the oracle's SVD solve (``pinv_svd`` restated, mapmaker.py:184-201,287-300)
    on the same tiles: the batch scheduler (half-batches, deferred orders, probes, chunk slots) is between the two."""
    from oracle import mapmaker as omm

    mv_h, mw_h = mv[:, :, f_data].cpu().numpy(), mw[:, :, f_data].cpu().numpy()
    npairs, lmax = mv_h.shape[-1], out.shape[-1] - 1
    worst = 0.0
    for m in ms:
        full = osyn.beam_tile(seed, int(m), f_bt, npairs, 4, lmax)
        ref = omm.ml_solve(full, mv_h[m], mw_h[m])
        worst = max(worst, _rel(out[f_data, :, m, :], ref))
    assert worst < tol, worst
    return worst


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


def test_cfg4_tile_wiener_and_ml_properties():
    """cfg-4/5 tile size (ntel = 1526 -> padded order 1536, nsky up to 4100): Wiener normal equations; ML on the
    telescope side (m = 0) and on the sky side (m = 900, order 500) satisfies the least-squares / minimum-norm
    conditions of the pseudo-inverse with nothing cut, and agrees with the eigen path."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    rng = np.random.default_rng(44)
    tel = _tel(4, 1)
    assert tel.npairs == 763
    bt = SyntheticProvider(tel, seed=44)
    lmax = 1024
    w = WienerMapMaker()
    w.setup(bt)
    ml = MaximumLikelihoodMapMaker()
    ml.setup(bt)
    for m in (0, 900):
        v = rng.standard_normal((2, 763)) + 1j * rng.standard_normal((2, 763))
        Ni = rng.uniform(0.5, 1.5, (2, 763)) * 20
        Ni[rng.uniform(size=Ni.shape) < 0.03] = 0
        B = osyn.beam_tile(44, m, 0, 763, 4, lmax)[..., m:].reshape(1526, -1)
        nv, vv = Ni.reshape(-1), v.reshape(-1)
        a = w._solve_m(m, 0, v, Ni)
        x = a[:, m:].reshape(-1)
        S = omm.wiener_prior(lmax, m)
        lhs = x / S + B.conj().T @ (nv * (B @ x))
        assert _rel(lhs, B.conj().T @ (nv * vv)) < 1e-10, m
        a = ml._solve_m(m, 0, v, Ni)
        x = a[:, m:].reshape(-1)
        Bt = np.sqrt(nv)[:, None] * B  # D B
        r = Bt @ x - np.sqrt(nv) * vv
        if B.shape[1] >= 1526:  # more unknowns than data: exact fit of the weighted data, minimum norm (x in range(Bt^H))
            assert np.abs(r).max() < 1e-9 * np.abs(np.sqrt(nv) * vv).max(), m
            y = np.linalg.lstsq(Bt.conj().T, x, rcond=None)[0]
            assert np.abs(Bt.conj().T @ y - x).max() < 1e-9 * np.abs(x).max(), m
        else:  # overdetermined: normal equations
            assert np.abs(Bt.conj().T @ r).max() < 1e-9 * np.abs(Bt.conj().T @ (np.sqrt(nv) * vv)).max(), m
        for eig in (1, 4):  # blocked Jacobi; Householder tridiagonalisation + QL (order 1536: six column chunks per wave)
            try:
                _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
                _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", eig))
                a_eig = ml._solve_m(m, 0, v, Ni)
            finally:
                _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
                _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 0))
            assert _rel(a, a_eig) < 1e-9, (m, eig)


def test_cfg4_sht_roundtrip_nside512():
    """lmax = 1024, nside = 512 (cfg 4/5's SHT): Bluestein classes up to M = 2048, the direct-sum fallback for the
    widest cap rings (their M = 4096 transform does not fit the LDS), and a band-limited round trip."""
    from draco_amd import _lib
    from draco_amd.device import Context, ptr
    import torch

    ctx = Context.get()
    lmax, nside = 1024, 512
    gen = torch.Generator(device=ctx.device).manual_seed(12)
    alm = torch.randn((1, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)  # [f, pol, m, l]
    m_idx = torch.arange(lmax + 1, device=ctx.device)
    keep = (m_idx[None, :] >= m_idx[:, None]).to(alm.dtype)  # l >= m
    alm = alm * keep[None, None]
    alm[:, :, 0] = alm[:, :, 0].real.to(alm.dtype)  # m = 0 is real
    alm[:, 1:3, :, :2] = 0                          # E, B start at l = 2
    alm = alm.contiguous()
    maps = ctx.empty((1, 4, 12 * nside * nside), np.float64)
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), 1, 4, lmax, lmax, nside, ptr(maps)))
    try:  # every ring through the direct sums
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"sht_variant", 4))
        maps_d = ctx.empty((1, 4, 12 * nside * nside), np.float64)
        _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), 1, 4, lmax, lmax, nside, ptr(maps_d)))
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"sht_variant", 0))
    assert float((maps - maps_d).abs().max()) < 1e-11 * float(maps_d.abs().max())
    back = ctx.empty((1, 4, lmax + 1, lmax + 1), np.complex128)
    _lib.check(_lib.lib.dmm_map2alm(ctx.handle, ptr(maps), 1, 4, lmax, lmax, nside, 3, ptr(back)))
    err = float((back - alm).abs().max()) / float(alm.abs().max())
    assert err < 5e-5, err  # lmax = 2 nside, white spectrum: the HEALPix quadrature is not exact; 3 Jacobi iterations
    _lib.check(_lib.lib.dmm_map2alm(ctx.handle, ptr(maps), 1, 4, lmax, lmax, nside, 8, ptr(back)))
    err8 = float((back - alm).abs().max()) / float(alm.abs().max())
    assert err8 < 0.5 * err, (err, err8)  # and the iteration converges


def test_cfg3_ml_eigen_pass_mixes_pipelined_and_synchronous_batches():
    """All 513 m of one cfg-3 frequency with every tile sent to the eigen path: the 324 telescope-side tiles run as
    pipelined half-batches (QL on the second stream), most sky-side orders have too few tiles for that and run as
    synchronous Jacobi batches in the same workspace.  The two must not overlap in it (they once did: the
    synchronous batch overwrote reflectors and rotation logs still in use); the answer must equal the all-Jacobi one."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(3, 1)
    lmax = tel.lmax
    bt = SyntheticProvider(tel, seed=31)
    gen = torch.Generator(device=ctx.device).manual_seed(4)
    shape = (lmax + 1, 2, 1, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) * 30.0 + 5.0
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED)
    out = {}
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
        for eig in (0, 1):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", eig))
            out[eig] = eng.solve("ml", mv, mw, [0], lmax).cpu().numpy()
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 0))
    assert np.all(np.isfinite(out[0]))
    assert _rel(out[0], out[1]) < 1e-9
    # and the batched pass against the oracle's SVD on sampled rows: telescope side, near-square, sky side
    _check_rows_against_oracle_svd(out[0], 31, mv, mw, 0, 0, (0, 77, 200, 310, 322, 324, 331, 400, 470, 512), 1e-8)


@pytest.mark.parametrize("nfeed_cyl,np_expected", [(33, 832), (70, 1728), (90, 2176)])
def test_ml_tridiagonal_path_at_other_orders(nfeed_cyl, np_expected):
    """The kernel variants of the tridiagonal eigen path are chosen by matrix order: six column chunks per wave and two
    pending updates up to 1536, eight chunks and one up to 2048, full-matrix trailing sweeps beyond.  One telescope-side
    tile (m = 0) and one sky-side tile at each of three orders in those ranges, against the blocked Jacobi."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context

    ctx = Context.get()
    lmax = 600
    tel = TransitTelescope(osyn.frequencies(1), lmax=lmax, ncyl=2, nfeed_cyl=nfeed_cyl)
    ntel = 2 * tel.npairs
    assert (ntel + 63) // 64 * 64 == np_expected and 4 * (lmax + 1) >= ntel
    bt = SyntheticProvider(tel, seed=60 + nfeed_cyl)
    ml = MaximumLikelihoodMapMaker()
    ml.setup(bt)
    rng = np.random.default_rng(nfeed_cyl)
    for m in (0, lmax - 100):  # telescope side; sky side of order 404 (padded 448)
        v = rng.standard_normal((2, tel.npairs)) + 1j * rng.standard_normal((2, tel.npairs))
        Ni = rng.uniform(0.5, 1.5, (2, tel.npairs)) * 10
        Ni[rng.uniform(size=Ni.shape) < 0.05] = 0
        out = {}
        try:
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
            for eig in (1, 4):
                _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", eig))
                out[eig] = ml._solve_m(m, 0, v, Ni)
        finally:
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 0))
        assert np.all(np.isfinite(out[4]))
        assert _rel(out[4], out[1]) < 1e-9, (m, _rel(out[4], out[1]))


def test_cfg3_ml_rank_deficient_telescope_side_stays_on_the_tridiagonal_path():
    """One cfg-3 frequency with every system formed on the telescope side ("ml_shortcut" = 3): beyond m = 323 the order-758
    Gram matrix has rank 4 (513 - m) < 758, i.e. up to 750 eigenvalues that are rounding dust.  QL must deflate inside
    such a cluster (absolute floor of its negligibility test) instead of running into the iteration cap -- before, every
    one of those tiles was redone by the Jacobi solver, 12x slower -- and the minimum-norm answer must agree with the
    default route, which solves the same tiles on the sky side."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(3, 1)
    lmax = tel.lmax
    bt = SyntheticProvider(tel, seed=32)
    gen = torch.Generator(device=ctx.device).manual_seed(5)
    shape = (lmax + 1, 2, 1, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) * 30.0 + 5.0
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED)

    def counter(name):
        import ctypes as C

        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    ref = eng.solve("ml", mv, mw, [0], lmax).cpu().numpy()
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 3))
        q0, e0 = counter(b"ml_tiles_ql_failed"), counter(b"ml_tiles_eigen")
        out = eng.solve("ml", mv, mw, [0], lmax).cpu().numpy()
        assert counter(b"ml_tiles_eigen") - e0 >= lmax - 323  # the rank-deficient tiles cannot pass the certificate
        assert counter(b"ml_tiles_ql_failed") == q0
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
    assert np.all(np.isfinite(out))
    for m in (0, 200, 330, 400, 480, 510):
        assert _rel(out[..., m, :], ref[..., m, :]) < 1e-7, m


def test_cfg3_ml_low_pass_rate_scheduling_matches_the_eigen_only_pass():
    """One cfg-3 frequency whose noise weights span eight decades: hardly any tile passes the full-rank certificate, so
    the default options go through everything the scheduler has for that regime -- whole batches deferred, 128-tile
    probes, deferred sky-side orders decomposed in pairs, largest first.  Every tile must be solved exactly once, and
    the answer must be the one of the pass that decomposes every tile outright."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(3, 1)
    lmax = tel.lmax
    bt = SyntheticProvider(tel, seed=33)
    gen = torch.Generator(device=ctx.device).manual_seed(6)
    shape = (lmax + 1, 2, 1, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    spread = torch.pow(10.0, -8.0 * torch.rand((1, 1, 1, tel.npairs), dtype=torch.float64, device=ctx.device, generator=gen))
    mw = (torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) * 40.0 + 10.0) * spread
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED)

    def counter(name):
        import ctypes as C

        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    d0, e0, q0 = counter(b"ml_tiles_direct"), counter(b"ml_tiles_eigen"), counter(b"ml_tiles_ql_failed")
    out = eng.solve("ml", mv, mw, [0], lmax, acond=1e-4, rcond=1e-3).cpu().numpy()
    nd, ne = counter(b"ml_tiles_direct") - d0, counter(b"ml_tiles_eigen") - e0
    assert nd + ne == lmax + 1 and ne > nd  # every tile once; most of them decomposed
    assert counter(b"ml_tiles_ql_failed") == q0
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
        ref = eng.solve("ml", mv, mw, [0], lmax, acond=1e-4, rcond=1e-3).cpu().numpy()
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
    assert np.all(np.isfinite(out))
    assert _rel(out, ref) < 1e-8
    # sampled rows of the scheduled pass against the oracle's SVD.  The weights span eight decades: modes just above the
    # reference's cut are amplified by up to 1 / acond^2, so agreement is looser than on well-conditioned tiles
    _check_rows_against_oracle_svd(out, 33, mv, mw, 0, 0, (0, 50, 150, 250, 321, 323, 326, 380, 450, 505), 1e-6)


def test_cfg3_ml_early_reject_chunk_runs_beside_full_direct_batches():
    """The certificate pass decomposes early-known rejects on two small chunk slots at the END of the workspace while
    the remaining direct batches keep the front.  Each user of the workspace writes nmat + 1 column-block prefix sums
    for its back-projection; a FULL direct batch (nmat == cap_direct) ends exactly where chunk slot 0 begins, and the
    two once shared that entry: the chunk's back-projection, which runs on the second stream after its
    ~0.1 s QL, then read the batch's total instead of its own zero and wrote tile 0's a_lm to the wrong place.
    Two cfg-3 frequencies with an 8 GiB workspace (cap ~ 280 matrices, slots of ~35, direct batches of ~210: three
    full ones for the 648 telescope-side tiles); the weights of ten m near the square tiles span eight decades, so
    those 20 tiles are rejected in the FIRST direct batch and go to slot 0 under the full batches that follow.  The
    answer must be the one of the pass that decomposes every tile."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(3, 2)
    lmax = tel.lmax
    bt = SyntheticProvider(tel, seed=34)
    gen = torch.Generator(device=ctx.device).manual_seed(8)
    shape = (lmax + 1, 2, 2, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) * 30.0 + 5.0
    spread = torch.pow(10.0, -8.0 * torch.rand((10, 1, 1, tel.npairs), dtype=torch.float64, device=ctx.device, generator=gen))
    mw[290:300] *= spread  # ten ill-conditioned m inside the first telescope-side batch (m = 323 downwards)

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    class FixedWorkspace(SolveEngine):
        def _offer_workspace(self, option, cap_mib):  # (the engine would offer half of the free HBM)
            _lib.check(_lib.lib.dmm_ctx_set_option(self.ctx.handle, option, 8192))

    eng = FixedWorkspace(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED)
    try:
        c0, e0, d0 = counter(b"ml_early_chunks"), counter(b"ml_tiles_eigen"), counter(b"ml_tiles_direct")
        out = eng.solve("ml", mv, mw, [0, 1], lmax, acond=1e-4, rcond=1e-3).cpu().numpy()
        early = counter(b"ml_early_chunks") - c0
        ne, nd = counter(b"ml_tiles_eigen") - e0, counter(b"ml_tiles_direct") - d0
        assert early >= 1, "the early-reject path was not taken: the test no longer covers what it is for"
        assert ne + nd == 2 * (lmax + 1) and 20 <= ne <= 70, (ne, nd)
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
        ref = eng.solve("ml", mv, mw, [0, 1], lmax, acond=1e-4, rcond=1e-3).cpu().numpy()
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_workspace_mib", 0))
    assert np.all(np.isfinite(out))
    assert _rel(out, ref) < 1e-8
    # rows of both frequencies against the oracle's SVD: certified tiles, the ill-conditioned ones of the early chunk, sky side
    _check_rows_against_oracle_svd(out, 34, mv, mw, 0, 0, (0, 120, 291, 296, 322, 330, 500), 1e-6)
    _check_rows_against_oracle_svd(out, 34, mv, mw, 1, 1, (3, 240, 293, 299, 323, 410), 1e-6)


def test_cfg3_ml_two_stage_reduction_against_the_one_stage_reduction():
    """The eigen path's tridiagonal reduction in both forms on one cfg-3 frequency, every tile decomposed: two-stage
    (dense -> band of half-width 8 on the matrix cores -> tridiagonal by bulge chasing in LDS, "ml_reduce" = 0, the
    default wherever the band fits the LDS) and one-stage Householder ("ml_reduce" = 1).  Telescope-side order 768 and
    every sky-side order from 704 down to 64; the library's kernel-class timers say which reduction ran."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(3, 1)
    lmax = tel.lmax
    bt = SyntheticProvider(tel, seed=35)
    gen = torch.Generator(device=ctx.device).manual_seed(9)
    shape = (lmax + 1, 2, 1, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) * 30.0 + 5.0
    mw[torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) < 0.02] = 0.0
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED)

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    out, spans = {}, {}
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
        for red in (0, 1):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_reduce", red))
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"profile", 1))
            out[red] = eng.solve("ml", mv, mw, [0], lmax).cpu().numpy()
            spans[red] = (counter(b"prof_band_n"), counter(b"prof_tridiag_n"))
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"profile", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_reduce", 0))
    assert spans[0][0] > 0 and spans[0][1] == 0, spans  # two-stage only
    assert spans[1][0] == 0 and spans[1][1] > 0, spans  # one-stage only
    assert np.all(np.isfinite(out[0]))
    assert _rel(out[0], out[1]) < 1e-9
    _check_rows_against_oracle_svd(out[0], 35, mv, mw, 0, 0, (1, 160, 323, 324, 390, 449, 497, 511), 1e-8)


def test_ml_two_stage_reduction_at_the_largest_order_the_lds_takes():
    """Orders 832 (the largest whose band plus the chase kernel's scratch fits the 160 KB of LDS) down to 64 in one
    pass: ntel = 820 on the telescope side (m <= 6), every sky-side order below it.  The reductions against each
    other -- stage 1 with every other two-sided update deferred (default), the same with its reading sweeps as one block
    per matrix ("ml_reduce" = 3), as one kernel ("ml_reduce" = 5: only in a `make EXTRA=-DDMM_AB` build, the default form
    otherwise), with no update deferred ("ml_reduce" = 2: rounds 3-5's form), one-stage ("ml_reduce" = 1) -- and sampled
    rows against the oracle's SVD."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider, TransitTelescope
    from draco_amd.device import Context

    ctx = Context.get()
    lmax = 210
    tel = TransitTelescope(osyn.frequencies(1), lmax=lmax, ncyl=1, nfeed_cyl=3, npairs=410)
    assert 2 * tel.npairs == 820
    bt = SyntheticProvider(tel, seed=36)
    gen = torch.Generator(device=ctx.device).manual_seed(10)
    shape = (lmax + 1, 2, 1, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) * 30.0 + 5.0
    mw[torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) < 0.02] = 0.0
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED)

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    out, spans = {}, {}
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
        for red in (0, 3, 5, 2, 1):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_reduce", red))
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"profile", 1))
            out[red] = eng.solve("ml", mv, mw, [0], lmax).cpu().numpy()
            spans[red] = (counter(b"prof_band_n"), counter(b"prof_chase_n"), counter(b"prof_tridiag_n"))
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"profile", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_reduce", 0))
    for red in (0, 3, 5, 2):
        assert spans[red][0] > 0 and spans[red][1] > 0 and spans[red][2] == 0, spans  # two-stage only
    assert spans[1][0] == 0 and spans[1][2] > 0, spans  # one-stage only
    assert np.all(np.isfinite(out[0]))
    assert _rel(out[0], out[1]) < 1e-9
    assert _rel(out[2], out[1]) < 1e-9
    assert _rel(out[3], out[1]) < 1e-9
    assert _rel(out[5], out[1]) < 1e-9
    _check_rows_against_oracle_svd(out[0], 36, mv, mw, 0, 0, (0, 6, 7, 100, 195, 209, 210), 1e-8)

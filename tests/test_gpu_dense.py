"""GPU parity: Wiener and maximum-likelihood solves vs the oracle and the reference's golden vectors.

float64 throughout.  Wiener: Gram (f64 MFMA) + Cholesky; conditioning of I + B~ S B~^H is
mild, asserted 1e-10 relative.  ML: Hermitian Jacobi on the Gram matrix; singular values
come out as sqrt(eigenvalue), so modes close to the cut lose digits (sigma ~ 1e-3 sigma_max
-> relative 1e-10 on the eigenvalue); asserted 1e-8 relative on well separated spectra.
"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import synth as osyn


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _tel(nfreq, lmax, ncyl=1, nfeed_cyl=3, npairs=None):
    from draco_amd.core.products import TransitTelescope

    return TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, ncyl=ncyl, nfeed_cyl=nfeed_cyl, npairs=npairs)


def test_solve_m_golden(golden_dir):
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.core.products import ArrayProvider

    g = np.load(os.path.join(golden_dir, "mapmaker_solve_m.npz"))
    for i in range(int(g["ncase"])):
        bm, m, v, Ni = g[f"c{i}_bm"], int(g[f"c{i}_m"]), g[f"c{i}_v"], g[f"c{i}_Ni"]
        npairs, lmax = bm.shape[1], bm.shape[3] - 1
        tel = _tel(3, lmax, npairs=npairs)
        bt = ArrayProvider(tel, lambda mm, ff, bm=bm: bm)
        w = WienerMapMaker()
        w.setup(bt)
        a = w._solve_m(m, 0, v, Ni)
        assert _rel(a, g[f"c{i}_wiener"]) < 1e-10, f"wiener case {i}"
        assert np.all(a[:, :m] == 0)
        w2 = WienerMapMaker(prior_amp=2.5, prior_tilt=1.25)
        w2.setup(bt)
        assert _rel(w2._solve_m(m, 0, v, Ni), g[f"c{i}_wiener_p"]) < 1e-10, f"wiener_p case {i}"
        ml = MaximumLikelihoodMapMaker()
        ml.setup(bt)
        a = ml._solve_m(m, 0, v, Ni)
        assert _rel(a, g[f"c{i}_ml"]) < 1e-8, f"ml case {i}"


def test_pinv_svd_golden(golden_dir):
    from draco_amd.analysis.mapmaker import pinv_svd

    g = np.load(os.path.join(golden_dir, "mapmaker_pinv_svd.npz"))
    for i in range(int(g["ncase"])):
        M, ref = g[f"c{i}_M"], g[f"c{i}_pinv"]
        out = pinv_svd(M)
        assert out.shape == ref.shape
        assert _rel(out, ref) < 1e-7, f"case {i}"  # case 1/2 have singular values within 20 % of the cut


@pytest.mark.parametrize("kind,tol", [("wiener", 1e-10), ("ml", 1e-8)])
@pytest.mark.parametrize("nfreq,lmax,ncyl,nfeed,b_dtype", [(2, 12, 1, 3, "complex128"), (2, 40, 2, 4, "complex128"), (1, 30, 1, 5, "complex64")])
def test_alm_vs_oracle(kind, tol, nfreq, lmax, ncyl, nfeed, b_dtype):
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    tel = _tel(nfreq, lmax, ncyl, nfeed)  # ntel = 46 / 182 / 78: 1..3 tiles of 64, exercises the padding
    seed = 500 + lmax
    bt = SyntheticProvider(tel, seed=seed)
    rng = np.random.default_rng(lmax)
    mv = rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape) * 30.0
    mw[rng.uniform(size=mw.shape) < 0.1] = 0.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    cls = WienerMapMaker if kind == "wiener" else MaximumLikelihoodMapMaker
    task = cls(b_dtype=b_dtype)
    task.setup(bt)
    alm = task.alm_square(task.make_alm(mm))
    beam = lambda m, f: osyn.beam_tile(seed, m, f, tel.npairs, 4, lmax)  # noqa: E731
    if b_dtype == "complex64":
        beam = lambda m, f: osyn.beam_tile(seed, m, f, tel.npairs, 4, lmax).astype(np.complex64).astype(np.complex128)  # noqa: E731
    ref = omm.solve_alm(kind, beam, mv, mw, lmax, tel.mmax, list(range(nfreq)))
    # ML: compare where the spectrum is well separated from the cut; otherwise the rank decision itself may differ
    assert _rel(alm, ref) < tol


@pytest.mark.parametrize("kind,tol", [("wiener", 1e-10), ("ml", 1e-8)])
def test_gram_operands_by_lds_dma_vs_oracle(kind, tol):
    """The A/B form of the beam Gram kernel (`gram_stage` = 1: operands DMA'd into a source-swizzled LDS image, the
    prior applied after the LDS read, the last partial chunk masked) against the oracle: ntel = 182 (three row tiles,
    padding), K = 4 (lmax + 1 - m) from 164 down to 4, so every m has a partial last chunk of a different length."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    nfreq, lmax = 2, 40
    tel = _tel(nfreq, lmax, 2, 4)
    bt = SyntheticProvider(tel, seed=540)
    rng = np.random.default_rng(41)
    mv = rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape) * 30.0
    mw[rng.uniform(size=mw.shape) < 0.1] = 0.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    ctx = Context.get()
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"gram_stage", 1))
    try:
        task = (WienerMapMaker(prior_amp=2.5, prior_tilt=1.25) if kind == "wiener" else MaximumLikelihoodMapMaker())
        task.setup(bt)
        alm = task.alm_square(task.make_alm(mm))
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"gram_stage", 0))
    beam = lambda m, f: osyn.beam_tile(540, m, f, tel.npairs, 4, lmax)  # noqa: E731
    kw = dict(prior_amp=2.5, prior_tilt=1.25) if kind == "wiener" else {}
    ref = omm.solve_alm(kind, beam, mv, mw, lmax, tel.mmax, list(range(nfreq)), **kw)
    assert _rel(alm, ref) < tol


def test_wiener_two_batches_in_flight_is_the_same_solve(monkeypatch):
    """`dmm_wiener_run` alternates its batches between two streams, half the workspace each (`wiener_overlap`, default
    on).  A small workspace forces several batches per side; every tile's arithmetic is independent of its batch, so
    the one-stream pass must give the same bits -- and both agree with the oracle."""
    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.analysis.mapmaker import WienerMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    nfreq, lmax = 2, 60
    tel = _tel(nfreq, lmax, 2, 4)  # ntel = 182: telescope side for m <= 15 (4 (61 - m) >= 182), sky side above
    bt = SyntheticProvider(tel, seed=541)
    rng = np.random.default_rng(43)
    mv = rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape) * 30.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    ctx = Context.get()
    out, batches = {}, {}
    # the engine offers the solver half of the free HBM (one batch would take every tile): pin 24 MiB instead
    monkeypatch.setattr(SolveEngine, "_offer_workspace", lambda self, option, cap_mib: _lib.check(_lib.lib.dmm_ctx_set_option(self.ctx.handle, option, 24)))
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"profile", 1))
        for ov in (1, 0):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"wiener_overlap", ov))
            task = WienerMapMaker(prior_amp=1.5, prior_tilt=0.75)
            task.setup(bt)
            n0 = _counter(ctx, b"prof_gram_n")
            out[ov] = task.alm_square(task.make_alm(mm))
            batches[ov] = _counter(ctx, b"prof_gram_n") - n0  # one Gram span per batch
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"profile", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"wiener_overlap", 1))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"wiener_workspace_mib", 0))
    # batches alternate between the two streams: more than two per stream = a half's device lists were reused
    assert batches[1] >= 6 and batches[0] >= 3, batches
    assert np.array_equal(out[1], out[0])
    beam = lambda m, f: osyn.beam_tile(541, m, f, tel.npairs, 4, lmax)  # noqa: E731
    ref = omm.solve_alm("wiener", beam, mv, mw, lmax, tel.mmax, list(range(nfreq)), prior_amp=1.5, prior_tilt=0.75)
    assert _rel(out[1], ref) < 1e-10


def _counter(ctx, name):
    import ctypes as C

    from draco_amd import _lib

    v = C.c_int64()
    _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
    return int(v.value)


def test_ml_certified_shortcut_vs_eigen_path():
    """ML: the certified full-rank shortcut (Cholesky on the smaller Gram matrix, telescope or sky side)
    against the eigen-decomposition path on the same tiles, and both against the oracle.

    ntel = 86, nsky_m = 4 (61 - m): m <= 39 solve on the telescope side, m >= 40 on the sky side;
    10 % of the weights are zero (pinned rows) and two (m, f) have a rank-deficient B (equal rows,
    equal columns, a zero column) so that their certificates must fail and the eigen path run.
    """
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import ArrayProvider
    from draco_amd.device import Context

    ctx = Context.get()
    nfreq, lmax = 2, 60
    tel = _tel(nfreq, lmax, 2, 4)
    assert 2 * tel.npairs == 86
    seed = 4242

    def beam(m, f):
        b = osyn.beam_tile(seed, m, f, tel.npairs, 4, lmax).copy()
        if f == 1 and m in (3, 50):  # deficient on either side: nsky_3 = 232 > 86 (telescope), nsky_50 = 44 (sky)
            b[0, 1] = b[0, 0]
            b[1, 5] = b[1, 2]
            b[:, :, 2, m + 1] = 0.0
            b[:, :, :, m + 2] = b[:, :, :, m + 3]
        return b

    bt = ArrayProvider(tel, beam)
    rng = np.random.default_rng(9)
    mv = rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape) * 30.0
    mw[rng.uniform(size=mw.shape) < 0.1] = 0.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    ref = omm.solve_alm("ml", beam, mv, mw, lmax, tel.mmax, list(range(nfreq)))
    out = {}
    stats = {}
    try:
        for mode in (0, 2, 3):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", mode))
            d0, e0 = _counter(ctx, b"ml_tiles_direct"), _counter(ctx, b"ml_tiles_eigen")
            task = MaximumLikelihoodMapMaker()
            task.setup(bt)
            out[mode] = task.alm_square(task.make_alm(mm))
            stats[mode] = (_counter(ctx, b"ml_tiles_direct") - d0, _counter(ctx, b"ml_tiles_eigen") - e0)
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
    ntile = nfreq * (lmax + 1)
    assert stats[2] == (0, ntile)
    assert stats[0][0] + stats[0][1] == ntile and stats[0][1] >= 2  # the two deficient tiles at least
    assert stats[0][0] > ntile // 2, stats  # and most tiles take the shortcut
    for mode in (0, 2, 3):
        assert _rel(out[mode], ref) < 1e-8, (mode, stats[mode])
    assert _rel(out[0], out[2]) < 1e-9


def test_wiener_sky_side_vs_telescope_side():
    """Wiener: at high m (nsky_m < ntel) the sky-side system S^-1 + B^H Ni B is solved instead of the telescope-side
    one; both are the reference's own branches (mapmaker.py:267-278) and must agree with each other and the oracle."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import WienerMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    nfreq, lmax = 2, 60
    tel = _tel(nfreq, lmax, 2, 4)  # ntel = 86: sky side for m >= 40
    bt = SyntheticProvider(tel, seed=606)
    rng = np.random.default_rng(6)
    mv = rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((lmax + 1, 2, nfreq, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape) * 30.0
    mw[rng.uniform(size=mw.shape) < 0.1] = 0.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    out = {}
    try:
        for mode in (0, 3):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", mode))
            task = WienerMapMaker(prior_amp=2.0, prior_tilt=0.75)
            task.setup(bt)
            out[mode] = task.alm_square(task.make_alm(mm))
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
    beam = lambda m, f: osyn.beam_tile(606, m, f, tel.npairs, 4, lmax)  # noqa: E731
    ref = omm.solve_alm("wiener", beam, mv, mw, lmax, tel.mmax, list(range(nfreq)), prior_amp=2.0, prior_tilt=0.75)
    assert _rel(out[0], ref) < 1e-10
    assert _rel(out[3], ref) < 1e-10
    assert _rel(out[0], out[3]) < 1e-11


def test_wiener_cfg2_sized_tile_properties():
    """cfg-2 sized tile (374 x 1028): Wiener solution satisfies its normal equations (size-independent check)."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(2, 256, 2, 16)
    assert tel.npairs == 187
    bt = SyntheticProvider(tel, seed=77)
    ms = [0, 100, 256]
    gen = torch.Generator(device=ctx.device).manual_seed(3)
    mv = torch.randn((257, 2, 2, 187), dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand((257, 2, 2, 187), dtype=torch.float64, device=ctx.device, generator=gen) * 50
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, pool_bytes=3 * 2**30, cache=False)
    # restrict to a few m by zeroing the rest is not possible: run mmax=256 over 2 freqs would be 514 tiles; keep it small
    from draco_amd.analysis.mapmaker import WienerMapMaker

    task = WienerMapMaker()
    task.setup(bt)
    for m in ms:
        v = mv[m, :, 0].cpu().numpy()
        Ni = mw[m, :, 0].cpu().numpy()
        a = task._solve_m(m, 0, v, Ni)
        B = osyn.beam_tile(77, m, 0, 187, 4, 256)[..., m:].reshape(374, -1)
        S = omm.wiener_prior(256, m)
        lhs = a[:, m:].reshape(-1) / S + B.conj().T @ (Ni.reshape(-1) * (B @ a[:, m:].reshape(-1)))
        rhs = B.conj().T @ (Ni.reshape(-1) * v.reshape(-1))
        assert _rel(lhs, rhs) < 1e-10, m


@pytest.mark.parametrize("kind", ["wiener", "ml"])
def test_dense_full_layout_matches_packed(kind):
    """B handed over in the FULL [.., lmax+1] layout (driftscan's on-disk shape) must give the packed layout's
    alm for both dense solvers, on both the telescope-side (low m) and sky-side (high m) systems."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    lmax = 50
    tel = _tel(2, lmax, 2, 4)  # ntel = 86: both sides of the nsky_m / ntel split occur
    bt = SyntheticProvider(tel, seed=77)
    gen = torch.Generator(device=ctx.device).manual_seed(3)
    shape = (lmax + 1, 2, 2, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) * 20.0
    kw = {"prior_amp": 1.5, "prior_tilt": 0.5} if kind == "wiener" else {}
    outs = {}
    for dt in (_lib.DMM_C128, _lib.DMM_C64):
        for layout in (_lib.DMM_B_PACKED, _lib.DMM_B_FULL):
            eng = SolveEngine(bt, ctx, dt, layout)
            outs[dt, layout] = eng.solve(kind, mv, mw, [0, 1], lmax, **kw).cpu().numpy()
    for dt in (_lib.DMM_C128, _lib.DMM_C64):
        assert _rel(outs[dt, _lib.DMM_B_PACKED], outs[dt, _lib.DMM_B_FULL]) < 1e-12, dt


_VARIANT_REF = {}


@pytest.mark.parametrize("variant", [4, 1, 2, 3, 0])
def test_ml_eigen_path_variants(variant):
    """The eigen path of ML (`ml_shortcut` = 2 sends every tile there) in all its implementations:
    4 Householder tridiagonalisation (upper-triangle trailing updates) + QL in factored form, 1 blocked Jacobi,
    2 as 4 with full-matrix trailing updates, 3 QL forced to give up on every other matrix -> Jacobi fallback on the re-formed Gram matrices of those,
    0 the default (by batch size).
    Rank-deficient tiles, zero weights, telescope- and sky-side orders (64 ... 192) against the oracle's SVD."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import ArrayProvider
    from draco_amd.device import Context

    ctx = Context.get()
    nfreq, lmax = 2, 70
    tel = _tel(nfreq, lmax, 2, 6)  # ntel = 134: padded order 192; sky side from m = 38
    seed = 515

    def beam(m, f):
        b = osyn.beam_tile(seed, m, f, tel.npairs, 4, lmax).copy()
        if f == 0 and m in (2, 45, 69):
            b[0, 3] = b[0, 1]
            b[:, :, 1, min(m + 1, lmax)] = 0.0
        return b

    bt = ArrayProvider(tel, beam)
    rng = np.random.default_rng(10)
    shape = (lmax + 1, 2, nfreq, tel.npairs)
    mv = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    mw = rng.uniform(0.5, 1.5, shape) * 10.0
    mw[rng.uniform(size=shape) < 0.1] = 0.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    if "ref" not in _VARIANT_REF:  # the oracle's SVDs take most of the time: same inputs for every variant
        _VARIANT_REF["ref"] = omm.solve_alm("ml", beam, mv, mw, lmax, tel.mmax, list(range(nfreq)))
    ref = _VARIANT_REF["ref"]
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", variant))
        e0 = _counter(ctx, b"ml_tiles_eigen")
        task = MaximumLikelihoodMapMaker()
        task.setup(bt)
        out = task.alm_square(task.make_alm(mm))
        assert _counter(ctx, b"ml_tiles_eigen") - e0 == nfreq * (lmax + 1)
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 0))
    assert _rel(out, ref) < 1e-8, variant


@pytest.mark.parametrize("variant", [4, 1])
def test_ml_eigen_path_degenerate_tiles(variant):
    """Degenerate Gram matrices through the eigen path: a tile whose weights are all zero (G = 0, the answer is 0),
    a tile with orthogonal rows of equal norm (G = c I: every eigenvalue equal, nothing to rotate), a tile with one
    non-zero row (rank 1) and a tile scaled by 1e-3 so that the absolute cut `acond` removes modes the relative one
    keeps.  All against the oracle's SVD."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import ArrayProvider
    from draco_amd.device import Context

    ctx = Context.get()
    nfreq, lmax = 1, 12
    tel = _tel(nfreq, lmax, 1, 3)
    npairs, ntel = tel.npairs, 2 * tel.npairs
    seed = 99

    def beam(m, f):
        b = osyn.beam_tile(seed, m, f, npairs, 4, lmax).copy()
        if m == 1:  # orthogonal rows of equal norm: row i of the (ntel, nsky) matrix = unit vector i
            b[:] = 0.0
            flat = b.reshape(ntel, 4 * (lmax + 1))
            for i in range(ntel):
                flat[i, (i % 4) * (lmax + 1) + 1 + i // 4] = 2.0  # columns with l >= m = 1
        if m == 2:
            b[:] = 0.0
            b[0, 0, :, 2:] = osyn.beam_tile(seed, 2, f, npairs, 4, lmax)[0, 0, :, 2:]
        if m == 3:
            b *= 1e-3
        return b

    assert ntel <= 4 * lmax  # room for the unit vectors of m = 1
    bt = ArrayProvider(tel, beam)
    rng = np.random.default_rng(3)
    shape = (lmax + 1, 2, nfreq, npairs)
    mv = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    mw = rng.uniform(0.5, 1.5, shape)
    mw[0] = 0.0  # m = 0: no data at all
    mw[1] = 0.7  # equal weights keep G = c I
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    ref = omm.solve_alm("ml", beam, mv, mw, lmax, tel.mmax, [0])
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 2))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", variant))
        task = MaximumLikelihoodMapMaker()
        task.setup(bt)
        out = task.alm_square(task.make_alm(mm))
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 0))
    assert np.all(np.isfinite(out))
    assert np.all(out[..., 0] == 0)  # [freq, pol, l, m]
    for m in range(lmax + 1):
        scale = max(np.abs(ref[..., m]).max(), 1e-300)
        assert np.abs(out[..., m] - ref[..., m]).max() / scale < 1e-8, m


@pytest.mark.parametrize("variant", [4, 1])
def test_ml_weights_over_eight_decades(variant):
    """Well-conditioned beam transfers but noise weights spread over 1 ... 1e-8 (one per baseline): the weighted
    matrix D B is ill conditioned, the certificate fails at low m, and the reference's cut drops the baselines whose
    weight is negligible.  Against the oracle's SVD of D B."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    nfreq, lmax = 1, 40
    tel = _tel(nfreq, lmax, 2, 4)
    bt = SyntheticProvider(tel, seed=909)
    rng = np.random.default_rng(8)
    shape = (lmax + 1, 2, nfreq, tel.npairs)
    mv = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    mw = np.broadcast_to(10.0 ** (-8.0 * rng.uniform(size=(1, 2, nfreq, tel.npairs))), shape) * 30.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    beam = lambda m, f: osyn.beam_tile(909, m, f, tel.npairs, 4, lmax)  # noqa: E731
    ref = omm.solve_alm("ml", beam, mv, np.array(mw), lmax, tel.mmax, [0])
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", variant))
        e0 = _counter(ctx, b"ml_tiles_eigen")
        task = MaximumLikelihoodMapMaker()
        task.setup(bt)
        out = task.alm_square(task.make_alm(mm))
        assert _counter(ctx, b"ml_tiles_eigen") - e0 > 0
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 0))
    for m in range(lmax + 1):
        err = np.abs(out[..., m] - ref[..., m]).max() / np.abs(ref[..., m]).max()
        assert err < 1e-7, (m, err)


@pytest.mark.parametrize("variant", [4, 1])
def test_ml_ill_conditioned_tiles(variant):
    """Beam transfers with a geometric singular spectrum (1 ... 1e-8, the regime of real telescopes): about a third of
    the modes survive pinv_svd's relative cut at 1e-3.  The Gram route squares the condition number, so a kept mode
    sigma carries a relative error ~ eps (sigma_max / sigma)^2 <= 1e-10; agreement with the oracle's SVD of D B itself
    is asserted at 1e-7 of the solution's scale (far inside the 1e-5 map tolerance of the task)."""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import ArrayProvider
    from draco_amd.device import Context

    ctx = Context.get()
    nfreq, lmax = 1, 40
    tel = _tel(nfreq, lmax, 2, 4)
    npairs, ntel = tel.npairs, 2 * tel.npairs
    rng = np.random.default_rng(77)
    tiles = {}

    def beam(m, f):
        if m not in tiles:
            nsky = 4 * (lmax + 1 - m)
            k = min(ntel, nsky)
            u, _ = np.linalg.qr(rng.standard_normal((ntel, k)) + 1j * rng.standard_normal((ntel, k)))
            v, _ = np.linalg.qr(rng.standard_normal((nsky, k)) + 1j * rng.standard_normal((nsky, k)))
            sig = np.logspace(0, -8, k)
            b = np.zeros((2, npairs, 4, lmax + 1), dtype=np.complex128)
            b[..., m:] = ((u * sig) @ v.conj().T).reshape(2, npairs, 4, lmax + 1 - m)
            tiles[m] = b
        return tiles[m]

    bt = ArrayProvider(tel, beam)
    shape = (lmax + 1, 2, nfreq, npairs)
    mv = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    mw = np.ones(shape)  # unit weights keep the singular values those of B; acond = 1e-4 then cuts below the relative rule
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    ref = omm.solve_alm("ml", beam, mv, mw, lmax, tel.mmax, [0])
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", variant))
        d0, e0 = _counter(ctx, b"ml_tiles_direct"), _counter(ctx, b"ml_tiles_eigen")
        task = MaximumLikelihoodMapMaker()
        task.setup(bt)
        out = task.alm_square(task.make_alm(mm))
        assert _counter(ctx, b"ml_tiles_direct") == d0  # no tile passes the full-rank certificate
        assert _counter(ctx, b"ml_tiles_eigen") - e0 == lmax + 1
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 0))
    for m in range(lmax + 1):
        err = np.abs(out[..., m] - ref[..., m]).max() / np.abs(ref[..., m]).max()
        assert err < 1e-7, (m, err)


def test_ml_rank_deficient_tiles_stay_on_the_tridiagonal_path():
    """Beam transfers of rank ntel/2 (half of G's spectrum is a cluster of rounding-dust zeros: nsky < ntel, dead or
    redundant baselines) against the oracle's SVD, every tile solved by the tridiagonal path itself.  (Small-order
    companion of tests/test_gpu_fullsize.py::test_cfg3_ml_rank_deficient_telescope_side_stays_on_the_tridiagonal_path,
    where clusters of hundreds of zeros made QL give up before its negligibility test got an absolute floor.)"""
    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import ArrayProvider
    from draco_amd.device import Context

    ctx = Context.get()
    nfreq, lmax = 1, 40
    tel = _tel(nfreq, lmax, 2, 4)
    npairs, ntel = tel.npairs, 2 * tel.npairs
    rng = np.random.default_rng(78)
    tiles = {}

    def beam(m, f):
        if m not in tiles:
            nsky = 4 * (lmax + 1 - m)
            k = max(1, min(ntel, nsky) // 2)
            u, _ = np.linalg.qr(rng.standard_normal((ntel, k)) + 1j * rng.standard_normal((ntel, k)))
            v, _ = np.linalg.qr(rng.standard_normal((nsky, k)) + 1j * rng.standard_normal((nsky, k)))
            sig = np.linspace(1.0, 0.1, k)
            b = np.zeros((2, npairs, 4, lmax + 1), dtype=np.complex128)
            b[..., m:] = ((u * sig) @ v.conj().T).reshape(2, npairs, 4, lmax + 1 - m)
            tiles[m] = b
        return tiles[m]

    bt = ArrayProvider(tel, beam)
    shape = (lmax + 1, 2, nfreq, npairs)
    mv = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    mw = np.ones(shape)
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    ref = omm.solve_alm("ml", beam, mv, mw, lmax, tel.mmax, [0])
    try:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 4))
        e0, q0 = _counter(ctx, b"ml_tiles_eigen"), _counter(ctx, b"ml_tiles_ql_failed")
        task = MaximumLikelihoodMapMaker()
        task.setup(bt)
        out = task.alm_square(task.make_alm(mm))
        assert _counter(ctx, b"ml_tiles_eigen") - e0 == lmax + 1
        assert _counter(ctx, b"ml_tiles_ql_failed") == q0
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_eigen", 0))
    for m in range(lmax + 1):
        err = np.abs(out[..., m] - ref[..., m]).max() / np.abs(ref[..., m]).max()
        assert err < 1e-7, (m, err)

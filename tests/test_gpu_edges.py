"""GPU edge cases the reference's semantics imply: empty/ragged batches, mmax < lmax, one polarisation,
the `_solve_m` subclass hook, masked (zero-weight) m-modes, error paths of the C ABI."""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import sht as osht
from oracle import synth as osyn
from oracle import transform as otr


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _tel(nfreq, lmax, **kw):
    from draco_amd.core.products import TransitTelescope

    kw.setdefault("ncyl", 1)
    kw.setdefault("nfeed_cyl", 3)
    return TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, **kw)


def test_empty_and_single_row_transform():
    from draco_amd.analysis import transform as T

    out = T._make_marray(np.zeros((0, 16), np.complex64), mmax=8)
    assert out.shape == (9, 2, 0)
    ts = (np.arange(16) + 1j).astype(np.complex64)[None]
    assert _rel(T._make_marray(ts, mmax=8, dtype=np.complex128), otr.make_marray(ts, mmax=8, dtype=np.complex128)) < 2e-6


def test_mmax_smaller_than_lmax_and_fewer_m_rows():
    """telescope.mmax < lmax, and data with fewer m rows than the telescope: mmax = min(...) (mapmaker.py:52)."""
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    nfreq, lmax, mmax_tel, nside = 2, 14, 9, 8
    tel = _tel(nfreq, lmax, mmax=mmax_tel)
    bt = SyntheticProvider(tel, seed=3)
    rng = np.random.default_rng(3)
    for n_m in (mmax_tel + 4, 6):  # more rows than the telescope's mmax / fewer
        mv = rng.standard_normal((n_m, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((n_m, 2, nfreq, tel.npairs))
        mw = rng.uniform(0.5, 1.5, mv.shape)
        mm = containers.MModes(mmax=n_m - 1, freq=tel.frequencies, stack=tel.npairs)
        mm.vis[:] = mv
        mm.weight[:] = mw
        task = DirtyMapMaker(nside=nside)
        task.setup(bt)
        alm = task.alm_square(task.make_alm(mm))
        ref = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(3, m, f, tel.npairs, 4, lmax), mv, mw, lmax, mmax_tel, [0, 1])
        assert _rel(alm, ref) < 1e-12
        out = task.process(mm)
        refmap = osht.sphtrans_inv_sky(ref, nside)
        assert _rel(out.map[:], refmap) < 1e-11


def test_single_polarisation_broadcasts_like_the_reference():
    """num_pol_sky = 1: the reference's 4-slot alm receives the T solution in every slot (mapmaker.py:71,94)."""
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    tel = _tel(2, 9, num_pol_sky=1)
    bt = SyntheticProvider(tel, seed=4)
    rng = np.random.default_rng(4)
    mv = rng.standard_normal((10, 2, 2, tel.npairs)) + 1j * rng.standard_normal((10, 2, 2, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape)
    mm = containers.MModes(mmax=9, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    task = DirtyMapMaker(nside=4)
    task.setup(bt)
    alm = task.alm_square(task.make_alm(mm))
    ref = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(4, m, f, tel.npairs, 1, 9), mv, mw, 9, 9, [0, 1], npol=1)
    ref[:, 1:] = ref[:, :1]  # NumPy broadcast of a [1, lmax+1] result into the 4 slots
    assert _rel(alm, ref) < 1e-12
    out = task.process(mm)
    assert out.map.shape == (2, 4, 192)
    assert _rel(out.map[:], osht.sphtrans_inv_sky(ref, 4)) < 1e-11


def test_solve_m_subclass_hook_is_honoured():
    """A user subclass overriding `_solve_m` (the reference's extension point, mapmaker.py:120-140) is called."""
    from draco_amd.analysis.mapmaker import BaseMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    calls = []

    class HalfDirty(BaseMapMaker):
        def _solve_m(self, m, f, v, Ni):
            calls.append((m, f))
            bt = self.beamtransfer
            bm = bt.beam_m(m, fi=f).reshape(bt.ntel, bt.nsky)
            return 0.5 * (bm.T.conj() @ (Ni.reshape(-1) * v.reshape(-1))).reshape(4, -1)

    tel = _tel(2, 5)
    bt = SyntheticProvider(tel, seed=6)
    rng = np.random.default_rng(6)
    mv = rng.standard_normal((6, 2, 2, tel.npairs)) + 1j * rng.standard_normal((6, 2, 2, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape)
    mm = containers.MModes(mmax=5, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    task = HalfDirty(nside=4)
    task.setup(bt)
    out = task.process(mm)
    assert len(calls) == 12
    alm = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(6, m, f, tel.npairs, 4, 5), mv, mw, 5, 5, [0, 1])
    assert _rel(out.map[:], 0.5 * osht.sphtrans_inv_sky(alm, 4)) < 1e-11


def test_masked_mmodes_all_three_makers():
    """MaskMModeData-style input (flagging.py:113-173): zero weights for autos, m = 0 and one sign: no NaNs, parity holds."""
    from draco_amd.analysis.mapmaker import DirtyMapMaker, MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    tel = _tel(1, 10)
    bt = SyntheticProvider(tel, seed=8)
    rng = np.random.default_rng(8)
    mv = rng.standard_normal((11, 2, 1, tel.npairs)) + 1j * rng.standard_normal((11, 2, 1, tel.npairs))
    mw = rng.uniform(0.5, 1.5, mv.shape) * 40
    mw[0] = 0.0          # m = 0 masked entirely
    mw[:, 1, :, ::3] = 0  # a third of the baselines masked on the negative side
    mw[5] = 0.0          # one whole m masked
    mm = containers.MModes(mmax=10, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    beam = lambda m, f: osyn.beam_tile(8, m, f, tel.npairs, 4, 10)  # noqa: E731
    for cls, kind, tol in ((DirtyMapMaker, "dirty", 1e-12), (WienerMapMaker, "wiener", 1e-10), (MaximumLikelihoodMapMaker, "ml", 1e-8)):
        task = cls()
        task.setup(bt)
        alm = task.alm_square(task.make_alm(mm))
        assert np.all(np.isfinite(alm))
        ref = omm.solve_alm(kind, beam, mv, mw, 10, 10, [0])
        assert _rel(alm, ref) < tol, kind
        assert np.all(alm[:, :, :, 0] == 0) and np.all(alm[:, :, :, 5] == 0)


def test_abi_argument_errors_on_gpu():
    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    x = ctx.zeros((4, 8), np.complex64)
    out = ctx.zeros((5, 2, 4), np.complex128)
    with pytest.raises(ValueError, match="bad out_dtype"):
        _lib.check(_lib.lib.dmm_mfft_pack(ctx.handle, ptr(x), 4, 8, ptr(out), 4, 7, None))
    with pytest.raises(_lib.DmmError, match="nra"):
        _lib.check(_lib.lib.dmm_mfft_pack(ctx.handle, ptr(x), 4, 20000, ptr(out), 4, 1, None))
    with pytest.raises(ValueError, match="nside must be a power of two"):
        _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(out), 1, 4, 3, 3, 6, ptr(out)))
    tiles = _lib.tile_array([9], [0], [0])
    h = C.c_void_p()
    with pytest.raises(ValueError, match="out of range"):
        _lib.check(_lib.lib.dmm_solve_plan_create(ctx.handle, tiles, 1, 3, 4, 5, 1, 6, 1, 1, C.byref(h)))
    with pytest.raises(ValueError, match="unknown option"):
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"nope", 1))


def test_mask_mmode_data_golden(golden_dir):
    """MaskMModeData against the reference's own outputs; in place, returns the same container."""
    import os

    from draco_amd.analysis.flagging import MaskMModeData
    from draco_amd.core import containers

    g = np.load(os.path.join(golden_dir, "flagging_mask_mmode.npz"))
    prod = np.zeros(len(g["prod_a"]), dtype=[("input_a", int), ("input_b", int)])
    prod["input_a"], prod["input_b"] = g["prod_a"], g["prod_b"]
    for i in range(int(g["ncase"])):
        w = g[f"c{i}_w"]
        auto, mzero, pos, neg, low = (int(x) for x in g[f"c{i}_opts"])
        mm = containers.MModes(mmax=w.shape[0] - 1, freq=np.arange(w.shape[2]) + 400.0, prod=prod, stack=len(prod), input=3)
        mm.weight[:] = w
        t = MaskMModeData(auto_correlations=bool(auto), m_zero=bool(mzero), positive_m=bool(pos), negative_m=bool(neg))
        t.mask_low_m = None if low < 0 else low
        out = t.process(mm)
        assert out is mm
        assert np.array_equal(mm.weight[:], g[f"c{i}_out"]), i


def test_collate_products_golden(golden_dir):
    """CollateProducts against the reference's own outputs (unstacked full-triangle inputs, extra feed,
    permuted/extra frequencies, all three weight schemes; already stacked inputs)."""
    import os

    from draco_amd.analysis.transform import CollateProducts
    from draco_amd.core import containers

    g = np.load(os.path.join(golden_dir, "transform_collate.npz"))
    nfeed = int(g["nfeed_tel"])

    class Tel:  # the same 1-cylinder pairing the golden generator used
        pass

    tel = Tel()
    tel.nfeed, tel.npairs = nfeed, nfeed
    tel.lmax = tel.mmax = 1
    tel.frequencies = g["tel_freq"]
    tel.input_index = np.array([(100 + i,) for i in range(nfeed)], dtype=[("chan_id", "<u2")])
    tel.uniquepairs = np.array([(0, d) for d in range(nfeed)])
    idx = np.arange(nfeed)
    tel.feedmap = np.abs(idx[None, :] - idx[:, None])
    tel.feedconj = idx[:, None] > idx[None, :]
    tel.feedmask = np.ones((nfeed, nfeed), bool)
    for i in range(int(g["ncase"])):
        ids = g[f"c{i}_file_ids"]
        ninp = len(ids)
        inputs = np.array([(c,) for c in ids], dtype=[("chan_id", "<u2")])
        prod = np.array([(a, b) for a in range(ninp) for b in range(a, ninp)], dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        fm = np.zeros(len(g[f"c{i}_ffreq"]), dtype=[("centre", float), ("width", float)])
        fm["centre"], fm["width"] = g[f"c{i}_ffreq"], 10.0
        vis = g[f"c{i}_vis"]
        ss = containers.SiderealStream(freq=fm, ra=vis.shape[-1], input=inputs, prod=prod, stack=len(prod))
        ss.vis[:] = vis
        ss.weight[:] = g[f"c{i}_w"]
        ss.add_dataset("input_flags")
        ss.input_flags[:] = g[f"c{i}_flags"]
        t = CollateProducts(weight=str(g[f"c{i}_weight"]))
        t.setup(tel)
        sp = t.process(ss)
        assert isinstance(sp, containers.SiderealStream)
        np.testing.assert_allclose(sp.vis[:], g[f"c{i}_out_vis"], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(sp.weight[:], g[f"c{i}_out_w"], rtol=2e-6)
        assert np.array_equal(sp.input_flags[:], g[f"c{i}_out_flags"])
        assert np.array_equal(sp.index_map["stack"]["prod"], g[f"c{i}_out_stack_prod"])
        assert np.array_equal(sp.reverse_map["stack"]["stack"], g[f"c{i}_out_rev_stack"])
        assert np.array_equal(sp.index_map["freq"]["centre"], g["tel_freq"])
    # already redundancy-stacked inputs: representatives re-derived (input 999 absent, one telescope pair masked)
    for i in range(int(g["nstacked"])):
        ids = g[f"s{i}_file_ids"]
        ninp = len(ids)
        inputs = np.array([(c,) for c in ids], dtype=[("chan_id", "<u2")])
        prod = np.array([(a, b) for a in range(ninp) for b in range(a, ninp)], dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        stack = np.zeros(len(g[f"s{i}_stack_prod"]), dtype=[("prod", "<u4"), ("conjugate", "u1")])
        stack["prod"], stack["conjugate"] = g[f"s{i}_stack_prod"], g[f"s{i}_stack_conj"]
        rev = np.zeros(len(prod), dtype=[("stack", "<u4"), ("conjugate", "u1")])
        rev["stack"], rev["conjugate"] = g[f"s{i}_rev_stack"], g[f"s{i}_rev_conj"]
        fm = np.zeros(len(g["tel_freq"]), dtype=[("centre", float), ("width", float)])
        fm["centre"], fm["width"] = g["tel_freq"], 10.0
        vis = g[f"s{i}_vis"]
        ss2 = containers.SiderealStream(freq=fm, ra=vis.shape[-1], input=inputs, prod=prod, stack=stack, reverse_map_stack=rev)
        assert ss2.is_stacked
        ss2.vis[:] = vis
        ss2.weight[:] = g[f"s{i}_w"]
        ss2.add_dataset("input_flags")
        ss2.input_flags[:] = g[f"s{i}_flags"]
        tel.feedmask = np.ones((nfeed, nfeed), bool)
        if int(g[f"s{i}_mask_pair"]):
            tel.feedmask[0, 1] = tel.feedmask[1, 0] = False
        t2 = CollateProducts(weight=str(g[f"s{i}_weight"]))
        t2.setup(tel)
        sp = t2.process(ss2)
        np.testing.assert_allclose(sp.vis[:], g[f"s{i}_out_vis"], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(sp.weight[:], g[f"s{i}_out_w"], rtol=2e-6)
    tel.feedmask = np.ones((nfeed, nfeed), bool)
    # the telescope needs every one of its frequencies in the file
    tel.frequencies = np.array([400.0, 123.0])
    t.setup(tel)
    with pytest.raises(ValueError, match="Could not find all of the keys"):
        t.process(ss)


def test_collate_with_package_telescope_feeds_mapmaker():
    """Full-triangle stream of the package's own telescope -> CollateProducts -> stacked stream of npairs baselines."""
    from draco_amd.analysis.transform import CollateProducts
    from draco_amd.core import containers
    from draco_amd.core.products import TransitTelescope

    tel = TransitTelescope(np.array([400.0, 410.0]), lmax=4, ncyl=2, nfeed_cyl=2)
    rng = np.random.default_rng(1)
    nprod = len(tel.index_map_prod)
    true = rng.standard_normal((2, tel.npairs, 5)) + 1j * rng.standard_normal((2, tel.npairs, 5))
    # every product carries its stack's value (conjugated where the map says so): stacking must give it back
    rev = tel.reverse_map_stack
    vis = np.where(rev["conjugate"][None, :, None].astype(bool), true[:, rev["stack"]].conj(), true[:, rev["stack"]]).astype(np.complex64)
    ss = containers.SiderealStream(freq=tel.frequencies, ra=5, input=tel.input_index, prod=tel.index_map_prod, stack=nprod)
    ss.vis[:] = vis
    ss.weight[:] = 1.0
    t = CollateProducts(weight="natural")
    t.setup(tel)
    sp = t.process(ss)
    assert sp.vis.shape == (2, tel.npairs, 5)
    assert _rel(sp.vis[:], true.astype(np.complex64)) < 1e-6
    # weight = (sum w)^2 / sum(w^2 / 1) = redundancy for unit weights
    np.testing.assert_allclose(sp.weight[:][0, :, 0], tel.redundancy, rtol=1e-6)


def test_expand_products_golden(golden_dir):
    """ExpandProducts (the inverse data-format step of CollateProducts) against the reference's own outputs, bit exact;
    and CollateProducts undoes it."""
    import os

    from draco_amd.analysis.transform import CollateProducts
    from draco_amd.core import containers
    from draco_amd.synthesis.stream import ExpandProducts

    g = np.load(os.path.join(golden_dir, "stream_expand.npz"))
    for i in range(int(g["ncase"])):
        nfeed = int(g[f"c{i}_nfeed"])

        class Tel:
            pass

        tel = Tel()
        tel.nfeed, tel.npairs = nfeed, nfeed
        tel.lmax = tel.mmax = 1
        tel.frequencies = np.array([400.0, 410.0, 420.0])
        tel.input_index = np.array([(100 + k,) for k in range(nfeed)], dtype=[("chan_id", "<u2")])
        tel.uniquepairs = np.array([(0, d) for d in range(nfeed)])
        idx = np.arange(nfeed)
        tel.feedmap = np.abs(idx[None, :] - idx[:, None])
        tel.feedconj = idx[:, None] > idx[None, :]
        tel.feedmask = np.ones((nfeed, nfeed), bool)
        if int(g[f"c{i}_mask"]):
            tel.feedmap[1, 3] = tel.feedmap[3, 1] = -1
            tel.feedmask[1, 3] = tel.feedmask[3, 1] = False
        vis = g[f"c{i}_vis"]
        ss = containers.SiderealStream(freq=tel.frequencies, ra=vis.shape[-1], input=tel.input_index, stack=nfeed)
        ss.vis[:] = vis
        ss.weight[:] = 1.0
        t = ExpandProducts()
        t.setup(tel)
        full = t.process(ss)
        assert np.array_equal(full.vis[:], g[f"c{i}_out_vis"])
        assert np.array_equal(full.weight[:], g[f"c{i}_out_w"])
        assert len(full.index_map["prod"]) == nfeed * (nfeed + 1) // 2
        # round trip: stacking the expanded products gives the unique baselines back
        c = CollateProducts(weight="natural")
        c.setup(tel)
        back = c.process(full)
        np.testing.assert_allclose(back.vis[:], vis, rtol=2e-6, atol=1e-6)


def test_single_polarisation_dense_solvers():
    """num_pol_sky = 1 through the dense solvers.  ML: nsky_m = lmax+1-m is not a multiple of four (the generic operand
    path), telescope- and sky-side systems both occur (ntel = 22, nsky_m from 31 down to 1).  Wiener: the reference's
    prior is hard-wired to four polarisations (mapmaker.py:264) and raises ValueError on its first solve; so do we."""
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    lmax = 30
    tel = _tel(2, lmax, num_pol_sky=1)
    bt = SyntheticProvider(tel, seed=8)
    rng = np.random.default_rng(8)
    shape = (lmax + 1, 2, 2, tel.npairs)
    mv = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    mw = rng.uniform(0.5, 1.5, shape) * 10.0
    mw[rng.uniform(size=shape) < 0.1] = 0.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    task = MaximumLikelihoodMapMaker()
    task.setup(bt)
    alm = task.alm_square(task.make_alm(mm))
    ref = omm.solve_alm("ml", lambda m, f: osyn.beam_tile(8, m, f, tel.npairs, 1, lmax), mv, mw, lmax, lmax, [0, 1], npol=1)
    ref[:, 1:] = ref[:, :1]
    assert _rel(alm, ref) < 1e-8
    w = WienerMapMaker()
    w.setup(bt)
    with pytest.raises(ValueError, match="could not be broadcast"):
        w.make_alm(mm)
    with pytest.raises(ValueError, match="could not be broadcast"):
        w._solve_m(0, 0, mv[0, :, 0], mw[0, :, 0])


def test_allgather_map_through_the_c_abi_on_a_one_rank_communicator():
    """`dmm_allgather_map` (the one collective, behind the C ABI): RCCL is loaded at run time, a
    communicator is made from an id (`dmm_comm_unique_id` / `dmm_comm_init`), and the all-gather runs on the
    context's stream.  One GPU here, so one rank: the gathered map is the shard.  (Several ranks: the same calls, the id
    handed round by the caller; the Python layer's gather over torch.distributed is covered in test_dist_gloo.py.)"""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    lib = _lib.lib
    ident = (C.c_char * 128)()
    _lib.check(lib.dmm_comm_unique_id(ident))
    assert bytes(ident) != bytes(128)
    comm = C.c_void_p()
    _lib.check(lib.dmm_comm_init(ctx.handle, ident, 0, 1, C.byref(comm)))
    try:
        shard = torch.randn((3, 4, 12 * 16 * 16), dtype=torch.float64, device=ctx.device)
        full = torch.zeros_like(shard)
        _lib.check(lib.dmm_allgather_map(ctx.handle, comm, ptr(shard), shard.numel(), ptr(full)))
        ctx.sync()
        assert torch.equal(full, shard)
    finally:
        _lib.check(lib.dmm_comm_destroy(comm))


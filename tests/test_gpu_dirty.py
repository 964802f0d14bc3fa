"""GPU parity: batched Dirty solves and the forward projection vs the oracle / golden vectors.

The contraction accumulates in float64 like the reference (complex128 np.dot); only the
summation order differs, so complex128 B agrees to ~1e-14 relative; we assert 1e-12.
complex64 B storage rounds B to 24 bits: asserted 3e-7 relative.
"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import synth as osyn


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _tel(nfreq, lmax, ncyl=1, nfeed_cyl=3, npairs=None, mmax=None):
    from draco_amd.core.products import TransitTelescope

    return TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, mmax=mmax, ncyl=ncyl, nfeed_cyl=nfeed_cyl, npairs=npairs)


def test_synth_fill_bit_exact():
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import Slab
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(3, 9)
    bt = SyntheticProvider(tel, seed=77)
    ms = np.array([0, 3, 9, 5], np.int32)
    fs = np.array([0, 2, 1, 1], np.int32)
    for layout in (_lib.DMM_B_PACKED, _lib.DMM_B_FULL):
        for dt, npdt in ((_lib.DMM_C128, np.complex128), (_lib.DMM_C64, np.complex64)):
            slab = Slab(ctx, bt, ms, fs, fs, dt, layout, 3, 10)
            pool = slab.pool.cpu().numpy()
            for i in range(4):
                ref = osyn.beam_tile(77, ms[i], fs[i], tel.npairs, 4, 9).reshape(bt.ntel, 4, 10)
                if layout == _lib.DMM_B_PACKED:
                    ref = ref[..., ms[i]:]
                got = pool[slab.tiles[i].b_off : slab.tiles[i].b_off + ref.size].reshape(ref.shape)
                assert np.array_equal(got, ref.astype(npdt)), (layout, dt, i)
            slab.close()


def test_solve_m_golden(golden_dir):
    """DirtyMapMaker._solve_m against the reference's outputs on the reference's inputs."""
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core.products import ArrayProvider

    g = np.load(os.path.join(golden_dir, "mapmaker_solve_m.npz"))
    for i in range(int(g["ncase"])):
        bm, m, v, Ni = g[f"c{i}_bm"], int(g[f"c{i}_m"]), g[f"c{i}_v"], g[f"c{i}_Ni"]
        npairs, lmax = bm.shape[1], bm.shape[3] - 1
        tel = _tel(3, lmax, npairs=npairs)
        task = DirtyMapMaker()
        task.setup(ArrayProvider(tel, lambda mm, ff, bm=bm: bm))
        a = task._solve_m(m, 1, v, Ni)
        ref = g[f"c{i}_dirty"]
        assert a.shape == ref.shape
        assert _rel(a, ref) < 1e-12, f"case {i}"
        assert np.all(a[:, :m] == 0)


@pytest.mark.parametrize("b_dtype,tol", [("complex128", 1e-12), ("complex64", 3e-7)])
@pytest.mark.parametrize("nfreq,lmax,ncyl,nfeed", [(3, 12, 1, 3), (2, 70, 1, 5), (4, 33, 2, 2)])
def test_dirty_alm_vs_oracle(b_dtype, tol, nfreq, lmax, ncyl, nfeed):
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    tel = _tel(nfreq, lmax, ncyl, nfeed)
    bt = SyntheticProvider(tel, seed=3000 + lmax)
    rng = np.random.default_rng(lmax)
    n_m = lmax + 3  # more m rows in the data than the telescope's mmax: trimmed (mapmaker.py:52)
    mv = rng.standard_normal((n_m, 2, nfreq, tel.npairs)) + 1j * rng.standard_normal((n_m, 2, nfreq, tel.npairs))
    mw = rng.uniform(0.5, 1.5, (n_m, 2, nfreq, tel.npairs))
    mw[rng.uniform(size=mw.shape) < 0.1] = 0.0
    mm = containers.MModes(mmax=n_m - 1, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    task = DirtyMapMaker(b_dtype=b_dtype)
    task.setup(bt)
    alm = task.alm_square(task.make_alm(mm))
    ref = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(3000 + lmax, m, f, tel.npairs, 4, lmax), mv, mw, lmax, tel.mmax, list(range(nfreq)))
    assert alm.shape == ref.shape == (nfreq, 4, lmax + 1, lmax + 1)
    assert _rel(alm, ref) < tol
    assert np.array_equal(alm == 0, ref == 0)  # l<m and m>l structure identical


def test_dirty_slabbed_and_freq_subset():
    """A pool budget that forces several slabs, data frequencies a permuted subset of the telescope's."""
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider

    tel = _tel(5, 20, 1, 3)
    bt = SyntheticProvider(tel, seed=11)
    sel = [3, 0, 4]
    rng = np.random.default_rng(1)
    mv = rng.standard_normal((21, 2, 3, tel.npairs)) + 1j * rng.standard_normal((21, 2, 3, tel.npairs))
    mw = rng.uniform(0.5, 1.5, (21, 2, 3, tel.npairs))
    mm = containers.MModes(mmax=20, freq=tel.frequencies[sel], stack=tel.npairs)
    mm.vis[:] = mv
    mm.weight[:] = mw
    task = DirtyMapMaker(pool_bytes=200_000)
    task.setup(bt)
    alm = task.alm_square(task.make_alm(mm))
    ref = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(11, m, f, tel.npairs, 4, 20), mv, mw, 20, 20, sel)
    assert _rel(alm, ref) < 1e-12
    bad = containers.MModes(mmax=20, freq=[123.0, 400.0, 480.0], stack=tel.npairs)
    with pytest.raises(ValueError, match="Could not find all of the keys"):
        task.make_alm(bad)


def test_full_layout_matches_packed():
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(2, 40, 1, 4)
    bt = SyntheticProvider(tel, seed=5)
    gen = torch.Generator(device=ctx.device).manual_seed(1)
    mv = torch.randn((41, 2, 2, tel.npairs), dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand((41, 2, 2, tel.npairs), dtype=torch.float64, device=ctx.device, generator=gen)
    outs = []
    for dt in (_lib.DMM_C128, _lib.DMM_C64):
        for layout in (_lib.DMM_B_PACKED, _lib.DMM_B_FULL):
            eng = SolveEngine(bt, ctx, dt, layout)
            outs.append(eng.solve("dirty", mv, mw, [0, 1], 40).cpu().numpy())
    assert np.array_equal(outs[0], outs[1])  # same arithmetic order: bit-identical
    assert _rel(outs[2], outs[3]) < 1e-15
    assert _rel(outs[2], outs[0]) < 3e-7


def test_unit_vector_reads_back_B_rows_cfg3_tile():
    """Size-independent property at cfg-3 tile size: v = e_i, Ni = 1  =>  a = conj(B[i, :]) exactly."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis._solve import SolveEngine
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(2, 512, 2, 32)  # 128 feeds, lmax 512: the cfg-3 tile 758 x 2052
    assert tel.npairs == 379
    bt = SyntheticProvider(tel, seed=3003)
    ms = [0, 1, 255, 511, 512]
    n_m = 513
    mv = torch.zeros((n_m, 2, 2, 379), dtype=torch.complex128, device=ctx.device)
    mw = torch.ones((n_m, 2, 2, 379), dtype=torch.float64, device=ctx.device)
    rows = {}
    for m in ms:
        i = (m * 37 + 5) % 758
        rows[m] = i
        mv[m, i // 379, :, i % 379] = 1.0
    eng = SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, pool_bytes=30 * 2**30)
    alm = eng.solve("dirty", mv, mw, [0, 1], 512)
    for m in ms:
        for f in (0, 1):
            ref = np.conj(osyn.beam_tile(3003, m, f, 379, 4, 512).reshape(758, 4, 513)[rows[m]])
            got = alm[f, :, m, :].cpu().numpy()
            assert np.array_equal(got, ref), (m, f)
    # linearity at the same size
    gen = torch.Generator(device=ctx.device).manual_seed(2)
    v1 = torch.randn(mv.shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    v2 = torch.randn(mv.shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    a1, a2 = eng.solve("dirty", v1, mw, [0, 1], 512), eng.solve("dirty", v2, mw, [0, 1], 512)
    a12 = eng.solve("dirty", v1 + v2, mw, [0, 1], 512)
    assert ((a12 - a1 - a2).abs().max() / a12.abs().max()).item() < 1e-13


def test_project_vs_oracle():
    from draco_amd.core.products import SyntheticProvider

    tel = _tel(3, 14, 1, 3)
    bt = SyntheticProvider(tel, seed=9)
    rng = np.random.default_rng(0)
    for mi in (0, 5, 14):
        vec = rng.standard_normal((3, 4, 15)) + 1j * rng.standard_normal((3, 4, 15))
        vec[..., :mi] = 0
        out = bt.project_vector_sky_to_telescope(mi, vec)
        ref = np.stack([osyn.beam_tile(9, mi, f, tel.npairs, 4, 14).reshape(bt.ntel, -1) @ vec[f].reshape(-1) for f in range(3)])
        assert out.shape == (3, bt.ntel)
        assert _rel(out, ref) < 1e-13


@pytest.mark.parametrize("b_dtype", ["c128", "c64"])
def test_slabs_share_one_pool_sized_for_the_largest(b_dtype):
    """Slabs are cut where the budget is reached, so their sizes differ by a few tiles and a later one may be larger
    than the first: the pool is allocated once, for the largest (growing it would need a second pool-sized block while
    the first is alive -- 168 GB twice at cfg 3)."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis import _solve
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import Context

    ctx = Context.get()
    tel = _tel(3, 30, 1, 4)
    bt = SyntheticProvider(tel, seed=21)
    gen = torch.Generator(device=ctx.device).manual_seed(2)
    shape = (31, 2, 3, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen)
    dt = _lib.DMM_C128 if b_dtype == "c128" else _lib.DMM_C64
    ref = _solve.SolveEngine(bt, ctx, dt, _lib.DMM_B_PACKED).solve("dirty", mv, mw, [0, 1, 2], 30).cpu().numpy()
    eng = _solve.SolveEngine(bt, ctx, dt, _lib.DMM_B_PACKED, pool_bytes=400_000 if b_dtype == "c128" else 200_000, cache=False)
    pools, sizes = set(), []
    for slab in eng.slabs([0, 1, 2], 30, 3, 31):
        pools.add(slab.pool.data_ptr())
        sizes.append(slab.nelem)
        assert slab.nelem <= slab.pool.numel()
    assert len(sizes) >= 4 and len(pools) == 1
    assert max(sizes[1:]) > sizes[0]  # the case that used to re-allocate
    assert np.array_equal(eng.solve("dirty", mv, mw, [0, 1, 2], 30).cpu().numpy(), ref)

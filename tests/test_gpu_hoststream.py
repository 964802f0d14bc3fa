"""B tiles from host memory: the staged / double-buffered upload path and the pool-residency bookkeeping.

The reference reads one tile per ``_solve_m`` call (mapmaker.py:160-162).  Here a provider hands tiles over in bulk
(``beam_block``) and the engine overlaps the upload of slab k+1 with the solves of slab k.  Checked: results are
bit-identical to the device-generated pool (same tile contents, same kernels) whatever the route -- pinned store
(direct copies), pageable store (staged by worker threads), per-tile ``beam_m`` providers, complex64 wire format,
pool budgets that force many slabs and both buffers -- and that resident contents are not uploaded twice.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import synth as osyn


def _setup(nfreq=6, lmax=40, nfeed_cyl=6, seed=71):
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider, TransitTelescope

    tel = TransitTelescope(osyn.frequencies(nfreq), lmax=lmax, ncyl=1, nfeed_cyl=nfeed_cyl)
    bt = SyntheticProvider(tel, seed=seed)
    rng = np.random.default_rng(seed)
    shape = (lmax + 1, 2, nfreq, tel.npairs)
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs)
    mm.vis[:] = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    w = rng.uniform(0.5, 1.5, shape)
    w[rng.uniform(size=shape) < 0.05] = 0
    mm.weight[:] = w
    return tel, bt, mm


def _alm(task_cls, bt, mm, **attrs):
    t = task_cls(**attrs)
    t.setup(bt)
    a = t.make_alm(mm).cpu().numpy()
    return a, t._engine


@pytest.mark.parametrize("b_dtype", ["complex128", "complex64"])
@pytest.mark.parametrize("pinned", [True, False])
def test_packed_store_streams_bit_identically(b_dtype, pinned):
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core.products import PackedStoreProvider
    from draco_amd.device import Context

    tel, bt, mm = _setup()
    ref, _ = _alm(DirtyMapMaker, bt, mm, b_dtype=b_dtype)
    npdt = np.complex128 if b_dtype == "complex128" else np.complex64
    store = PackedStoreProvider.from_provider(bt, Context.get(), npdt, pin=pinned)
    assert store.block_is_pinned == pinned and store.store.dtype == npdt
    # the store serves the reference-visible per-tile call too
    np.testing.assert_array_equal(store.beam_m(7, fi=3), bt.beam_m(7, fi=3).astype(npdt))
    per_freq_bytes = store.per_freq * np.dtype(npdt).itemsize
    # two buffers of ~1.4 frequencies each: 6 frequencies -> >= 5 slabs, partial-frequency runs, both buffers in use
    out, eng = _alm(DirtyMapMaker, store, mm, b_dtype=b_dtype, pool_bytes=int(2.8 * per_freq_bytes))
    assert eng.nbuf == 2 and eng.fills >= 5
    assert np.array_equal(out, ref)
    _solve.release_pools()


def test_per_tile_provider_through_the_staging_ring_vs_oracle():
    """A provider with nothing but ``beam_m`` (what a driftscan BeamTransfer offers): generic ``beam_block`` packs
    tile by tile on the worker threads."""
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker
    from draco_amd.core.products import ArrayProvider, ForeignProvider

    tel, bt, mm = _setup(nfreq=3, lmax=24, nfeed_cyl=4, seed=72)
    calls = []

    def beam(m, f):
        calls.append((m, f))
        return osyn.beam_tile(72, m, f, tel.npairs, 4, tel.lmax)

    prov = ArrayProvider(tel, beam)
    tile_bytes = 2 * tel.npairs * 4 * (tel.lmax + 1) * 16
    out, eng = _alm(DirtyMapMaker, prov, mm, pool_bytes=20 * tile_bytes)
    assert eng.fills >= 3 and sorted(calls) == sorted((m, f) for m in range(25) for f in range(3))  # every tile fetched once
    ref = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(72, m, f, tel.npairs, 4, tel.lmax), mm.vis[:], mm.weight[:], tel.lmax, tel.mmax, [0, 1, 2])
    got = np.zeros_like(ref)
    got[:, :, :, : tel.mmax + 1] = out.transpose(0, 1, 3, 2)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()

    class Foreign:  # any object with telescope + beam_m + ntel/nsky
        telescope = tel
        ntel, nsky = 2 * tel.npairs, 4 * (tel.lmax + 1)

        def beam_m(self, m, fi=None):
            return osyn.beam_tile(72, m, fi, tel.npairs, 4, tel.lmax)

    fp = ForeignProvider(Foreign())
    assert fp.stage_workers == 1
    out2, _ = _alm(DirtyMapMaker, fp, mm, pool_bytes=20 * tile_bytes)
    assert np.array_equal(out2, out)
    _solve.release_pools()


def test_resident_contents_are_filled_once_and_pool_policy_provider():
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import DirtyMapMaker, WienerMapMaker
    from draco_amd.core.products import PoolCycledProvider, SyntheticProvider

    _solve.release_pools()
    tel, bt, mm = _setup(nfreq=8, lmax=30, nfeed_cyl=5, seed=73)
    t = DirtyMapMaker()
    t.setup(bt)
    a1 = t.make_alm(mm).cpu().numpy()
    assert t._engine.fills == 1
    a2 = t.make_alm(mm).cpu().numpy()  # the same day again: the slab is still resident
    assert t._engine.fills == 1 and np.array_equal(a1, a2)
    w = WienerMapMaker()  # another task, an equal provider: same contents, no fill
    w.setup(SyntheticProvider(tel, seed=73))
    w.make_alm(mm)
    assert w._engine.fills == 0
    w2 = DirtyMapMaker()  # a different seed is different contents
    w2.setup(SyntheticProvider(tel, seed=74))
    w2.make_alm(mm)
    assert w2._engine.fills == 1

    # hbm-pool policy: 8 frequencies cycling through 2 frequencies' worth of distinct tiles
    cyc = PoolCycledProvider(bt, 2)
    per_freq = sum(2 * tel.npairs * 4 * (tel.lmax + 1 - m) for m in range(tel.lmax + 1)) * 16
    tc = DirtyMapMaker(pool_bytes=int(2.5 * per_freq))
    tc.setup(cyc)
    ac = tc.make_alm(mm).cpu().numpy()
    assert tc._engine.fills == 1  # four slabs of two frequencies, one fill
    ac2 = tc.make_alm(mm).cpu().numpy()
    assert tc._engine.fills == 1 and np.array_equal(ac, ac2)
    # same numbers as solving every frequency against the tiles of f % 2
    ref = omm.solve_alm("dirty", lambda m, f: osyn.beam_tile(73, m, f % 2, tel.npairs, 4, tel.lmax), mm.vis[:], mm.weight[:], tel.lmax, tel.mmax, list(range(8)))
    got = np.zeros_like(ref)
    got[:, :, :, : tel.mmax + 1] = ac.transpose(0, 1, 3, 2)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    np.testing.assert_array_equal(cyc.beam_m(3, fi=5), bt.beam_m(3, fi=1))
    _solve.release_pools()


def test_wiener_and_ml_over_streamed_slabs():
    """The dense solvers consume double-buffered slabs too (their workspaces are per slab)."""
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker, WienerMapMaker
    from draco_amd.core.products import PackedStoreProvider
    from draco_amd.device import Context

    tel, bt, mm = _setup(nfreq=4, lmax=20, nfeed_cyl=4, seed=75)
    store = PackedStoreProvider.from_provider(bt, Context.get(), np.complex128, pin=True)
    per_freq_bytes = store.per_freq * 16
    for cls, tol in ((WienerMapMaker, 1e-12), (MaximumLikelihoodMapMaker, 1e-9)):
        ref, _ = _alm(cls, bt, mm)
        out, eng = _alm(cls, store, mm, pool_bytes=int(2.5 * per_freq_bytes))
        assert eng.fills >= 3
        assert np.abs(out - ref).max() <= tol * np.abs(ref).max()
    _solve.release_pools()


def test_stager_rejects_oversized_chunks_before_any_producer_runs_and_drains_on_failure():
    """`HostStager.upload`: a staged chunk larger than a slot is refused before a single producer has been
    started (a producer must never be handed a truncated view), and when a producer raises, no other producer of the
    call is still writing into ring slots after `upload` has returned."""
    import threading
    import time

    import torch

    from draco_amd.core.hoststage import HostStager

    st = HostStager(torch.device("cuda", 0), slot_bytes=1 << 16, workers=4)
    try:
        pool = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.Stream()
        calls = []

        def ok(out):
            calls.append(out.size)
            out[:] = 7

        with pytest.raises(ValueError, match="exceeds the slot size"):
            st.upload([(0, 1 << 15, ok), (1 << 15, (1 << 16) + 1, ok)], pool, stream)
        assert calls == []

        running = []
        lock = threading.Lock()

        def slow(out):
            with lock:
                running.append(1)
            time.sleep(0.05)
            out[:] = 1
            with lock:
                running.pop()

        def bad(out):
            raise RuntimeError("producer failed")

        jobs = [(k << 12, 1 << 12, bad if k == 1 else slow) for k in range(12)]
        with pytest.raises(RuntimeError, match="producer failed"):
            st.upload(jobs, pool, stream)
        assert running == []  # everything that had been started has finished; the rest was cancelled
        # and the stager still works
        st.upload([(0, 1 << 12, ok)], pool, stream)
        stream.synchronize()
        assert int(pool[: 1 << 12].sum()) == 7 * (1 << 12)
    finally:
        st._pool.shutdown(wait=True)

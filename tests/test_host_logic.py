"""Host-side logic that needs no GPU: containers, adapters, telescope, provider protocol."""

import numpy as np
import pytest

from draco_amd.core import containers, io
from draco_amd.core.products import ArrayProvider, ProductManager, SyntheticProvider, TransitTelescope, synth_beam_tile
from draco_amd.util import tools
from oracle import synth as osyn


def test_find_keys_and_invert():
    assert tools.find_keys([400.0, 401.0, 402.0], [402.0, 400.0]) == [2, 0]
    with pytest.raises(ValueError, match="Could not find all of the keys"):
        tools.find_keys([400.0], [401.0], require_match=True)
    x = np.array([0.0, 2.0, -4.0], np.float32)
    r = tools.invert_no_zero(x)
    assert r.dtype == np.float32 and np.array_equal(r, np.array([0.0, 0.5, -0.25], np.float32))


def test_telescope_pairs_match_survey_formula():
    for ncyl, nf in ((1, 8), (2, 16), (2, 4)):
        tel = TransitTelescope(np.arange(3.0), lmax=4, ncyl=ncyl, nfeed_cyl=nf)
        assert tel.npairs == osyn.npairs_of(ncyl, nf)
        assert tel.nfeed == ncyl * nf * 2
        assert len(tel.index_map_prod) == tel.nfeed * (tel.nfeed + 1) // 2
        assert tel.uniquepairs.shape == (tel.npairs, 2)
        # every product maps to a stack entry and back
        rev = tel.reverse_map_stack
        assert rev["stack"].max() == tel.npairs - 1
        st = tel.index_map_stack
        assert np.all(rev["stack"][st["prod"]] == np.arange(tel.npairs))


def test_adapters():
    tel = TransitTelescope(np.arange(3.0), lmax=4, ncyl=1, nfeed_cyl=2)
    bt = SyntheticProvider(tel)
    assert io.get_beamtransfer(bt) is bt
    assert io.get_beamtransfer(ProductManager(bt)) is bt
    assert io.get_telescope(bt) is tel and io.get_telescope(tel) is tel
    with pytest.raises(RuntimeError, match="Could not get BeamTransfer instance"):
        io.get_beamtransfer(object())
    with pytest.raises(RuntimeError, match="Could not get telescope instance"):
        io.get_telescope(42)

    class Foreign:  # looks like driftscan's BeamTransfer
        telescope = tel
        ntel, nsky = 2 * tel.npairs, 4 * 5

        def beam_m(self, m, fi=None):
            return np.zeros((2, tel.npairs, 4, 5), complex)

    f = io.get_beamtransfer(Foreign())
    assert f.ntel == 2 * tel.npairs and f.beam_m(0, fi=0).shape == (2, tel.npairs, 4, 5)


def test_synthetic_tiles_twin():
    a = osyn.beam_tile(3000, 3, 2, 7, 4, 9)
    b = synth_beam_tile(3000, 3, 2, 7, 4, 9)
    assert np.array_equal(a, b)
    assert np.all(a[..., :3] == 0) and np.all(a[..., 3:] != 0)
    assert abs(a[..., 3:].var() * 14 - 1) < 0.1
    assert not np.array_equal(a, osyn.beam_tile(3000, 3, 1, 7, 4, 9))


def test_containers_axes_and_dtypes():
    tel = TransitTelescope(np.linspace(400, 800, 5, endpoint=False), lmax=4, ncyl=1, nfeed_cyl=2)
    ss = containers.SiderealStream(freq=tel.frequencies, ra=16, stack=tel.npairs)
    assert ss.vis.shape == (5, tel.npairs, 16) and ss.vis.dtype == np.complex64
    assert ss.weight.dtype == np.float32
    assert ss.ra[1] == 360.0 / 16
    ss.attrs["tag"] = "x"
    mm = containers.MModes(mmax=8, oddra=False, axes_from=ss, attrs_from=ss)
    assert mm.vis.shape == (9, 2, 5, tel.npairs) and mm.vis.dtype == np.complex128 and mm.weight.dtype == np.float64
    assert mm.mmax == 8 and mm.oddra is False and mm.attrs["tag"] == "x"
    assert list(mm.index_map["msign"]) == ["+", "-"]
    assert np.array_equal(mm.index_map["freq"]["centre"], tel.frequencies)
    mp = containers.Map(nside=4, axes_from=mm)
    assert mp.map.shape == (5, 4, 192) and mp.map.dtype == np.float64 and mp.nside == 4
    ss.vis[0, 0, 0] = 1 + 2j
    assert ss.vis[:][0, 0, 0] == np.complex64(1 + 2j)


def test_array_provider():
    tel = TransitTelescope(np.arange(2.0), lmax=3, ncyl=1, nfeed_cyl=1)
    tiles = {(m, f): osyn.beam_tile(1, m, f, tel.npairs, 4, 3) for m in range(4) for f in range(2)}
    bt = ArrayProvider(tel, lambda m, f: tiles[(m, f)])
    assert bt.ntel == 2 * tel.npairs and bt.nsky == 16
    assert bt.beam_m(2).shape == (2, 2, tel.npairs, 4, 4)
    assert np.array_equal(bt.beam_m(1, fi=1), tiles[(1, 1)])


def test_task_config():
    from draco_amd.analysis.mapmaker import WienerMapMaker
    from draco_amd.analysis.transform import MModeTransform

    t = MModeTransform()
    assert t.remove_integration_window is False and t.use_fftw is True
    w = WienerMapMaker(prior_amp=2, nside=64)
    assert w.prior_amp == 2.0 and isinstance(w.prior_amp, float) and w.prior_tilt == 0.5 and w.nside == 64
    with pytest.raises(AttributeError):
        WienerMapMaker(bogus=1)


def test_workloads_match_the_oracles_table():
    """bench.py / tools take the BASELINE workloads from the package; the checker keeps its own copy."""
    from draco_amd import workloads
    from oracle import synth as osyn

    assert workloads.CONFIGS == osyn.CONFIGS
    assert np.array_equal(workloads.frequencies(64), osyn.frequencies(64))


def test_oracle_is_imported_only_where_it_may_be():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/ (test infrastructure)."""
    import ast
    import pathlib

    root = pathlib.Path(__file__).resolve().parent.parent

    def oracle_imports(path):
        tree = ast.parse(path.read_text())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
                hits.append(node.lineno)
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                hits.append(node.lineno)
        return hits

    for path in list((root / "draco_amd").rglob("*.py")) + list((root / "tools").rglob("*.py")):
        assert not oracle_imports(path), path
    # bench.py: inside the cpu_baseline legs (Dirty; ML / Wiener) and their workers only
    tree = ast.parse((root / "bench.py").read_text())
    allowed = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in ("cpu_baseline", "_cpu_worker", "cpu_baseline_dense", "_cpu_dense_worker"):
            allowed.update(range(node.lineno, node.end_lineno + 1))
    assert all(ln in allowed for ln in oracle_imports(root / "bench.py"))
    tree = ast.parse((root / "__graft_entry__.py").read_text())
    allowed = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "smoke":
            allowed.update(range(node.lineno, node.end_lineno + 1))
    assert all(ln in allowed for ln in oracle_imports(root / "__graft_entry__.py"))


def test_container_save_load_round_trip(tmp_path):
    """Stage outputs can be written and read back (the reference's save / resume-from-file pattern)."""
    from draco_amd.core import containers

    rng = np.random.default_rng(0)
    prod = np.array([(0, 0), (0, 1), (1, 1)], dtype=[("input_a", "<u2"), ("input_b", "<u2")])
    ss = containers.SiderealStream(freq=[400.0, 410.0], ra=6, prod=prod, input=2)
    ss.vis[:] = (rng.standard_normal(ss.vis.shape) + 1j * rng.standard_normal(ss.vis.shape)).astype(np.complex64)
    ss.weight[:] = rng.uniform(size=ss.weight.shape).astype(np.float32)
    ss.attrs["tag"] = "day 12"
    f = tmp_path / "sstream.npz"
    ss.save(f)
    back = containers.SiderealStream.load(f)
    assert np.array_equal(back.vis[:], ss.vis[:]) and back.vis.dtype == np.complex64
    assert np.array_equal(back.weight[:], ss.weight[:]) and back.attrs["tag"] == "day 12"
    assert np.array_equal(back.index_map["freq"]["centre"], [400.0, 410.0]) and np.array_equal(back.index_map["prod"], prod)
    mm = containers.MModes(mmax=3, oddra=True, freq=[400.0], stack=2)
    mm.vis[:] = 1.5 - 2j
    g = tmp_path / "mmodes"
    mm.save(g)
    any_back = containers.ContainerBase.load(g)  # class recovered from the file
    assert isinstance(any_back, containers.MModes) and any_back.oddra and any_back.mmax == 3
    assert np.array_equal(any_back.vis[:], mm.vis[:])
    import pytest

    with pytest.raises(TypeError):
        containers.Map.load(g)


def test_dataset_pending_orders_every_reader_until_the_producer_has_finished():
    """A producer may hand out a dataset whose device copy is still being written on another stream, with the object
    that orders a reader behind it (containers.Dataset `pending`: `done()` / `order()`): shape / dtype queries must not
    trigger it; EVERY access to the data orders the stream current at that moment while the producer is still running
    (a second reader on another stream is ordered too), and none once it has finished."""
    import torch

    from draco_amd.core import containers

    class Producer:
        finished = False
        orders = 0

        def done(self):
            return self.finished

        def order(self):
            self.orders += 1

    prod = Producer()
    t = torch.arange(6, dtype=torch.float64).reshape(2, 3)
    ds = containers.Dataset(dev=t, pending=prod)
    assert ds.shape == (2, 3) and ds.dtype == np.float64 and ds.on_device
    assert prod.orders == 0
    ds._dev
    ds._dev  # (a second reader, possibly on another stream)
    assert prod.orders == 2
    prod.finished = True
    assert ds[1, 2] == 5.0
    np.asarray(ds)
    ds._dev
    assert prod.orders == 2 and ds._pending is None


def _ss_container():
    from draco_amd.core import containers

    ss = containers.SiderealStream(stack=5, input=3, ra=16, freq=np.linspace(800.0, 750.0, 5))
    ss.attrs["test_attr1"] = "hello"
    ss.vis.attrs["test_attr2"] = "hello2"
    ss.weight.attrs["test_attr3"] = "hello3"
    ss.index_attrs["freq"]["alignment"] = 1
    return ss


def test_containers_attrs_from_like_the_reference():
    """The reference's own test of the constructor protocol (test/test_containers.py:25-39), on the containers of this
    path: container, dataset and axis attributes travel with ``attrs_from``; a dataset's ``axis`` tuple is its own."""
    from draco_amd.core import containers

    ss = _ss_container()
    other = containers.SiderealStream(ra=10, axes_from=ss, attrs_from=ss)
    assert len(other.attrs) == 1 and other.attrs["test_attr1"] == "hello"
    assert other.vis.attrs["test_attr2"] == "hello2"
    assert other.weight.attrs["test_attr3"] == "hello3"
    assert other.index_attrs["freq"]["alignment"] == 1
    assert len(other.vis.attrs) == 2 and len(other.weight.attrs) == 2
    assert tuple(other.vis.attrs["axis"]) == ("freq", "stack", "ra")
    assert tuple(other.weight.attrs["axis"]) == ("freq", "stack", "ra")
    assert other.vis.shape == (5, 5, 10)
    mm = containers.MModes(mmax=4, axes_from=ss, attrs_from=ss)  # another container class: attributes by dataset NAME
    assert mm.attrs["test_attr1"] == "hello" and mm.vis.attrs["test_attr2"] == "hello2"
    assert tuple(mm.vis.attrs["axis"]) == ("m", "msign", "freq", "stack")


def test_containers_copy_shared_like_the_reference():
    """test/test_containers.py:42-76 (the single-process part): ``copy(shared=("vis",))`` shares that dataset -- data and
    attributes -- and copies the others."""
    ss = _ss_container()
    ss.vis[:] = np.arange(16)
    ss.weight[:] = np.arange(16)
    cp = ss.copy(shared=("vis",))
    assert (cp.vis[:] == ss.vis[:]).all() and (cp.weight[:] == ss.weight[:]).all()
    assert cp.attrs["test_attr1"] == "hello" and cp.vis.attrs["test_attr2"] == "hello2"
    assert cp.weight.attrs["test_attr3"] == "hello3" and cp.index_attrs["freq"]["alignment"] == 1
    ss.vis[:] = 1.0
    ss.weight[:] = 2.0
    ss.vis.attrs["test_attr4"] = "hello4"
    ss.weight.attrs["test_attr5"] = "hello5"
    assert (cp.vis[:] == 1.0).all()
    assert (cp.weight[:] == np.arange(16)).all()
    assert cp.vis.attrs["test_attr4"] == "hello4"
    assert "test_attr5" not in cp.weight.attrs


def test_container_save_load_keeps_datasets_and_attributes_like_the_reference(tmp_path):
    """The round trip of test/test_io.py::test_LoadBasicCont_simple (a stage output written and read back): datasets,
    container attributes, dataset attributes and axis attributes survive."""
    from draco_amd.core import containers

    ss = _ss_container()
    ss.vis[:] = (np.arange(5)[:, None, None] + 1j * np.arange(16)[None, None, :]).astype(np.complex64)
    ss.weight[:] = np.arange(5)[None, :, None]
    fname = str(tmp_path / "ss.npz")
    ss.save(fname)
    back = containers.SiderealStream.load(fname)
    assert (back.vis[:] == ss.vis[:]).all() and (back.weight[:] == ss.weight[:]).all()
    assert back.attrs["test_attr1"] == "hello"
    assert back.vis.attrs["test_attr2"] == "hello2" and back.weight.attrs["test_attr3"] == "hello3"
    assert back.index_attrs["freq"]["alignment"] == 1
    assert tuple(back.vis.attrs["axis"]) == ("freq", "stack", "ra")
    assert np.array_equal(back.freq, ss.freq)


def test_packed_store_pack_once_from_a_per_tile_provider(tmp_path):
    """`PackedStoreProvider.pack`: a provider that only has per-tile `beam_m` (what a driftscan BeamTransfer offers,
    mapmaker.py:160-162) is written ONCE, by worker processes, into a .npy file in the pool's wire format; the store
    over the memory-mapped file returns the same tiles, `beam_block` hands out views of it, and a later process reopens
    it without packing again."""
    from draco_amd import _lib
    from draco_amd.core.products import ArrayProvider, PackedStoreProvider, TransitTelescope, synth_beam_tile

    lmax = 9
    tel = TransitTelescope(np.linspace(400.0, 500.0, 3), lmax=lmax, ncyl=1, nfeed_cyl=3)
    bt = ArrayProvider(tel, lambda m, f: synth_beam_tile(11, m, f, tel.npairs, 4, lmax))
    for procs in (1, 3):
        path = str(tmp_path / f"b{procs}.npy")
        st = PackedStoreProvider.pack(bt, path, processes=procs, chunk_bytes=2048)  # many small chunks: every boundary
        assert isinstance(st.store, np.memmap) and st.store.dtype == np.complex128
        for f in range(tel.nfreq):
            for m in range(lmax + 1):
                assert np.array_equal(st.beam_m(m, fi=f), bt.beam_m(m, fi=f))
        blk = st.beam_block(0, lmax + 1, 1, 2, np.complex128, _lib.DMM_B_PACKED)
        assert np.shares_memory(blk, st.store)  # a view: nothing is packed again on the way to the GPU
        ref = bt.beam_block(0, lmax + 1, 1, 2, np.complex128, _lib.DMM_B_PACKED)
        assert np.array_equal(np.asarray(blk), ref)
    again = PackedStoreProvider.open(tel, str(tmp_path / "b3.npy"))
    assert np.array_equal(again.beam_m(5, fi=2), bt.beam_m(5, fi=2))
    st64 = PackedStoreProvider.pack(bt, str(tmp_path / "c.npy"), dtype=np.complex64, processes=2)
    assert st64.store.dtype == np.complex64
    assert np.allclose(st64.beam_m(3, fi=0), bt.beam_m(3, fi=0), rtol=1e-6, atol=1e-7)


def test_bench_gpus_flag_starts_ranks_or_refuses():
    """`python bench.py --gpus N` (the driver's command shape, no torchrun around it) must never run ONE rank and report
    n_gpus = 1 (VERDICT r3 missing 1): with fewer than N devices visible -- none in the CPU container -- it refuses."""
    import os
    import pathlib
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box: the command would be a real run")
    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "GPU(s) visible" in res.stderr and not res.stdout.strip()


class _PicklableTiles:
    """A per-tile provider body that survives pickling (module level, no closure)."""

    def __init__(self, npairs, lmax):
        self.npairs, self.lmax = npairs, lmax

    def __call__(self, m, f):
        from draco_amd.core.products import synth_beam_tile

        return synth_beam_tile(11, m, f, self.npairs, 4, self.lmax)


def test_pack_never_forks_a_process_that_holds_the_gpu(tmp_path, monkeypatch):
    """`PackedStoreProvider.pack` forked its workers whatever the parent's state.  A provider that
    fills on the device is refused (workers must not touch the GPU); once this process has initialised the GPU the
    workers are SPAWNED from a pickled provider (or a factory), and an unpicklable provider is a clear error."""
    import torch

    from draco_amd.core.products import ArrayProvider, PackedStoreProvider, TransitTelescope

    lmax = 7
    tel = TransitTelescope(np.linspace(400.0, 500.0, 2), lmax=lmax, ncyl=1, nfeed_cyl=3)
    host = ArrayProvider(tel, _PicklableTiles(tel.npairs, lmax))
    dev = ArrayProvider(tel, _PicklableTiles(tel.npairs, lmax))
    dev.fill_mode = "device"
    with pytest.raises(ValueError, match="generates its tiles on the GPU"):
        PackedStoreProvider.pack(dev, str(tmp_path / "d.npy"), processes=2, chunk_bytes=2048)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    st = PackedStoreProvider.pack(host, str(tmp_path / "s.npy"), processes=2, chunk_bytes=2048)  # spawned workers, pickled provider
    for m in (0, 3, lmax):
        assert np.array_equal(st.beam_m(m, fi=1), host.beam_m(m, fi=1))
    closure = ArrayProvider(tel, lambda m, f: host.beam_m(m, fi=f))
    with pytest.raises(RuntimeError, match="need a picklable provider"):
        PackedStoreProvider.pack(closure, str(tmp_path / "c.npy"), processes=2, chunk_bytes=2048)

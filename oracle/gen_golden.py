"""Generate ``tests/golden/*.npz`` by EXECUTING the reference's own functions.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs
``/root/reference``).  Usage::

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

The reference modules are imported from where they lie under placeholder modules
for the absent third-party packages (``oracle/_refstub.py``).  What is executed,
unmodified, is the reference source of

* ``transform._make_marray`` / ``_unpack_marray`` / ``_make_ssarray``
* ``transform.MModeTransform.process`` / ``MModeInverseTransform.process``
  (on duck-typed containers: the arithmetic at ``transform.py:594-639`` and
  ``:756-790`` is the reference's own)
* ``mapmaker.pinv_svd`` and the three ``_solve_m`` methods (against a duck-typed
  beam-transfer object supplying seeded synthetic ``beam_m`` tiles).

Fixtures hold inputs and the reference's outputs only -- no reference source.
"""

from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def crandn(rng, shape, dtype=np.complex128):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


# --------------------------------------------------------------------------- containers
class _Arr(np.ndarray):
    @property
    def local_array(self):
        return self.view(np.ndarray)


class _DS:
    def __init__(self, arr):
        self.arr = arr.view(_Arr)

    def __getitem__(self, k):
        return self.arr[k]

    def __setitem__(self, k, v):
        self.arr[k] = v

    @property
    def shape(self):
        return self.arr.shape


class _FakeCont:
    comm = None

    def redistribute(self, axis):
        pass


class FakeSiderealStream(_FakeCont):
    def __init__(self, vis=None, weight=None, ra=None, axes_from=None, attrs_from=None, **kw):
        if vis is None:
            mm = axes_from
            nfreq, nstack = mm.vis.shape[2:]
            vis = np.zeros((nfreq, nstack, ra), np.complex64)
            weight = np.zeros((nfreq, nstack, ra), np.float32)
        self.vis = _DS(vis)
        self.weight = _DS(weight)
        self.attrs = {}


class FakeMModes(_FakeCont):
    def __init__(self, mmax=None, oddra=None, axes_from=None, attrs_from=None, comm=None, vis=None, weight=None):
        if vis is None:
            shp = axes_from.vis.shape[:-1]
            vis = np.zeros((mmax + 1, 2, *shp), np.complex128)
            weight = np.zeros((mmax + 1, 2, *shp), np.float64)
        self.vis = _DS(vis)
        self.weight = _DS(weight)
        self.attrs = {"oddra": bool(oddra)}
        self.mmax = vis.shape[0] - 1
        self.oddra = bool(oddra)


class FakeHybridVisStream(_FakeCont):
    def __init__(self, vis, weight):
        self.vis = _DS(vis)
        self.weight = _DS(weight)
        self.attrs = {}


class FakeHybridVisMModes(_FakeCont):
    """complex64 vis [m, msign, pol, freq, ew, el], float32 weight [m, msign, pol, freq, ew] (containers.py:1559-1574)."""

    def __init__(self, mmax=None, oddra=None, axes_from=None, attrs_from=None, comm=None):
        self.vis = _DS(np.zeros((mmax + 1, 2, *axes_from.vis.shape[:-1]), np.complex64))
        self.weight = _DS(np.zeros((mmax + 1, 2, *axes_from.weight.shape[:-1]), np.float32))
        self.attrs = {"oddra": bool(oddra)}


class _FakeContainers:
    SiderealStream = FakeSiderealStream
    MModes = FakeMModes
    HybridVisStream = FakeHybridVisStream
    HybridVisMModes = FakeHybridVisMModes


class _Tel:
    def __init__(self, lmax, mmax, nfreq, npol=4):
        self.lmax = lmax
        self.mmax = mmax
        self.nfreq = nfreq
        self.num_pol_sky = npol
        self.frequencies = np.linspace(400.0, 800.0, nfreq, endpoint=False)


class _BT:
    """Duck-typed beam-transfer object serving seeded tiles with the l<m zeros."""

    def __init__(self, npairs, lmax, nfreq, seed, npol=4):
        self.telescope = _Tel(lmax, lmax, nfreq, npol)
        self.npairs = npairs
        self.ntel = 2 * npairs
        self.nsky = npol * (lmax + 1)
        self.seed = seed
        self.npol = npol

    def beam_m(self, m, fi=None):
        rng = np.random.default_rng([self.seed, m, fi])
        lmax = self.telescope.lmax
        b = crandn(rng, (2, self.npairs, self.npol, lmax + 1)) / np.sqrt(self.ntel)
        b[..., :m] = 0.0
        return b


def gen_transform(transform, out):
    rng = np.random.default_rng(1001)
    cases = {}
    idx = 0
    for N in (16, 15):
        for mmax in (N // 2, N // 2 - 3, N):
            for lead in ((3, 5), (2, 2, 3, 2)):
                ts = crandn(rng, (*lead, N), np.complex64)
                mm = transform._make_marray(ts, mmax=mmax, use_fftw=False)
                # also the write-into-complex128 form used by MModeTransform (:623-624)
                mm128 = np.zeros((mmax + 1, 2, *lead), np.complex128)
                transform._make_marray(ts, mm128, use_fftw=False)
                cases[f"c{idx}_ts"] = ts
                cases[f"c{idx}_mmax"] = np.int64(mmax)
                cases[f"c{idx}_out64"] = mm
                cases[f"c{idx}_out128"] = mm128
                idx += 1
    # complex128 input (exact path, used for tight-tolerance checks of the packing)
    for N, mmax in ((16, 8), (15, 7), (12, 4), (9, 11)):
        ts = crandn(rng, (2, 3, N), np.complex128)
        mm = transform._make_marray(ts, mmax=mmax, dtype=np.complex128, use_fftw=False)
        cases[f"c{idx}_ts"] = ts
        cases[f"c{idx}_mmax"] = np.int64(mmax)
        cases[f"c{idx}_out64"] = mm
        cases[f"c{idx}_out128"] = mm
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "transform_make_marray.npz"), **cases)

    # inverse: _unpack_marray / _make_ssarray incl. n= resampling and the Nyquist rule
    cases = {}
    idx = 0
    for N, mmax, n in ((16, 8, None), (15, 7, None), (16, 8, 12), (16, 8, 24), (15, 7, 9), (15, 7, 32), (16, 5, None), (15, 4, 15)):
        ts = crandn(rng, (2, 3, N), np.complex128)
        mm = transform._make_marray(ts, mmax=mmax, dtype=np.complex128, use_fftw=False)
        up = transform._unpack_marray(mm, n=n)
        ss = transform._make_ssarray(mm, n=n)
        cases[f"c{idx}_mm"] = mm
        cases[f"c{idx}_n"] = np.int64(-1 if n is None else n)
        cases[f"c{idx}_unpack"] = up
        cases[f"c{idx}_ss"] = ss
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "transform_unpack.npz"), **cases)

    # task level: MModeTransform.process / MModeInverseTransform.process on fake containers
    transform.containers = _FakeContainers
    transform.mpiarray.MPIArray.wrap = staticmethod(lambda a, axis=0, comm=None: a)
    cases = {}
    idx = 0
    for nra, tel_mmax, window in ((16, None, False), (15, None, False), (16, 5, False), (16, None, True), (15, 6, True), (32, 40, False)):
        vis = crandn(rng, (3, 4, nra), np.complex64)
        w = rng.uniform(0.5, 1.5, (3, 4, nra)).astype(np.float32)
        w[rng.uniform(size=w.shape) < 0.1] = 0.0
        w[1, 2, :] = 0.0  # a fully flagged baseline
        task = transform.MModeTransform.__new__(transform.MModeTransform)
        task.remove_integration_window = window
        task.use_fftw = False
        task.telescope = None if tel_mmax is None else _Tel(tel_mmax, tel_mmax, 3)
        ma = task.process(FakeSiderealStream(vis.copy(), w.copy()))
        cases[f"c{idx}_vis"] = vis
        cases[f"c{idx}_weight"] = w
        cases[f"c{idx}_mmax"] = np.int64(-1 if tel_mmax is None else tel_mmax)
        cases[f"c{idx}_window"] = np.bool_(window)
        cases[f"c{idx}_mvis"] = ma.vis.arr.view(np.ndarray)
        cases[f"c{idx}_mweight"] = ma.weight.arr.view(np.ndarray)
        cases[f"c{idx}_oddra"] = np.bool_(ma.attrs["oddra"])
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "transform_mmode_task.npz"), **cases)

    cases = {}
    idx = 0
    for nra_in, mmax, nra_out, window in ((16, 8, None, False), (15, 7, None, False), (16, 8, 24, False), (16, 8, 10, True), (15, 7, 15, True)):
        ts = crandn(rng, (3, 4, nra_in), np.complex128)
        mv = transform._make_marray(ts, mmax=mmax, dtype=np.complex128, use_fftw=False)
        mw = np.broadcast_to(rng.uniform(0.5, 1.5, (1, 1, 3, 4)), mv.shape).copy()
        task = transform.MModeInverseTransform.__new__(transform.MModeInverseTransform)
        task.nra = nra_out
        task.apply_integration_window = window
        mcont = FakeMModes(oddra=bool(nra_in % 2), vis=mv.copy(), weight=mw.copy())
        ss = task.process(mcont)
        cases[f"c{idx}_mvis"] = mv
        cases[f"c{idx}_mweight"] = mw
        cases[f"c{idx}_oddra"] = np.bool_(nra_in % 2)
        cases[f"c{idx}_nra"] = np.int64(-1 if nra_out is None else nra_out)
        cases[f"c{idx}_window"] = np.bool_(window)
        cases[f"c{idx}_vis"] = ss.vis.arr.view(np.ndarray)
        cases[f"c{idx}_weight"] = ss.weight.arr.view(np.ndarray)
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "transform_inverse_task.npz"), **cases)


def gen_mapmaker(mapmaker, out):
    rng = np.random.default_rng(2002)

    # pinv_svd: full rank, rank-deficient by the rcond rule, by the acond rule
    cases = {}
    A = crandn(rng, (12, 7))
    u, s, vh = np.linalg.svd(crandn(rng, (10, 14)), full_matrices=False)
    s_r = np.array([5.0, 3.0, 1.0, 0.5, 0.2, 0.05, 6e-3, 4e-3, 1e-3, 1e-5])  # rcond cut at 5e-3
    B = (u * s_r) @ vh
    s_a = np.array([0.05, 0.03, 0.01, 5e-3, 2e-3, 1e-3, 3e-4, 1.5e-4, 5e-5, 1e-6])  # acond cut 1e-4
    C = (u * s_a) @ vh
    for i, M in enumerate((A, B, C, A.real.copy())):
        cases[f"c{i}_M"] = M
        cases[f"c{i}_pinv"] = mapmaker.pinv_svd(M)
    cases["ncase"] = np.int64(4)
    np.savez_compressed(os.path.join(out, "mapmaker_pinv_svd.npz"), **cases)

    # _solve_m for the three map-makers
    cases = {}
    idx = 0
    # (npairs, lmax): ntel > nsky (Wiener branch 1) and ntel <= nsky (branch 2)
    for npairs, lmax, seed in ((20, 5, 31), (7, 9, 32), (9, 8, 33)):
        bt = _BT(npairs, lmax, 3, seed)
        tasks = {}
        for name, cls in (("dirty", mapmaker.DirtyMapMaker), ("ml", mapmaker.MaximumLikelihoodMapMaker), ("wiener", mapmaker.WienerMapMaker)):
            t = cls.__new__(cls)
            t.beamtransfer = bt
            t.bt_cache = None
            t.prior_amp = 1.0
            t.prior_tilt = 0.5
            tasks[name] = t
        for m in (0, 2, lmax):
            for f in (0, 2):
                v = crandn(rng, (2, npairs))
                Ni = rng.uniform(0.5, 1.5, (2, npairs))
                Ni[rng.uniform(size=Ni.shape) < 0.15] = 0.0
                if m == 2:
                    Ni *= 25.0  # move the singular values relative to acond
                cases[f"c{idx}_bm"] = bt.beam_m(m, fi=f)
                cases[f"c{idx}_m"] = np.int64(m)
                cases[f"c{idx}_v"] = v
                cases[f"c{idx}_Ni"] = Ni
                for name, t in tasks.items():
                    cases[f"c{idx}_{name}"] = t._solve_m(m, f, v.copy(), Ni.copy())
                # non-default prior
                tw = tasks["wiener"]
                tw.prior_amp, tw.prior_tilt = 2.5, 1.25
                cases[f"c{idx}_wiener_p"] = tw._solve_m(m, f, v.copy(), Ni.copy())
                tw.prior_amp, tw.prior_tilt = 1.0, 0.5
                idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "mapmaker_solve_m.npz"), **cases)


def gen_hybrid(transform, out):
    """MModeTransform.process on a HybridVisStream-shaped input (transform.py:585-590 contmap)."""
    transform.containers = _FakeContainers
    rng = np.random.default_rng(4004)
    cases = {}
    idx = 0
    for nra, tel_mmax, window in ((16, None, False), (15, 5, True)):
        vis = crandn(rng, (2, 3, 2, 5, nra), np.complex64)  # [pol, freq, ew, el, ra]
        w = rng.uniform(0.5, 1.5, (2, 3, 2, nra)).astype(np.float32)  # [pol, freq, ew, ra]
        w[rng.uniform(size=w.shape) < 0.1] = 0.0
        task = transform.MModeTransform.__new__(transform.MModeTransform)
        task.remove_integration_window = window
        task.use_fftw = False
        task.telescope = None if tel_mmax is None else _Tel(tel_mmax, tel_mmax, 3)
        ma = task.process(FakeHybridVisStream(vis.copy(), w.copy()))
        cases[f"c{idx}_vis"] = vis
        cases[f"c{idx}_weight"] = w
        cases[f"c{idx}_mmax"] = np.int64(-1 if tel_mmax is None else tel_mmax)
        cases[f"c{idx}_window"] = np.bool_(window)
        cases[f"c{idx}_mvis"] = ma.vis.arr.view(np.ndarray)
        cases[f"c{idx}_mweight"] = ma.weight.arr.view(np.ndarray)
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "transform_hybrid_task.npz"), **cases)


class _EnumArr(np.ndarray):
    """ndarray with the two MPIArray methods the noise tasks use (single process)."""

    def enumerate(self, axis):
        return [(i, i) for i in range(self.shape[axis])]

    @property
    def local_shape(self):
        return self.shape


class _NoiseDS:
    def __init__(self, arr):
        self.arr = arr.view(_EnumArr)

    def __getitem__(self, k):
        return self.arr[k]

    def __setitem__(self, k, v):
        self.arr[k] = v

    @property
    def local_shape(self):
        return self.arr.shape


class FakeNoiseStream(FakeSiderealStream):
    def __init__(self, vis, weight, freq_width, prod, ninput):
        self.vis = _NoiseDS(vis)
        self.weight = _NoiseDS(weight)
        nra = vis.shape[-1]
        self.ra = np.linspace(0.0, 360.0, nra, endpoint=False)
        fm = np.zeros(vis.shape[0], dtype=[("centre", float), ("width", float)])
        fm["centre"] = 400.0 + np.arange(vis.shape[0])
        fm["width"] = freq_width
        self.index_map = {"freq": fm, "input": np.arange(ninput), "prod": prod}
        self.prodstack = prod
        self.attrs = {}


def _unpack_product_array(utv, mat, feeds, nfeed):
    """NumPy stand-in for the compiled helper draco/util/_fast_tools.pyx:91-128 (index shuffle only)."""
    for i, fi in enumerate(feeds):
        for j, fj in enumerate(feeds):
            a, b = (fi, fj) if fi <= fj else (fj, fi)
            pi = (nfeed * (nfeed + 1) // 2) - ((nfeed - a) * (nfeed - a + 1) // 2) + (b - a)
            mat[i, j] = utv[pi] if fi <= fj else np.conj(utv[pi])


def gen_noise(out):
    import importlib

    from draco.util import random as rrandom

    astro = importlib.import_module("caput.astro.constants")
    astro.STELLAR_S = 1.0 / 1.002737909350795
    from draco.synthesis import noise as rnoise

    rnoise.STELLAR_S = astro.STELLAR_S
    rnoise.containers = type("NS", (), {"SiderealStream": FakeSiderealStream})
    ft = importlib.import_module("draco.util._fast_tools")
    ft._unpack_product_array_fast = _unpack_product_array

    cases = {}
    rng = np.random.default_rng(77)
    cases["cn"] = rrandom.complex_normal(size=(3, 4), scale=np.array([1.0, 2.0, 3.0, 4.0]), rng=np.random.default_rng(5))
    cases["cn64"] = rrandom.complex_normal(size=(2, 5), dtype=np.complex64, loc=1 + 2j, rng=np.random.default_rng(6))
    cases["scw"] = rrandom.standard_complex_wishart(4, 50, rng=np.random.default_rng(7))
    X = crandn(rng, (5, 30))
    C = X @ X.T.conj() / 30
    cases["cw_C"] = C
    cases["cw"] = rrandom.complex_wishart(C, 100, rng=np.random.default_rng(8))

    # GaussianNoise on a full-triangle stream of 3 inputs (6 products)
    ninput, nfreq, nra = 3, 2, 8
    prod = np.array([(i, j) for i in range(ninput) for j in range(i, ninput)], dtype=[("input_a", int), ("input_b", int)])
    vis = crandn(rng, (nfreq, len(prod), nra), np.complex64)
    w = np.ones(vis.shape, np.float32)
    t = rnoise.GaussianNoise.__new__(rnoise.GaussianNoise)
    t.recv_temp, t.ndays, t.set_weights, t.add_noise, t.telescope = 50.0, 2.0, True, True, None
    t.rng = np.random.default_rng(9)
    d = FakeNoiseStream(vis.copy(), w.copy(), 0.390625, prod, ninput)
    t.process(d)
    cases["gn_vis_in"] = vis
    cases["gn_vis"] = d.vis.arr.view(np.ndarray)
    cases["gn_weight"] = d.weight.arr.view(np.ndarray)

    # SampleNoise: expectation = positive-definite covariance per (freq, time)
    exp = np.zeros((nfreq, len(prod), nra), np.complex128)
    for f in range(nfreq):
        for ti in range(nra):
            Y = crandn(rng, (ninput, 12))
            Cm = Y @ Y.T.conj() / 12 + np.eye(ninput)
            exp[f, :, ti] = Cm[np.triu_indices(ninput)]
    w2 = np.ones(exp.shape, np.float64)
    s_ = rnoise.SampleNoise.__new__(rnoise.SampleNoise)
    s_.sample_frac, s_.set_weights = 1e-4, True
    s_.rng = np.random.default_rng(10)
    d2 = FakeNoiseStream(exp.copy(), w2.copy(), 0.390625, prod, ninput)
    s_.process(d2)
    cases["sn_vis_in"] = exp
    cases["sn_vis"] = d2.vis.arr.view(np.ndarray)
    cases["sn_weight"] = d2.weight.arr.view(np.ndarray)

    # ReceiverTemperature (noise.py:21-45): a constant offset on the auto-correlations only (no random draw)
    r_ = rnoise.ReceiverTemperature.__new__(rnoise.ReceiverTemperature)
    r_.recv_temp = 42.5
    d3 = FakeNoiseStream(vis.copy(), w.copy(), 0.390625, prod, ninput)
    r_.process(d3)
    cases["rt_vis"] = d3.vis.arr.view(np.ndarray)
    np.savez_compressed(os.path.join(out, "noise.npz"), **cases)


def gen_mask(out):
    """MaskMModeData.process (flagging.py:113-173) on a duck-typed MModes."""
    from draco.analysis import flagging

    rng = np.random.default_rng(5005)
    ninput = 3
    prod = np.array([(i, j) for i in range(ninput) for j in range(i, ninput)], dtype=[("input_a", int), ("input_b", int)])
    cases = {}
    idx = 0
    for auto, mzero, pos, neg, low in ((False, False, True, True, None), (True, True, True, True, None), (False, True, False, True, None), (True, False, True, False, 3), (False, False, True, True, 2)):
        w = rng.uniform(0.5, 1.5, (6, 2, 2, len(prod)))
        mm = FakeMModes(oddra=False, vis=np.zeros(w.shape, complex), weight=w.copy())
        mm.prodstack = prod
        t = flagging.MaskMModeData.__new__(flagging.MaskMModeData)
        t.auto_correlations, t.m_zero, t.positive_m, t.negative_m, t.mask_low_m = auto, mzero, pos, neg, low
        t.process(mm)
        cases[f"c{idx}_w"] = w
        cases[f"c{idx}_opts"] = np.array([auto, mzero, pos, neg, -1 if low is None else low], dtype=np.int64)
        cases[f"c{idx}_out"] = mm.weight.arr.view(np.ndarray)
        idx += 1
    cases["ncase"] = np.int64(idx)
    cases["prod_a"] = prod["input_a"]
    cases["prod_b"] = prod["input_b"]
    np.savez_compressed(os.path.join(out, "flagging_mask_mmode.npz"), **cases)


class _ShapeDS(_DS):
    """Dataset stub with the MPIArray attributes DeconvolveHybridMBase.process reads."""

    @property
    def local_shape(self):
        return self.arr.shape

    @property
    def local_offset(self):
        return (0,) * self.arr.ndim


class FakeHybridM(_FakeCont):
    distributed = False

    def __init__(self, vis, weight, freq, ew, el, oddra):
        self.vis = _ShapeDS(vis)
        self.weight = _ShapeDS(weight)
        self.freq = np.asarray(freq, dtype=float)
        nm, _, npol = vis.shape[:3]
        self.index_map = {"m": np.arange(nm), "el": np.asarray(el), "ew": np.asarray(ew), "pol": np.arange(npol), "freq": self.freq}
        self.mmax = nm - 1
        self.oddra = bool(oddra)
        self.attrs = {}


class FakeRingMap(_FakeCont):
    def __init__(self, beam=None, ra=None, axes_from=None, attrs_from=None, distributed=None, comm=None):
        hv = axes_from
        npol, nfreq = hv.vis.shape[2], hv.vis.shape[3]
        nel = hv.vis.shape[5]
        self.index_map = dict(hv.index_map)
        self.attrs = {}
        self._shape = (beam, npol, nfreq, ra, nel)
        self.map = _DS(np.zeros(self._shape))
        self.weight = _DS(np.zeros(self._shape[1:]))

    def add_dataset(self, name):
        b, npol, nfreq, ra, nel = self._shape
        shp = {"dirty_beam_power": (b, npol, nfreq, nel), "dirty_beam": self._shape}[name]
        setattr(self, name, _DS(np.zeros(shp)))


class _Log:
    def info(self, *a, **k):
        pass

    debug = warning = info


def gen_ringmap(out):
    """TikhonovRingMapMaker / WienerRingMapMaker.process (ringmapmaker.py:538-1186) on duck-typed containers."""
    import importlib

    from draco.analysis import ringmapmaker as rmk

    rmk.containers = type("NS", (), {"RingMap": FakeRingMap, "HybridVisMModes": FakeHybridM})
    rng = np.random.default_rng(6006)
    cases = {}
    idx = 0
    nm, npol, nfreq, new, nel = 9, 2, 3, 4, 5
    freq = np.array([600.0, 650.0, 700.0])
    ew = np.array([0.0, 22.0, 44.0, 66.0])
    el = np.linspace(-0.8, 0.8, nel)

    class _T:
        latitude = 49.0

    configs = [
        ("tikhonov", dict(weight_ew="natural", inv_SN=1e-3), dict(exclude_cyl=[], skip=False, window="none", oddra=False)),
        ("tikhonov", dict(weight_ew="uniform", inv_SN=1e-2), dict(exclude_cyl=[0], skip=False, window="none", oddra=True)),
        ("tikhonov", dict(weight_ew="inverse_variance", inv_SN=1e-3), dict(exclude_cyl=[1], skip=False, window="hann", oddra=False)),
        ("tikhonov", dict(weight_ew="natural", inv_SN=1e-3), dict(exclude_cyl=[], skip=True, window="none", oddra=False)),
        ("wiener", dict(), dict(exclude_cyl=[], skip=False, window="none", oddra=False)),
        ("wiener", dict(gal_amp=2.0, psrc_amp=0.1), dict(exclude_cyl=[0], skip=False, window="blackman_harris", oddra=True)),  # window is zero at every m <= 8: all-zero outputs
        ("wiener", dict(gal_amp=2.0, psrc_amp=0.1), dict(exclude_cyl=[2], skip=False, window="blackman_harris", oddra=True)),
    ]
    for kind, attrs, base in configs:
        hv = crandn(rng, (nm, 2, npol, nfreq, new, nel), np.complex64)
        bv = crandn(rng, (nm + 2, 2, npol, nfreq, new, nel), np.complex64)
        hw = rng.uniform(0.5, 1.5, (nm, 2, npol, nfreq, new)).astype(np.float32)
        hw[rng.uniform(size=hw.shape) < 0.1] = 0.0
        hw[0, 1] = 0.0
        cls = rmk.TikhonovRingMapMaker if kind == "tikhonov" else rmk.WienerRingMapMaker
        t = cls.__new__(cls)
        t.log = _Log()
        t.exclude_cyl = list(base["exclude_cyl"])
        t.exclude_intracyl = False
        t.skip_deconvolution = base["skip"]
        t.reference_declination = None
        t.save_dirty_beam = True
        t.window_type = base["window"]
        t.window_size = 1.0
        t.window_scaled = False
        t.telescope = _T()
        if kind == "tikhonov":
            t.weight_ew = attrs["weight_ew"]
            t.inv_SN = attrs["inv_SN"]
        else:
            t.gal_amp, t.gal_alpha, t.gal_beta = attrs.get("gal_amp", 1.41), -1.75, -0.75
            t.psrc_amp, t.psrc_alpha = attrs.get("psrc_amp", 0.045), -1.0
        rm = t.process(FakeHybridM(hv.copy(), hw.copy(), freq, ew, el, base["oddra"]), FakeHybridM(bv.copy(), np.ones(bv.shape[:-1], np.float32), freq, ew, el, base["oddra"]))
        cases[f"c{idx}_kind"] = np.array(kind)
        cases[f"c{idx}_hv"], cases[f"c{idx}_bv"], cases[f"c{idx}_hw"] = hv, bv, hw
        cases[f"c{idx}_opts"] = np.array([base["skip"], base["oddra"]], dtype=np.int64)
        cases[f"c{idx}_exclude"] = np.array(base["exclude_cyl"], dtype=np.int64)
        cases[f"c{idx}_window"] = np.array(base["window"])
        cases[f"c{idx}_weight_ew"] = np.array(attrs.get("weight_ew", "inverse_variance"))
        cases[f"c{idx}_params"] = np.array([attrs.get("inv_SN", 0.0), attrs.get("gal_amp", 1.41), attrs.get("psrc_amp", 0.045)])
        cases[f"c{idx}_map"] = rm.map.arr.view(np.ndarray)
        cases[f"c{idx}_wgt"] = rm.weight.arr.view(np.ndarray)
        cases[f"c{idx}_dbp"] = rm.dirty_beam_power.arr.view(np.ndarray)
        cases[f"c{idx}_db"] = rm.dirty_beam.arr.view(np.ndarray)
        idx += 1
    cases["ncase"] = np.int64(idx)
    cases["freq"], cases["ew"], cases["el"] = freq, ew, el
    cases["latitude"] = np.float64(49.0)
    np.savez_compressed(os.path.join(out, "ringmap_deconvolve.npz"), **cases)


class _HybEnumArr(_Arr):
    def enumerate(self, axis):
        return [(i, i) for i in range(self.shape[axis])]


class _HybEnumDS(_ShapeDS):
    def __init__(self, arr):
        self.arr = arr.view(_HybEnumArr)


def gen_ringmap_analytic(out):
    """DeconvolveAnalyticalBeam._get_beam_mmodes (ringmapmaker.py:1001-1072) and the two ...Analytical makers
    (:1189-1194) end to end.  caput's pyfftw wrapper [3P] is absent: `transform.fft.fftw.fft` is served by
    np.fft.fft (the same complex128 DFT)."""
    import types

    from draco.analysis import ringmapmaker as rmk
    from draco.analysis import transform

    transform.fft = types.SimpleNamespace(fftw=types.SimpleNamespace(fft=lambda x, axes=-1: np.fft.fft(x, axis=axes)))
    rmk.containers = type("NS", (), {"RingMap": FakeRingMap, "HybridVisMModes": FakeHybridM})

    def _empty_like(c):
        o = FakeHybridM(np.zeros(c.vis.shape, np.complex64), np.zeros(c.weight.shape, np.float32), c.freq, c.index_map["ew"], c.index_map["el"], c.oddra)
        o.index_map = dict(c.index_map)
        return o

    rmk.empty_like = _empty_like

    class _T:
        latitude = 49.3

    rng = np.random.default_rng(6106)
    pol = np.array(["XX", "XY", "YX", "YY"])
    freq = np.array([410.0, 600.0, 790.0])
    ew = np.array([0.0, 22.0, 44.0, 66.0])
    cases = {"pol": pol, "freq": freq, "ew": ew, "latitude": np.float64(_T.latitude)}
    idx = 0
    for nm, oddra, nel, kind in ((9, False, 5, "tikhonov"), (8, True, 6, "wiener"), (17, False, 3, "tikhonov")):
        el = np.linspace(-0.7, 0.9, nel)
        hv = crandn(rng, (nm, 2, 4, len(freq), len(ew), nel), np.complex64)
        hw = rng.uniform(0.5, 1.5, (nm, 2, 4, len(freq), len(ew))).astype(np.float32)
        hw[rng.uniform(size=hw.shape) < 0.1] = 0.0
        cls = rmk.TikhonovRingMapMakerAnalytical if kind == "tikhonov" else rmk.WienerRingMapMakerAnalytical
        t = cls.__new__(cls)
        t.log = _Log()
        t.exclude_cyl, t.exclude_intracyl, t.skip_deconvolution = [], False, False
        t.reference_declination, t.save_dirty_beam = None, True
        t.window_type, t.window_size, t.window_scaled = "none", 1.0, False
        t.telescope = _T()
        if kind == "tikhonov":
            t.weight_ew, t.inv_SN = "natural", 1e-3
        else:
            t.gal_amp, t.gal_alpha, t.gal_beta, t.psrc_amp, t.psrc_alpha = 1.41, -1.75, -0.75, 0.045, -1.0
        hvm = FakeHybridM(hv.copy(), hw.copy(), freq, ew, el, oddra)
        hvm.vis = _HybEnumDS(hv.copy())
        hvm.index_map["pol"] = pol
        bm = t._get_beam_mmodes(hvm)
        rm = t.process(hvm)
        cases[f"c{idx}_kind"] = np.array(kind)
        cases[f"c{idx}_hv"], cases[f"c{idx}_hw"], cases[f"c{idx}_el"] = hv, hw, el
        cases[f"c{idx}_oddra"] = np.int64(oddra)
        cases[f"c{idx}_beam_m"] = bm.vis.arr.view(np.ndarray)
        cases[f"c{idx}_map"] = rm.map.arr.view(np.ndarray)
        cases[f"c{idx}_wgt"] = rm.weight.arr.view(np.ndarray)
        cases[f"c{idx}_dbp"] = rm.dirty_beam_power.arr.view(np.ndarray)
        cases[f"c{idx}_db"] = rm.dirty_beam.arr.view(np.ndarray)
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "ringmap_analytic.npz"), **cases)


class _LA(np.ndarray):
    @property
    def local_array(self):
        return self.view(np.ndarray)

    @property
    def local_bounds(self):
        return slice(0, self.shape[-1])


class _LDS:
    def __init__(self, arr):
        self.arr = arr.view(_LA)

    def __getitem__(self, k):
        r = self.arr[k]
        return r.view(_LA) if isinstance(r, np.ndarray) else r

    def __setitem__(self, k, v):
        self.arr[k] = v

    @property
    def shape(self):
        return self.arr.shape


class FakeCollateStream(_FakeCont):
    """SiderealStream duck type for CollateProducts (vis/weight/input_flags, prod/stack maps)."""

    def __init__(self, freq=None, input=None, prod=None, stack=None, reverse_map_stack=None, copy_from=None, distributed=None, comm=None, vis=None, weight=None, input_flags=None):
        nra = copy_from.vis.shape[-1] if copy_from is not None else vis.shape[-1]
        self.index_map = {"freq": freq, "input": input, "prod": prod}
        self.input, self.prod = input, prod
        self.freq = freq["centre"]
        self.is_stacked = stack is not None and copy_from is None and len(stack) != len(prod)
        self.stack = stack
        self.reverse_map = {"stack": reverse_map_stack}
        nstack = len(stack) if stack is not None else len(prod)
        if vis is None:
            vis = np.zeros((len(freq), nstack, nra), np.complex64)
            weight = np.zeros((len(freq), nstack, nra), np.float32)
            input_flags = np.zeros((len(input), nra), np.float32)
        self.vis, self.weight, self.input_flags = _LDS(vis), _LDS(weight), _LDS(input_flags)
        self.attrs = {}


class _CollateTel:
    """Telescope attributes CollateProducts reads; the pairing is a regular 1-cylinder grid built here."""

    def __init__(self, nfeed, freqs):
        self.nfeed = nfeed
        self.frequencies = np.asarray(freqs, dtype=float)
        self.input_index = np.array([(100 + i,) for i in range(nfeed)], dtype=[("chan_id", "<u2")])
        # unique baselines = separations d = j - i >= 0 (all feeds alike); (i, j) with i > j is the conjugate
        self.uniquepairs = np.array([(0, d) for d in range(nfeed)])
        self.npairs = nfeed
        self.feedmap = np.zeros((nfeed, nfeed), dtype=int)
        self.feedconj = np.zeros((nfeed, nfeed), dtype=bool)
        self.feedmask = np.ones((nfeed, nfeed), dtype=bool)
        for i in range(nfeed):
            for j in range(nfeed):
                self.feedmap[i, j] = abs(j - i)
                self.feedconj[i, j] = i > j


def gen_collate(transform, out):
    """CollateProducts.process (transform.py:168-330) on duck-typed streams (unstacked full-triangle input)."""
    import importlib

    ft = importlib.import_module("draco.util._fast_tools")

    def _calc_redundancy(input_flags, pm, stack_index, nstack, redundancy):
        for ii in range(pm.shape[0]):
            ist = stack_index[ii]
            if 0 <= ist < nstack:
                redundancy[ist] += input_flags[pm[ii, 0]] * input_flags[pm[ii, 1]]

    transform.tools._calc_redundancy = _calc_redundancy
    transform.copy_datasets_filter = lambda *a, **k: None
    transform.io.get_telescope = lambda t: t
    rng = np.random.default_rng(7007)
    cases = {}
    idx = 0
    nfeed_tel, nra = 4, 6
    tel_freq = [400.0, 410.0]
    for weight, extra_feed, perm_freq in (("inverse_variance", False, False), ("natural", False, False), ("uniform", True, True), ("inverse_variance", True, True)):
        tel = _CollateTel(nfeed_tel, tel_freq)
        # the file may hold one more input than the telescope and an extra / permuted frequency
        file_ids = [100 + i for i in range(nfeed_tel)] + ([999] if extra_feed else [])
        if extra_feed:
            file_ids = [file_ids[i] for i in (2, 0, 4, 1, 3)]
        ninp = len(file_ids)
        inputs = np.array([(c,) for c in file_ids], dtype=[("chan_id", "<u2")])
        ffreq = [410.0, 405.0, 400.0] if perm_freq else list(tel_freq)
        fm = np.zeros(len(ffreq), dtype=[("centre", float), ("width", float)])
        fm["centre"], fm["width"] = ffreq, 10.0
        prod = np.array([(i, j) for i in range(ninp) for j in range(i, ninp)], dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        rev = np.zeros(len(prod), dtype=[("stack", "<u4"), ("conjugate", "u1")])
        rev["stack"] = np.arange(len(prod))
        vis = crandn(rng, (len(ffreq), len(prod), nra), np.complex64)
        w = rng.uniform(0.5, 1.5, vis.shape).astype(np.float32)
        w[rng.uniform(size=w.shape) < 0.15] = 0.0
        flags = (rng.uniform(size=(ninp, nra)) > 0.2).astype(np.float32)
        ss = FakeCollateStream(freq=fm, input=inputs, prod=prod, stack=None, reverse_map_stack=rev, vis=vis.copy(), weight=w.copy(), input_flags=flags.copy())
        t = transform.CollateProducts.__new__(transform.CollateProducts)
        t.log = _Log()
        t.weight = weight
        transform.TelescopeStreamMixIn.setup(t, tel)
        sp = t.process(ss)
        cases[f"c{idx}_weight"] = np.array(weight)
        cases[f"c{idx}_file_ids"] = np.array(file_ids)
        cases[f"c{idx}_ffreq"] = np.array(ffreq)
        cases[f"c{idx}_vis"], cases[f"c{idx}_w"], cases[f"c{idx}_flags"] = vis, w, flags
        cases[f"c{idx}_out_vis"] = sp.vis.arr.view(np.ndarray)
        cases[f"c{idx}_out_w"] = sp.weight.arr.view(np.ndarray)
        cases[f"c{idx}_out_flags"] = sp.input_flags.arr.view(np.ndarray)
        cases[f"c{idx}_out_stack_prod"] = sp.stack["prod"]
        cases[f"c{idx}_out_stack_conj"] = sp.stack["conjugate"]
        cases[f"c{idx}_out_rev_stack"] = sp.reverse_map["stack"]["stack"]
        idx += 1
    cases["ncase"] = np.int64(idx)
    # already redundancy-stacked inputs (transform.py:206-221 + tools.redefine_stack_index_map): the file's
    # products are stacked by separation; representatives are chosen to involve the input the telescope
    # lacks, and one telescope pair is masked, so that the representative products must be re-derived
    sidx = 0
    for weight, mask_pair in (("inverse_variance", False), ("natural", True), ("uniform", True)):
        tel = _CollateTel(nfeed_tel, tel_freq)
        if mask_pair:
            tel.feedmask[0, 1] = tel.feedmask[1, 0] = False
        file_ids = [100 + i for i in range(nfeed_tel)] + [999]
        ninp = len(file_ids)
        inputs = np.array([(c,) for c in file_ids], dtype=[("chan_id", "<u2")])
        fm = np.zeros(len(tel_freq), dtype=[("centre", float), ("width", float)])
        fm["centre"], fm["width"] = tel_freq, 10.0
        prod = np.array([(i, j) for i in range(ninp) for j in range(i, ninp)], dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        rev = np.zeros(len(prod), dtype=[("stack", "<u4"), ("conjugate", "u1")])
        rev["stack"] = prod["input_b"].astype(int) - prod["input_a"].astype(int)
        rev["conjugate"] = prod["input_a"] % 2
        nst = ninp
        stack = np.zeros(nst, dtype=[("prod", "<u4"), ("conjugate", "u1")])
        for s_ in range(nst):  # representative: the LAST product of the stack (involves input 999)
            members = np.flatnonzero(rev["stack"] == s_)
            stack["prod"][s_] = members[-1]
            stack["conjugate"][s_] = rev["conjugate"][members[-1]]
        vis = crandn(rng, (len(tel_freq), nst, nra), np.complex64)
        w = rng.uniform(0.5, 1.5, vis.shape).astype(np.float32)
        w[rng.uniform(size=w.shape) < 0.15] = 0.0
        flags = (rng.uniform(size=(ninp, nra)) > 0.2).astype(np.float32)
        ss = FakeCollateStream(freq=fm, input=inputs, prod=prod, stack=stack, reverse_map_stack=rev, vis=vis.copy(), weight=w.copy(), input_flags=flags.copy())
        assert ss.is_stacked
        t = transform.CollateProducts.__new__(transform.CollateProducts)
        t.log = _Log()
        t.log.warning = lambda *a, **k: None
        t.weight = weight
        transform.TelescopeStreamMixIn.setup(t, tel)
        sp = t.process(ss)
        cases[f"s{sidx}_weight"] = np.array(weight)
        cases[f"s{sidx}_mask_pair"] = np.int64(mask_pair)
        cases[f"s{sidx}_file_ids"] = np.array(file_ids)
        cases[f"s{sidx}_vis"], cases[f"s{sidx}_w"], cases[f"s{sidx}_flags"] = vis, w, flags
        cases[f"s{sidx}_stack_prod"], cases[f"s{sidx}_stack_conj"] = stack["prod"], stack["conjugate"]
        cases[f"s{sidx}_rev_stack"], cases[f"s{sidx}_rev_conj"] = rev["stack"], rev["conjugate"]
        cases[f"s{sidx}_out_vis"] = sp.vis.arr.view(np.ndarray)
        cases[f"s{sidx}_out_w"] = sp.weight.arr.view(np.ndarray)
        sidx += 1
    cases["nstacked"] = np.int64(sidx)
    cases["nfeed_tel"] = np.int64(nfeed_tel)
    cases["tel_freq"] = np.array(tel_freq)
    np.savez_compressed(os.path.join(out, "transform_collate.npz"), **cases)


class _EnumDS(_DS):
    """Dataset whose slice also answers MPIArray.enumerate / local_array (single process)."""

    def __init__(self, arr):
        self.arr = arr.view(_EnumLocal)


class _EnumLocal(_Arr):
    def enumerate(self, axis):
        return [(i, i) for i in range(self.shape[axis])]


class _OneRankComm:
    def allreduce(self, x, op=None):
        return x


class FakeSVDSpectrum(_FakeCont):
    def __init__(self, singularvalue=None, axes_from=None, **kw):
        self.spectrum = _DS(np.zeros((axes_from.vis.shape[0], singularvalue)))


def gen_svd(out):
    """svd_em, SVDSpectrumEstimator.process and SVDFilter.process (svdfilter.py) on duck-typed MModes."""
    from draco.analysis import svdfilter

    svdfilter.containers = type("C", (), {"SVDSpectrum": FakeSVDSpectrum})
    rng = np.random.default_rng(8008)
    cases = {}
    # (a) svd_em on single matrices: no mask, 10 % mask, wide and tall
    idx = 0
    for shape, frac, niter, rank in (((6, 10), 0.0, 5, 5), ((6, 10), 0.15, 5, 5), ((12, 5), 0.1, 3, 2), ((8, 8), 0.2, 4, 3)):
        # low-rank foreground + noise, like the data the task is meant for
        nr, nc = shape
        A = sum(10.0 ** (2 - k) * np.outer(crandn(rng, nr), crandn(rng, nc)) for k in range(3)) + 0.01 * crandn(rng, shape)
        mask = rng.uniform(size=shape) < frac
        u, sig, vh = svdfilter.svd_em(A, mask, niter=niter, rank=rank)
        cases[f"e{idx}_A"], cases[f"e{idx}_mask"] = A, mask
        cases[f"e{idx}_opts"] = np.array([niter, rank])
        cases[f"e{idx}_sig"] = sig
        cases[f"e{idx}_recon"] = np.dot(u * sig, vh)
        idx += 1
    cases["nem"] = np.int64(idx)
    # (b) the two tasks
    idx = 0
    for nm, nfreq, nbase, frac, niter, gthr, lthr in ((4, 6, 5, 0.0, 5, 1e-3, 1e-2), (5, 8, 3, 0.1, 5, 1e-3, 1e-2), (3, 5, 6, 0.2, 2, 0.3, 0.5), (3, 7, 2, 0.05, 3, 2.0, 0.05)):
        fg = np.zeros((nm, 2, nfreq, nbase), complex)
        for k in range(2):  # frequency-smooth, bright components + noise
            fg += 10.0 ** (3 - 2 * k) * crandn(rng, (nm, 2, 1, nbase)) * np.cos(0.3 * (k + 1) * np.arange(nfreq))[None, None, :, None]
        vis = fg + 0.1 * crandn(rng, fg.shape)
        w = rng.uniform(0.5, 1.5, vis.shape)
        w[rng.uniform(size=w.shape) < frac] = 0.0
        mm = FakeMModes(oddra=False, vis=vis.copy(), weight=w.copy())
        mm.vis, mm.weight = _EnumDS(vis.copy()), _EnumDS(w.copy())
        t = svdfilter.SVDSpectrumEstimator.__new__(svdfilter.SVDSpectrumEstimator)
        t.log = _Log()
        t.niter = niter
        spec = t.process(mm)
        mm2 = FakeMModes(oddra=False, vis=vis.copy(), weight=w.copy())
        mm2.vis, mm2.weight = _EnumDS(vis.copy()), _EnumDS(w.copy())
        mm2.comm = _OneRankComm()
        t2 = svdfilter.SVDFilter.__new__(svdfilter.SVDFilter)
        t2.log = _Log()
        t2.niter, t2.global_threshold, t2.local_threshold = niter, gthr, lthr
        res = t2.process(mm2)
        cases[f"c{idx}_vis"], cases[f"c{idx}_w"] = vis, w
        cases[f"c{idx}_opts"] = np.array([niter, gthr, lthr])
        cases[f"c{idx}_spectrum"] = spec.spectrum.arr.view(np.ndarray)
        cases[f"c{idx}_filtered"] = res.vis.arr.view(np.ndarray)
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "svdfilter.npz"), **cases)


class FakeExpandStream(_FakeCont):
    """SiderealStream duck type for ExpandProducts: also the constructor it calls for its output."""

    last = None

    def __init__(self, prod=None, stack=None, axes_from=None, vis=None, input=None):
        if axes_from is not None:
            nfreq, _, nra = axes_from.vis.shape
            vis = np.zeros((nfreq, len(prod), nra), np.complex64)
            input = axes_from.input
        self.input = input
        self.prod = prod
        self.vis = _DS(vis)
        self.weight = _DS(np.zeros(vis.shape, np.float32))
        self.maps = {}

    def create_index_map(self, name, arr):
        self.maps[name] = arr

    def create_reverse_map(self, name, arr):
        self.maps["rev_" + name] = arr


def gen_expand(out):
    """ExpandProducts.process (synthesis/stream.py:193-246) on a duck-typed stacked stream."""
    from draco.synthesis import stream

    stream.containers = type("C", (), {"SiderealStream": FakeExpandStream})
    stream.io.get_telescope = lambda t: t
    rng = np.random.default_rng(9009)
    cases = {}
    idx = 0
    for nfeed, mask in ((4, False), (5, True)):
        tel = _CollateTel(nfeed, [400.0, 410.0, 420.0])
        if mask:
            tel.feedmap[1, 3] = tel.feedmap[3, 1] = -1
        vis = crandn(rng, (3, tel.npairs, 6), np.complex64)
        ss = FakeExpandStream(vis=vis.copy(), input=np.arange(nfeed))
        t = stream.ExpandProducts.__new__(stream.ExpandProducts)
        t.setup(tel)
        res = t.process(ss)
        cases[f"c{idx}_nfeed"] = np.int64(nfeed)
        cases[f"c{idx}_mask"] = np.int64(mask)
        cases[f"c{idx}_vis"] = vis
        cases[f"c{idx}_out_vis"] = res.vis.arr.view(np.ndarray)
        cases[f"c{idx}_out_w"] = res.weight.arr.view(np.ndarray)
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "stream_expand.npz"), **cases)


# --------------------------------------------------------------------------- orchestration: fake MPIArray
class FakeMPIArray(np.ndarray):
    """Single-process stand-in for caput's ``MPIArray`` [3P]: an ndarray whose distributed axis is whole.

    Only what ``SimulateSidereal.process`` (stream.py:91-127) and ``BaseMapMaker.process``
    (mapmaker.py:62-113) touch: the constructor ``MPIArray(global_shape, axis=, dtype=, comm=)``, ``wrap``,
    ``redistribute`` (one rank: a no-op), ``enumerate`` (local index == global index), ``local_array``,
    ``local_shape`` and ``reshape`` with ``None`` standing for the distributed axis.
    """

    def __new__(cls, global_shape, axis=0, comm=None, dtype=np.float64, **kw):
        return np.zeros(global_shape, dtype=dtype).view(cls)

    @staticmethod
    def wrap(arr, axis=0, comm=None):
        return np.asarray(arr).view(FakeMPIArray)

    def redistribute(self, axis=None):
        return self

    def enumerate(self, axis):
        return [(i, i) for i in range(self.shape[axis])]

    @property
    def local_array(self):
        return self.view(np.ndarray)

    @property
    def local_shape(self):
        return self.shape

    def reshape(self, *shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], tuple | list):
            shape = tuple(shape[0])
        shape = tuple(-1 if s is None else s for s in shape)
        return np.ndarray.reshape(self, shape, **kw)


class _MPIDS:
    """Dataset whose slices are FakeMPIArrays (``mmodes.vis[: mmax + 1]`` must answer ``redistribute``)."""

    def __init__(self, arr):
        self.arr = np.asarray(arr)

    def __getitem__(self, k):
        return self.arr[k].view(FakeMPIArray)

    def __setitem__(self, k, v):
        self.arr[k] = v

    @property
    def shape(self):
        return self.arr.shape


def _freqmap(centres, width=1.0):
    fm = np.zeros(len(centres), dtype=[("centre", np.float64), ("width", np.float64)])
    fm["centre"], fm["width"] = centres, width
    return fm


class FakeMap(_FakeCont):
    """cora/draco ``Map`` duck type: ``map [freq, pol, pixel]`` (containers.py:470-486)."""

    def __init__(self, nside=None, axes_from=None, comm=None, freq=None, npol=4, map_=None):
        if map_ is None:
            nfreq = len(axes_from.index_map["freq"]) if axes_from is not None else len(freq)
            map_ = np.zeros((nfreq, npol, 12 * nside**2))
            freq = axes_from.index_map["freq"] if axes_from is not None else freq
        self.map = _MPIDS(map_)
        self.index_map = {"freq": freq}
        self.nside = nside


class FakeSimStream(_FakeCont):
    """Records what ``SimulateSidereal.process`` hands the SiderealStream constructor (stream.py:167-176)."""

    def __init__(self, freq=None, ra=None, input=None, distributed=None, comm=None, **kwargs):
        self.ctor = dict(freq=freq, ra=ra, input=input, **kwargs)
        npr = len(kwargs["stack"]) if "stack" in kwargs else len(kwargs["prod"])
        self.vis = _DS(np.zeros((len(freq), npr, ra), np.complex64))  # containers.py:501-503 dtype
        self.weight = _DS(np.zeros((len(freq), npr, ra), np.float32))


class _SimTel:
    """Telescope attributes stream.py:68-71,144-162 reads; a 1-cylinder grid of ``nfeed`` like feeds."""

    def __init__(self, nfeed, freqs, lmax, mmax, npol, stackable, with_input_index):
        self.nfeed, self.lmax, self.mmax, self.num_pol_sky = nfeed, lmax, mmax, npol
        self.frequencies = np.asarray(freqs, dtype=np.float64)
        self.nfreq = len(self.frequencies)
        if with_input_index:
            self.input_index = np.array([(7 + i, i) for i in range(nfeed)], dtype=[("chan_id", "<u2"), ("correlator_input", "<u2")])
        if stackable:  # unique baselines = separations (autos included): npairs = nfeed < nfeed(nfeed+1)/2
            self.uniquepairs = np.array([(0, d) for d in range(nfeed)])
            prod = [(i, j) for i in range(nfeed) for j in range(i, nfeed)]
            self.index_map_prod = np.array(prod, dtype=[("input_a", "<u2"), ("input_b", "<u2")])
            self.index_map_stack = np.array([(prod.index((0, d)), 0) for d in range(nfeed)], dtype=[("prod", "<u4"), ("conjugate", "u1")])
            self.reverse_map_stack = np.array([(j - i, 0) for i, j in prod], dtype=[("stack", "<u4"), ("conjugate", "u1")])
        else:  # the full triangle
            self.uniquepairs = np.array([(i, j) for i in range(nfeed) for j in range(i, nfeed)])
        self.npairs = len(self.uniquepairs)


class _SimBT(_BT):
    """``_BT`` + driftscan's ``project_vector_sky_to_telescope`` [3P] as the call site stream.py:109-112 uses it:
    ``[nfreq, npol, lmax+1] -> [nfreq, ntel]``, per frequency ``B_m[f] a`` with B reshaped ``[ntel, nsky]``."""

    def __init__(self, tel, seed):
        self.telescope = tel
        self.npairs, self.ntel = tel.npairs, 2 * tel.npairs
        self.npol = tel.num_pol_sky
        self.nsky = self.npol * (tel.lmax + 1)
        self.seed = seed

    def project_vector_sky_to_telescope(self, mi, vec):
        out = np.zeros((self.telescope.nfreq, self.ntel), dtype=np.complex128)
        for f in range(self.telescope.nfreq):
            out[f] = self.beam_m(mi, fi=f).reshape(self.ntel, self.nsky) @ vec[f].reshape(-1)
        return out


def _install_fake_mpi(mod):
    import types

    mod.mpiarray = types.SimpleNamespace(MPIArray=FakeMPIArray)


def gen_stream_simulate(out):
    """``SimulateSidereal.process`` (stream.py:48-178) executed from the reference source.

    Third-party pieces served by stand-ins: ``MPIArray`` -> :class:`FakeMPIArray` (one rank),
    ``mpitools.split_local(n)`` -> ``(n, 0, n)``, ``hputil.sphtrans_sky(map, lmax=)`` -> returns the seeded a_lm
    the fixture stores (so the fixture pins everything *after* the SHT: m trim, ``B_m a``, the +/-m unwrap with the
    conjugate-only rule, ``ifft * ntime``, the axis order, the complex64 cast and the prod/stack bookkeeping).
    """
    import types

    from draco.synthesis import stream

    _install_fake_mpi(stream)
    stream.mpitools = types.SimpleNamespace(split_local=lambda n: (n, 0, n))
    stream.containers = type("C", (), {"SiderealStream": FakeSimStream})
    stream.io.get_beamtransfer = lambda b: b
    stream.io.get_telescope = lambda b: b.telescope
    rng = np.random.default_rng(11011)
    cases = {}
    idx = 0
    # (nfeed, nfreq, lmax, mmax, npol, stackable telescope, task.stacked, telescope has input_index)
    for nfeed, nfreq, lmax, mmax, npol, stackable, stacked, has_idx in (
        (4, 3, 6, 6, 4, True, True, True),
        (4, 2, 7, 4, 4, True, False, True),
        (3, 2, 5, 5, 4, False, True, False),
        (5, 3, 6, 3, 1, True, True, True),
    ):
        freqs = 400.0 + 10.0 * np.arange(nfreq)
        tel = _SimTel(nfeed, freqs, lmax, mmax, npol, stackable, has_idx)
        bt = _SimBT(tel, 500 + idx)
        alm_in = crandn(rng, (nfreq, npol, lmax + 1, lmax + 1))
        alm_in[:, :, :, 0] = alm_in[:, :, :, 0].real  # a real sky has real m = 0 coefficients
        for l_ in range(lmax + 1):
            alm_in[:, :, l_, l_ + 1 :] = 0.0
        seen = {}

        def sphtrans_sky(m_, lmax=None, _alm=alm_in, _seen=seen):
            _seen["shape"], _seen["lmax"] = m_.shape, lmax
            return _alm.copy()

        stream.hputil = types.SimpleNamespace(sphtrans_sky=sphtrans_sky)
        nside = 4
        map_in = rng.standard_normal((nfreq, npol, 12 * nside**2))
        t = stream.SimulateSidereal.__new__(stream.SimulateSidereal)
        t.stacked = stacked
        t.setup(bt)
        ss = t.process(FakeMap(freq=_freqmap(freqs), map_=map_in))
        assert seen["shape"] == map_in.shape and seen["lmax"] == lmax
        c = f"c{idx}_"
        cases[c + "dims"] = np.array([nfeed, nfreq, lmax, mmax, npol, tel.npairs, int(stackable), int(stacked), int(has_idx)])
        cases[c + "freq"] = freqs
        cases[c + "alm"] = alm_in
        cases[c + "beam"] = np.array([[bt.beam_m(m, fi=f) for f in range(nfreq)] for m in range(mmax + 1)])
        cases[c + "uniquepairs"] = tel.uniquepairs
        if stackable:
            cases[c + "tel_prod"], cases[c + "tel_stack"], cases[c + "tel_rev"] = tel.index_map_prod, tel.index_map_stack, tel.reverse_map_stack
        if has_idx:
            cases[c + "tel_input"] = tel.input_index
        cases[c + "vis"] = ss.vis.arr.view(np.ndarray)
        cases[c + "weight"] = ss.weight.arr.view(np.ndarray)
        cases[c + "ctor_ra"] = np.int64(ss.ctor["ra"])
        cases[c + "ctor_input"] = np.asarray(ss.ctor["input"])
        cases[c + "ctor_prod_a"] = np.asarray(ss.ctor["prod"]["input_a"]).astype(np.int64)
        cases[c + "ctor_prod_b"] = np.asarray(ss.ctor["prod"]["input_b"]).astype(np.int64)
        cases[c + "ctor_has_stack"] = np.bool_("stack" in ss.ctor)
        idx += 1
    # frequency mismatch -> ValueError (stream.py:81-82)
    try:
        t.process(FakeMap(freq=_freqmap(freqs + 1.0), map_=map_in))
        raised = "none"
    except ValueError as e:
        raised = str(e)
    cases["mismatch_message"] = np.array(raised)
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "stream_simulate.npz"), **cases)


class FakeProcMModes(_FakeCont):
    def __init__(self, vis, weight, freqs):
        self.vis, self.weight = _MPIDS(vis), _MPIDS(weight)
        self.index_map = {"m": np.arange(vis.shape[0]), "freq": _freqmap(freqs)}


def gen_mapmaker_process(mapmaker, out):
    """``BaseMapMaker.process`` (mapmaker.py:35-118) executed from the reference source for the three makers.

    ``hputil.sphtrans_inv_sky(alm, nside)`` [3P] is served by a recorder: the fixture stores the square
    ``alm [nfreq, 4, lmax+1, lmax+1]`` the reference hands to the SHT (frequency matching, m trim, the (m, f)
    loop, the broadcast into the four polarisation slots and the zero padding in m are all the reference's own).
    """
    import importlib
    import types

    _install_fake_mpi(mapmaker)
    mapmaker.containers = type("C", (), {"Map": FakeMap})
    mapmaker.io.get_beamtransfer = lambda b: b
    hp = importlib.import_module("cora.util.hputil")
    rec = {}

    def sphtrans_inv_sky(alm, nside):
        rec["alm"] = np.array(alm.view(np.ndarray))
        rec["nside"] = nside
        return np.zeros((alm.shape[0], 4, 12 * nside**2))

    hp.sphtrans_inv_sky = sphtrans_inv_sky
    importlib.import_module("cora.util").hputil = hp
    rng = np.random.default_rng(12012)
    cases = {}
    idx = 0
    # (npairs, lmax, tel_mmax, n_m of the data, npol, data frequencies as indices into the telescope's 4)
    for npairs, lmax, tel_mmax, n_m, npol, fsel in (
        (6, 5, 5, 6, 4, [0, 1, 2, 3]),   # matched
        (5, 6, 6, 4, 4, [2, 0]),         # data hold fewer m than the telescope; permuted frequency subset
        (7, 4, 3, 7, 4, [3, 1, 2]),      # data hold excess m (trimmed, :63-66)
        (4, 5, 5, 6, 1, [1, 3]),         # num_pol_sky = 1: broadcast into the four slots (:91-94)
    ):
        bt = _BT(npairs, lmax, 4, 700 + idx, npol)
        bt.telescope.mmax = tel_mmax
        tfreq = bt.telescope.frequencies
        dfreq = tfreq[fsel]
        mv = crandn(rng, (n_m, 2, len(fsel), npairs))
        mw = rng.uniform(0.5, 1.5, mv.shape)
        mw[rng.uniform(size=mw.shape) < 0.1] = 0.0
        c = f"c{idx}_"
        cases[c + "dims"] = np.array([npairs, lmax, tel_mmax, n_m, npol])
        cases[c + "tel_freq"], cases[c + "freq"] = tfreq, dfreq
        cases[c + "mvis"], cases[c + "mweight"] = mv, mw
        mm_eff = min(tel_mmax, n_m - 1)
        cases[c + "beam"] = np.array([[bt.beam_m(m, fi=f) for f in range(4)] for m in range(mm_eff + 1)])
        for name, cls in (("dirty", mapmaker.DirtyMapMaker), ("ml", mapmaker.MaximumLikelihoodMapMaker), ("wiener", mapmaker.WienerMapMaker)):
            t = cls.__new__(cls)
            t.nside, t.prior_amp, t.prior_tilt, t.bt_cache = 2, 1.0, 0.5, None
            t.setup(bt)
            rec.clear()
            try:
                m_out = t.process(FakeProcMModes(mv.copy(), mw.copy(), dfreq))
                cases[c + name] = rec["alm"]
                assert rec["nside"] == 2 and m_out.map.shape == (len(fsel), 4, 48)
            except ValueError as e:  # the Wiener prior is hard-wired to four polarisations (:264)
                cases[c + name + "_error"] = np.array(type(e).__name__)
        idx += 1
    # a data frequency the telescope does not have -> ValueError (tools.py:124-125 via mapmaker.py:59)
    try:
        t = mapmaker.DirtyMapMaker.__new__(mapmaker.DirtyMapMaker)
        t.nside, t.bt_cache = 2, None
        t.setup(bt)
        t.process(FakeProcMModes(mv.copy(), mw.copy(), dfreq + 0.5))
        raised = "none"
    except ValueError as e:
        raised = str(e)
    cases["mismatch_message"] = np.array(raised)
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "mapmaker_process.npz"), **cases)


class _ModeDS:
    """[m, ...] dataset whose full slice answers ``enumerate`` (fgfilter.py:85,129,190,226)."""

    def __init__(self, arr):
        self.arr = np.asarray(arr)

    def __getitem__(self, k):
        r = self.arr[k]
        return r.view(_EnumLocal) if isinstance(r, np.ndarray) else r

    def __setitem__(self, k, v):
        self.arr[k] = v

    @property
    def shape(self):
        return self.arr.shape


class FakeModeCont(_FakeCont):
    """SVDModes / KLModes duck type: vis, weight [m, mode], nmode [m] (containers.py:1196-1246)."""

    def __init__(self, mode=None, axes_from=None, attrs_from=None, vis=None, weight=None, nmode=None):
        if vis is None:
            n_m = axes_from.vis.shape[0]
            vis, weight, nmode = np.zeros((n_m, mode), complex), np.zeros((n_m, mode)), np.zeros(n_m, np.int32)
        self.vis, self.weight, self.nmode = _ModeDS(vis), _ModeDS(weight), _ModeDS(nmode)
        self.attrs = {}


class FakeFGMModes(_FakeCont):
    def __init__(self, freq=None, prod=None, input=None, attrs_from=None, axes_from=None, vis=None, weight=None):
        if vis is None:
            n_m = axes_from.vis.shape[0]
            vis = np.zeros((n_m, 2, len(freq), len(prod)), complex)
            weight = np.zeros(vis.shape)
            self.ctor = dict(freq=freq, prod=prod, input=input)
        self.vis, self.weight = _ModeDS(vis), _ModeDS(weight)
        self.attrs = {}


def gen_fgfilter(out):
    """``SVDModeProject`` / ``KLModeProject`` (fgfilter.py:53-239) run from the reference source against duck-typed
    products: the four driftscan projections [3P] are plain basis-matrix products (what their call sites imply), so
    the fixture pins the TASK logic -- packing of the per-frequency modes, nmode, zero padding, the median weight
    rule, the axis order of the backward transform -- not driftscan's arithmetic."""
    from draco.analysis import fgfilter

    fgfilter.containers = type("C", (), {"SVDModes": FakeModeCont, "KLModes": FakeModeCont, "MModes": FakeFGMModes})
    fgfilter.io.get_beamtransfer = lambda b: b
    rng = np.random.default_rng(13013)
    cases = {}
    idx = 0
    for n_m, nfreq, npairs, ndofmax in ((4, 3, 5, 20), (3, 2, 4, 16)):
        ntel = 2 * npairs
        lens = rng.integers(0, min(ntel, ndofmax // nfreq) + 1, size=(n_m, nfreq))
        lens[0, 0] = 0  # a frequency that keeps no mode
        ut = {(m, f): crandn(rng, (int(lens[m, f]), ntel)) for m in range(n_m) for f in range(nfreq)}
        uinv = {k: crandn(rng, (ntel, v.shape[0])) for k, v in ut.items()}

        class BT:
            pass

        bt = BT()
        bt.ndofmax = ndofmax
        bt.telescope = type("T", (), {})()
        bt.telescope.nfreq, bt.telescope.npairs, bt.telescope.nfeed = nfreq, npairs, 3
        bt.telescope.frequencies = 400.0 + 5.0 * np.arange(nfreq)
        bt.telescope.uniquepairs = np.array([(0, d) for d in range(npairs)])

        def t2s(mi, vec, ut=ut, nfreq=nfreq):
            return np.concatenate([ut[(mi, f)] @ vec[f] for f in range(nfreq)])

        def s2t(mi, svec, uinv=uinv, lens=lens, nfreq=nfreq, npairs=npairs):
            b = np.concatenate([[0], np.cumsum(lens[mi])])
            return np.stack([(uinv[(mi, f)] @ svec[b[f] : b[f + 1]]).reshape(2, npairs) for f in range(nfreq)])

        bt.project_vector_telescope_to_svd, bt.project_vector_svd_to_telescope = t2s, s2t
        mv = crandn(rng, (n_m, 2, nfreq, npairs))
        mw = rng.uniform(0.5, 1.5, mv.shape)
        mw[rng.uniform(size=mw.shape) < 0.2] = 0.0
        c = f"c{idx}_"
        cases[c + "dims"] = np.array([n_m, nfreq, npairs, ndofmax])
        cases[c + "lens"] = lens
        for (m, f), a in ut.items():
            cases[c + f"ut_{m}_{f}"], cases[c + f"uinv_{m}_{f}"] = a, uinv[(m, f)]
        cases[c + "mvis"], cases[c + "mweight"] = mv, mw
        t = fgfilter.SVDModeProject.__new__(fgfilter.SVDModeProject)
        t.setup(bt)
        t.mode = "forward"
        sv = t.process(FakeFGMModes(vis=mv.copy(), weight=mw.copy()))
        cases[c + "svd_vis"], cases[c + "svd_weight"], cases[c + "svd_nmode"] = sv.vis.arr, sv.weight.arr, sv.nmode.arr
        t.mode = "backward"
        sv_in = FakeModeCont(vis=sv.vis.arr.copy(), weight=rng.uniform(0.5, 1.5, sv.weight.arr.shape), nmode=sv.nmode.arr.copy())
        cases[c + "back_in_weight"] = sv_in.weight.arr.copy()
        mm = t.process(sv_in)
        cases[c + "back_vis"], cases[c + "back_weight"] = mm.vis.arr, mm.weight.arr
        cases[c + "back_in_nmode_after"] = sv_in.nmode.arr
        cases[c + "back_freq_centre"], cases[c + "back_freq_width"] = mm.ctor["freq"]["centre"], mm.ctor["freq"]["width"]
        cases[c + "back_input"] = np.asarray(mm.ctor["input"])
        t.mode = "filter"
        mf = t.process(FakeFGMModes(vis=mv.copy(), weight=mw.copy()))
        cases[c + "filter_vis"], cases[c + "filter_weight"] = mf.vis.arr, mf.weight.arr
        # KL on top of the SVD modes
        nsvd = sv.nmode.arr
        evals = {m: np.sort(rng.uniform(0.0, 10.0, size=max(int(nsvd[m]) - 1, 0))) for m in range(n_m)}
        evecs = {m: crandn(rng, (len(evals[m]), int(nsvd[m]))) for m in range(n_m)}
        kinv = {m: crandn(rng, (int(nsvd[m]), len(evals[m]))) for m in range(n_m)}

        class KL:
            def project_vector_svd_to_kl(self, mi, vec, threshold=None):
                keep = np.arange(len(evals[mi])) if threshold is None else np.flatnonzero(evals[mi] >= threshold)
                return evecs[mi][keep] @ vec

            def project_vector_kl_to_svd(self, mi, vec, threshold=None):
                keep = np.arange(len(evals[mi])) if threshold is None else np.flatnonzero(evals[mi] >= threshold)
                return kinv[mi][:, keep] @ vec

        pm = type("PM", (), {})()
        pm.beamtransfer, pm.kltransforms = bt, {"kl_a": KL()}
        for m in range(n_m):
            cases[c + f"kl_evals_{m}"], cases[c + f"kl_evecs_{m}"], cases[c + f"kl_inv_{m}"] = evals[m], evecs[m], kinv[m]
        for thr_name, thr in (("none", None), ("thr", 4.0)):
            k = fgfilter.KLModeProject.__new__(fgfilter.KLModeProject)
            k.setup(pm)
            k.klname, k.threshold = "kl_a", thr
            k.mode = "forward"
            kin = FakeModeCont(vis=sv.vis.arr.copy(), weight=sv.weight.arr.copy(), nmode=sv.nmode.arr.copy())
            km = k.process(kin)
            cases[c + f"kl_{thr_name}_vis"], cases[c + f"kl_{thr_name}_weight"], cases[c + f"kl_{thr_name}_nmode"] = km.vis.arr, km.weight.arr, km.nmode.arr
            k.mode = "backward"
            sb = k.process(FakeModeCont(vis=km.vis.arr.copy(), weight=km.weight.arr.copy(), nmode=km.nmode.arr.copy()))
            cases[c + f"klback_{thr_name}_vis"], cases[c + f"klback_{thr_name}_nmode"] = sb.vis.arr, sb.nmode.arr
        # a KL basis that is not there: backward raises RuntimeError (:213-217); forward trips over `self.kname` (:180)
        k.klname = "missing"
        k.mode = "backward"
        try:
            k.process(kin)
            err_b = "none"
        except Exception as e:
            err_b = type(e).__name__
        k.mode = "forward"
        try:
            k.process(kin)
            err_f = "none"
        except Exception as e:
            err_f = type(e).__name__
        cases[c + "missing_errors"] = np.array([err_f, err_b])
        idx += 1
    cases["ncase"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "fgfilter.npz"), **cases)



# --------------------------------------------------------------------------- ring-map chain (MakeVisGrid, BeamformNS, BeamformEW)
class _GridDS(_DS):
    """Dataset stub whose `[:]` has `local_array` and `enumerate(axis)` (one rank)."""

    class _A(_Arr):
        def enumerate(self, axis):
            return [(i, i) for i in range(self.shape[axis])]

    def __init__(self, arr):
        self.arr = arr.view(_GridDS._A)


class FakeGridIn(_FakeCont):
    """The SiderealStream MakeVisGrid.process reads (ringmapmaker.py:68-176)."""

    def __init__(self, vis, weight, prodstack, prod, rev_stack, input_flags, ra, freq):
        self.vis = _GridDS(vis)
        self.weight = _GridDS(weight)
        self.prodstack = prodstack
        self.input_flags = _DS(input_flags)
        self.index_map = {"prod": prod, "ra": ra, "freq": freq}
        self.reverse_map = {"stack": rev_stack}
        self.ra = ra
        self.attrs = {"tag": "chain"}
        self.freq = freq


class FakeVisGridStream(_FakeCont):
    def __init__(self, pol=None, ew=None, ns=None, ra=None, axes_from=None, attrs_from=None):
        nfreq = axes_from.vis.shape[0]
        self.index_map = {"pol": np.asarray(pol), "ew": np.asarray(ew), "ns": np.asarray(ns), "ra": np.asarray(ra), "freq": axes_from.freq}
        shp = (len(pol), nfreq, len(ew), len(ns), len(ra))
        self.vis = _GridDS(np.zeros(shp, np.complex64))
        self.weight = _GridDS(np.zeros(shp, np.float32))
        self.datasets = {"vis": self.vis, "vis_weight": self.weight}
        self.freq = np.asarray(axes_from.freq, dtype=float)
        self.attrs = dict(attrs_from.attrs)

    def add_dataset(self, name):
        assert name == "redundancy"
        p, _, e, n, r = self.vis.shape
        self.redundancy = _GridDS(np.zeros((p, e, n, r), np.int32))
        self.datasets["redundancy"] = self.redundancy


class FakeHybridOut(_FakeCont):
    def __init__(self, el=None, axes_from=None, attrs_from=None):
        g = axes_from
        p, f, e, _, r = g.vis.shape
        self.index_map = dict(g.index_map)
        self.index_map["el"] = np.asarray(el)
        self.vis = _GridDS(np.zeros((p, f, e, len(el), r), np.complex64))
        self.weight = _GridDS(np.zeros((p, f, e, r), np.float32))
        self.datasets = {"vis": self.vis, "vis_weight": self.weight}
        self.freq = g.freq
        self.attrs = dict(attrs_from.attrs)

    def add_dataset(self, name):
        assert name == "dirty_beam"
        self.dirty_beam = _GridDS(np.zeros(self.vis.shape, np.float32))
        self.datasets["dirty_beam"] = self.dirty_beam


class FakeChainRingMap(_FakeCont):
    def __init__(self, beam=None, pol=None, axes_from=None, attrs_from=None):
        h = axes_from
        _, f, _, nel, nra = h.vis.shape
        self.index_map = dict(h.index_map)
        self.index_map["pol"] = np.asarray(pol)
        self._shape = (beam, len(pol), f, nra, nel)
        self.map = _GridDS(np.zeros(self._shape, np.float64))  # (containers.py:1597-1645: float64 datasets)
        self.weight = _GridDS(np.zeros(self._shape[1:], np.float64))
        self.attrs = dict(attrs_from.attrs)

    def add_dataset(self, name):
        shp = {"rms": self._shape[1:4], "dirty_beam": self._shape}[name]
        setattr(self, name, _GridDS(np.zeros(shp, np.float64)))


class _ChainTel:
    """The telescope attributes MakeVisGrid reads: a two-cylinder grid, two polarisations, stacked products."""

    def __init__(self, ncyl=2, nfeed_cyl=4, cyl_sep=22.0, feed_sep=0.3048):
        pos, pol = [], []
        for c in range(ncyl):
            for y in range(nfeed_cyl):
                for pl in "XY":
                    pos.append((c * cyl_sep, y * feed_sep))
                    pol.append(pl)
        self.feedpositions = np.array(pos)
        self.polarisation = np.array(pol)
        n = len(pos)
        groups, prod = {}, []
        for i in range(n):
            for j in range(i, n):
                d = self.feedpositions[i] - self.feedpositions[j]
                key = (pol[i], pol[j], round(d[0], 4), round(d[1], 4))
                groups.setdefault(key, []).append(len(prod))
                prod.append((i, j))
        self.prod = np.array(prod, dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        keys = list(groups)
        self.uniquepairs = np.array([prod[groups[k][0]] for k in keys])
        self.baselines = np.array([self.feedpositions[i] - self.feedpositions[j] for i, j in self.uniquepairs])
        self.prodstack = np.array([tuple(x) for x in self.uniquepairs], dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        rev = np.zeros(len(prod), dtype=[("stack", "<u4"), ("conjugate", "u1")])
        for s_, k in enumerate(keys):
            for pidx in groups[k]:
                rev[pidx] = (s_, 0)
        self.reverse_stack = rev
        self.nfeed = n


def gen_ringmap_chain(out):
    """MakeVisGrid.process (ringmapmaker.py:68-176), BeamformNS.process (:230-346), BeamformEW.process (:372-497) run
    from the reference source on duck-typed containers.  `tools.calculate_redundancy`'s compiled helper is the NumPy
    loop used for CollateProducts above."""
    import importlib
    import types

    from draco.analysis import ringmapmaker as rmk

    ft = importlib.import_module("draco.util._fast_tools")

    def _calc_redundancy(input_flags, pm, stack_index, nstack, redundancy):
        for ii in range(pm.shape[0]):
            ist = stack_index[ii]
            if 0 <= ist < nstack:
                redundancy[ist] += input_flags[pm[ii, 0]] * input_flags[pm[ii, 1]]

    rmk.tools._calc_redundancy = _calc_redundancy
    rmk.containers = type("NS", (), {"VisGridStream": FakeVisGridStream, "HybridVisStream": FakeHybridOut, "RingMap": FakeChainRingMap})
    rmk.io.get_telescope = lambda t: t
    rmk.MPI = types.SimpleNamespace(MAX="max")
    rng = np.random.default_rng(8008)
    cases = {}
    tel = _ChainTel()
    nfreq, nra = 3, 10
    freq = np.array([420.0, 610.0, 780.0])
    fmap = np.zeros(nfreq, dtype=[("centre", float), ("width", float)])
    fmap["centre"], fmap["width"] = freq, 10.0
    ra = np.linspace(0.0, 360.0, nra, endpoint=False)
    nstack = len(tel.prodstack)
    vis = crandn(rng, (nfreq, nstack, nra), np.complex64)
    w = rng.uniform(0.5, 1.5, (nfreq, nstack, nra)).astype(np.float32)
    w[rng.uniform(size=w.shape) < 0.1] = 0.0
    flags = (rng.uniform(size=(tel.nfeed, nra)) > 0.15).astype(np.float32)
    cases["freq"], cases["ra"], cases["vis"], cases["weight"], cases["input_flags"] = freq, ra, vis, w, flags
    cases["feedpositions"], cases["polarisation"] = tel.feedpositions, tel.polarisation
    cases["prod"], cases["uniquepairs"], cases["rev_stack"] = tel.prod, tel.uniquepairs, tel.reverse_stack["stack"]

    class _Comm:
        def allreduce(self, x, op=None):
            return x

    grids = {}
    for centered in (False, True):
        t = rmk.MakeVisGrid.__new__(rmk.MakeVisGrid)
        t.centered, t.save_redundancy, t.telescope = centered, True, tel
        sin = FakeGridIn(vis.copy(), w.copy(), tel.prodstack, tel.prod, tel.reverse_stack, flags.copy(), ra, freq)
        sin.freq = freq
        g = t.process(sin)
        grids[centered] = g
        k = f"grid{int(centered)}"
        cases[k + "_vis"], cases[k + "_weight"] = g.vis.arr.view(np.ndarray), g.weight.arr.view(np.ndarray)
        cases[k + "_red"] = g.redundancy.arr.view(np.ndarray)
        cases[k + "_pol"], cases[k + "_ew"], cases[k + "_ns"] = g.index_map["pol"], g.index_map["ew"], g.index_map["ns"]
    # BeamformNS on the (uncentred) grid, every weight scheme
    idx = 0
    hybrids = []
    for weight, scaled, include_auto, sdb, npix, span in (("natural", False, False, True, 9, 1.0), ("uniform", False, True, False, 8, 0.8),
                                                          ("inverse_variance", False, False, True, 7, 1.0), ("hann", True, False, False, 6, 0.9),
                                                          ("blackman_harris", False, True, True, 5, 1.0)):
        t = rmk.BeamformNS.__new__(rmk.BeamformNS)
        t.npix, t.span, t.weight, t.scaled, t.include_auto, t.save_dirty_beam, t.precision = npix, span, weight, scaled, include_auto, sdb, 64
        t.comm, t.log = _Comm(), _Log()
        g = grids[False]
        hv = t.process(g)
        k = f"ns{idx}"
        cases[k + "_opts"] = np.array([weight, str(int(scaled)), str(int(include_auto)), str(int(sdb)), str(npix), repr(span)])
        cases[k + "_vis"], cases[k + "_weight"], cases[k + "_el"] = hv.vis.arr.view(np.ndarray), hv.weight.arr.view(np.ndarray), hv.index_map["el"]
        if sdb:
            cases[k + "_db"] = hv.dirty_beam.arr.view(np.ndarray)
        cases[k + "_nsmax"] = np.float64(hv.attrs["beamform_ns_nsmax"])
        hybrids.append(hv)
        idx += 1
    cases["n_ns"] = np.int64(idx)
    # BeamformEW on two of the hybrid streams
    idx = 0
    for hvi, excl, single, wew, flag in ((0, False, False, "natural", None), (1, True, False, "uniform", None), (2, False, True, "natural", None),
                                         (4, False, False, "natural", np.array([True, False]))):
        t = rmk.BeamformEW.__new__(rmk.BeamformEW)
        t.exclude_intracyl, t.single_beam, t.weight_ew, t.flag_ew = excl, single, wew, flag
        hin = hybrids[hvi]
        if "dirty_beam" in hin.datasets:
            # with a dirty beam in the input the reference fails on its own broadcast (ringmapmaker.py:489: weight_ew has
            # already been reshaped at :426): recorded once, then the case runs without the dirty beam
            if "ew_db_error" not in cases:
                try:
                    t.process(hin)
                    cases["ew_db_error"] = np.array("none")
                except Exception as exc:  # noqa: BLE001
                    cases["ew_db_error"] = np.array(type(exc).__name__)
            hin.datasets.pop("dirty_beam")
        rm = t.process(hin)
        k = f"ew{idx}"
        cases[k + "_opts"] = np.array([str(hvi), str(int(excl)), str(int(single)), wew, "" if flag is None else "".join(str(int(x)) for x in flag)])
        cases[k + "_map"], cases[k + "_weight"], cases[k + "_rms"] = rm.map.arr.view(np.ndarray), rm.weight.arr.view(np.ndarray), rm.rms.arr.view(np.ndarray)
        cases[k + "_pol"] = rm.index_map["pol"]
        if hasattr(rm, "dirty_beam"):
            cases[k + "_db"] = rm.dirty_beam.arr.view(np.ndarray)
        idx += 1
    cases["n_ew"] = np.int64(idx)
    np.savez_compressed(os.path.join(out, "ringmap_chain.npz"), **cases)


def main():
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle._refstub import load_reference

    transform, mapmaker = load_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    only = [a for a in sys.argv[1:] if a.startswith("--only-")]
    if not only:
        gen_transform(transform, GOLDEN)
        gen_mapmaker(mapmaker, GOLDEN)
    if not only or "--only-hybrid" in only:
        gen_hybrid(transform, GOLDEN)
    if not only or "--only-noise" in only:
        gen_noise(GOLDEN)
    if not only or "--only-mask" in only:
        gen_mask(GOLDEN)
    if not only or "--only-ringmap" in only:
        gen_ringmap(GOLDEN)
    if not only or "--only-ringmap-analytic" in only:
        gen_ringmap_analytic(GOLDEN)
    if not only or "--only-collate" in only:
        gen_collate(transform, GOLDEN)
    if not only or "--only-expand" in only:
        gen_expand(GOLDEN)
    if not only or "--only-svd" in only:
        gen_svd(GOLDEN)
    if not only or "--only-fgfilter" in only:
        gen_fgfilter(GOLDEN)
    if not only or "--only-simulate" in only:
        gen_stream_simulate(GOLDEN)
    if not only or "--only-process" in only:
        gen_mapmaker_process(mapmaker, GOLDEN)
    if not only or "--only-ringmap-chain" in only:
        gen_ringmap_chain(GOLDEN)
    for f in sorted(os.listdir(GOLDEN)):
        print(f, os.path.getsize(os.path.join(GOLDEN, f)))


if __name__ == "__main__":
    main()

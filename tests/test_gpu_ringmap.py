"""GPU parity: deconvolving ring-map makers vs outputs of the reference classes and the oracle.

float64 on the GPU; the reference mixes float32/float64 depending on the weight scheme
(inverse-variance weights keep the complex64 products in single precision), so cases with
inverse-variance weights agree to ~1e-6, the others to ~1e-12.  Asserted per scheme.
"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ringmap as orm


def _containers(hv, hw, bv, freq, ew, el, oddra):
    from draco_amd.core import containers

    nm, _, npol, nfreq, new, nel = hv.shape
    kw = dict(pol=npol, freq=freq, ew=ew, el=el)
    v = containers.HybridVisMModes(mmax=nm - 1, oddra=oddra, **kw)
    v.vis[:] = hv
    v.weight[:] = hw
    b = containers.HybridVisMModes(mmax=bv.shape[0] - 1, oddra=oddra, **kw)
    b.vis[:] = bv
    b.weight[:] = 1.0
    return v, b


def _task(kind, g, i):
    from draco_amd.analysis.ringmapmaker import TikhonovRingMapMaker, WienerRingMapMaker

    skip, oddra = (bool(x) for x in g[f"c{i}_opts"])
    excl = [int(x) for x in g[f"c{i}_exclude"]]
    inv_SN, gal_amp, psrc_amp = (float(x) for x in g[f"c{i}_params"])
    common = dict(exclude_cyl=excl, skip_deconvolution=skip, window_type=str(g[f"c{i}_window"]), save_dirty_beam=True)
    if kind == "tikhonov":
        t = TikhonovRingMapMaker(weight_ew=str(g[f"c{i}_weight_ew"]), inv_SN=inv_SN, **common)
    else:
        t = WienerRingMapMaker(gal_amp=gal_amp, psrc_amp=psrc_amp, **common)

    class Tel:
        latitude = float(g["latitude"])
        lmax = mmax = 1
        frequencies = g["freq"]

    t.setup(Tel())
    return t, oddra


def test_reference_golden(golden_dir):
    from draco_amd.core import containers

    g = np.load(os.path.join(golden_dir, "ringmap_deconvolve.npz"))
    for i in range(int(g["ncase"])):
        kind = str(g[f"c{i}_kind"])
        t, oddra = _task(kind, g, i)
        v, b = _containers(g[f"c{i}_hv"], g[f"c{i}_hw"], g[f"c{i}_bv"], g["freq"], g["ew"], g["el"], oddra)
        rm = t.process(v, b)
        assert isinstance(rm, containers.RingMap)
        # the reference forms |b|^2 (np.abs of complex64) and, with inverse-variance weights, the whole
        # product in float32: agreement with the float64 kernel is single precision; the float64 oracle
        # comparison below is the tight one
        tol = 5e-6
        for ds, name in ((rm.map, "map"), (rm.weight, "wgt"), (rm.dirty_beam_power, "dbp"), (rm.dirty_beam, "db")):
            ref = g[f"c{i}_{name}"]
            assert ds.shape == ref.shape and ds.dtype == np.float64, (i, name)
            if np.abs(ref).max() == 0:  # e.g. a window that vanishes at every measured m
                assert np.all(ds[:] == 0), (i, name)
                continue
            err = np.abs(ds[:] - ref).max() / np.abs(ref).max()
            assert err < tol, (i, name, err)
        assert rm.attrs["exclude_cyl"] == [int(x) for x in g[f"c{i}_exclude"]]


@pytest.mark.parametrize("nm,oddra,nel,new", [(33, False, 70, 3), (65, False, 40, 4), (20, True, 130, 2), (129, True, 9, 4)])
def test_vs_oracle_float64_inputs(nm, oddra, nel, new):
    """Larger shapes (nra = 64, 128 powers of two; 39 and 257 through Bluestein) against the float64 oracle."""
    from draco_amd.analysis.ringmapmaker import TikhonovRingMapMaker, WienerRingMapMaker

    rng = np.random.default_rng(nm)
    npol, nfreq = 2, 2
    freq = np.array([500.0, 700.0])
    ew = 22.0 * np.arange(new)
    el = np.linspace(-0.9, 0.9, nel)
    hv = (rng.standard_normal((nm, 2, npol, nfreq, new, nel)) + 1j * rng.standard_normal((nm, 2, npol, nfreq, new, nel))).astype(np.complex64)
    bv = (rng.standard_normal((nm, 2, npol, nfreq, new, nel)) + 1j * rng.standard_normal((nm, 2, npol, nfreq, new, nel))).astype(np.complex64)
    hw = rng.uniform(0.5, 1.5, (nm, 2, npol, nfreq, new)).astype(np.float32)
    hw[rng.uniform(size=hw.shape) < 0.1] = 0
    v, b = _containers(hv, hw, bv, freq, ew, el, oddra)
    # the oracle in float64 on the same (float32-valued) inputs
    args = dict(hv=hv.astype(np.complex128), hw=hw.astype(np.float64), bv=bv.astype(np.complex128), freq=freq, el=el, ew=ew, oddra=oddra)
    for task, okw in (
        (TikhonovRingMapMaker(weight_ew="inverse_variance", inv_SN=1e-2, exclude_cyl=[0], save_dirty_beam=True), dict(kind="tikhonov", weight_ew="inverse_variance", inv_SN=1e-2, exclude_cyl=[0])),
        (WienerRingMapMaker(save_dirty_beam=True), dict(kind="wiener")),
        (TikhonovRingMapMaker(weight_ew="natural", inv_SN=1e-4, skip_deconvolution=True, save_dirty_beam=True), dict(kind="tikhonov", weight_ew="natural", inv_SN=1e-4, skip_deconvolution=True)),
    ):
        task.setup()
        rm = task.process(v, b)
        rmm, rmw, rmbp, rmb = orm.deconvolve(**args, **okw)
        for ds, ref in ((rm.map, rmm), (rm.weight, rmw), (rm.dirty_beam_power, rmbp), (rm.dirty_beam, rmb)):
            err = np.abs(ds[:] - ref).max() / np.abs(ref).max()
            assert err < 1e-11, (type(task).__name__, err)


@pytest.mark.parametrize("nm,oddra,nel,new,npol,nfreq", [(33, False, 7, 3, 1, 1), (17, True, 65, 2, 1, 3), (9, False, 3, 40, 1, 1), (65, True, 129, 4, 2, 1)])
def test_vs_oracle_without_dirty_beam_output(nm, oddra, nel, new, npol, nfreq):
    """The default path (RA-space dirty beam not kept): two rows share one inverse FFT and the dirty-beam power comes from
    the modes by Parseval.  Odd row counts (a transform with a lone row), elevation counts that are not multiples of the
    tile, more than 64 (sign, EW) terms (the weights are handed round the wave in two rounds), every weight scheme."""
    from draco_amd.analysis.ringmapmaker import TikhonovRingMapMaker, WienerRingMapMaker

    rng = np.random.default_rng(nm * 7 + nel)
    freq = np.linspace(500.0, 700.0, nfreq)
    ew = 22.0 * np.arange(new)
    el = np.linspace(-0.9, 0.9, nel)
    shp = (nm, 2, npol, nfreq, new, nel)
    hv = (rng.standard_normal(shp) + 1j * rng.standard_normal(shp)).astype(np.complex64)
    bv = (rng.standard_normal(shp) + 1j * rng.standard_normal(shp)).astype(np.complex64)
    hw = rng.uniform(0.5, 1.5, shp[:-1]).astype(np.float32)
    hw[rng.uniform(size=hw.shape) < 0.1] = 0
    v, b = _containers(hv, hw, bv, freq, ew, el, oddra)
    args = dict(hv=hv.astype(np.complex128), hw=hw.astype(np.float64), bv=bv.astype(np.complex128), freq=freq, el=el, ew=ew, oddra=oddra)
    for task, okw in (
        (TikhonovRingMapMaker(weight_ew="inverse_variance", inv_SN=1e-2), dict(kind="tikhonov", weight_ew="inverse_variance", inv_SN=1e-2)),
        (TikhonovRingMapMaker(weight_ew="uniform", inv_SN=1e-3), dict(kind="tikhonov", weight_ew="uniform", inv_SN=1e-3)),
        (WienerRingMapMaker(), dict(kind="wiener")),
        (TikhonovRingMapMaker(weight_ew="natural", inv_SN=1e-4, skip_deconvolution=True), dict(kind="tikhonov", weight_ew="natural", inv_SN=1e-4, skip_deconvolution=True)),
    ):
        task.setup()
        rm = task.process(v, b)
        rmm, rmw, rmbp, _ = orm.deconvolve(**args, **okw)
        for ds, ref in ((rm.map, rmm), (rm.weight, rmw), (rm.dirty_beam_power, rmbp)):
            err = np.abs(ds[:] - ref).max() / np.abs(ref).max()
            assert err < 1e-11, (type(task).__name__, err)


def test_validation_errors():
    from draco_amd.analysis.ringmapmaker import TikhonovRingMapMaker

    rng = np.random.default_rng(0)
    hv = np.zeros((5, 2, 1, 2, 2, 3), np.complex64)
    hw = np.ones((5, 2, 1, 2, 2), np.float32)
    freq, ew, el = np.array([500.0, 600.0]), np.array([0.0, 22.0]), np.array([-0.1, 0.0, 0.1])
    v, b = _containers(hv, hw, hv, freq, ew, el, False)
    t = TikhonovRingMapMaker()
    t.setup()
    _, b2 = _containers(hv, hw, hv, freq + 1, ew, el, False)
    with pytest.raises(ValueError, match="Frequencies do not match"):
        t.process(v, b2)
    _, b3 = _containers(hv, hw, hv[:3], freq, ew, el, False)
    with pytest.raises(ValueError, match="higher m-max"):
        t.process(v, b3)
    tw = TikhonovRingMapMaker(window_type="hann")
    with pytest.raises(RuntimeError, match="Must provide manager"):
        tw.setup()


_POL = np.array(["XX", "XY", "YX", "YY"])


class _Tel:
    lmax = mmax = 1
    frequencies = np.array([600.0])

    def __init__(self, latitude):
        self.latitude = latitude


def test_analytic_beam_reference_golden(golden_dir):
    """``...RingMapMakerAnalytical`` against the reference's own ``_get_beam_mmodes`` + ``process`` outputs.

    Beam m-modes are complex64 built from a float64 FFT on both sides: agreement is the rounding of the complex64
    store (a last-bit flip where device and host libm differ by an ulp), stated as 2e-7 of the largest mode."""
    from draco_amd.analysis.ringmapmaker import TikhonovRingMapMakerAnalytical, WienerRingMapMakerAnalytical
    from draco_amd.core import containers

    g = np.load(os.path.join(golden_dir, "ringmap_analytic.npz"))
    for i in range(int(g["ncase"])):
        hv, hw, el = g[f"c{i}_hv"], g[f"c{i}_hw"], g[f"c{i}_el"]
        kind = str(g[f"c{i}_kind"])
        cls = TikhonovRingMapMakerAnalytical if kind == "tikhonov" else WienerRingMapMakerAnalytical
        t = cls(save_dirty_beam=True, **({"inv_SN": 1e-3} if kind == "tikhonov" else {}))
        t.setup(_Tel(float(g["latitude"])))
        v = containers.HybridVisMModes(mmax=hv.shape[0] - 1, oddra=bool(g[f"c{i}_oddra"]), pol=_POL, freq=g["freq"], ew=g["ew"], el=el)
        v.vis[:] = hv
        v.weight[:] = hw
        bm = t._get_beam_mmodes(v)
        ref = g[f"c{i}_beam_m"]
        got = bm.vis[:]
        assert got.shape == ref.shape and got.dtype == np.complex64
        assert np.abs(got - ref).max() < 2e-7 * np.abs(ref).max(), i
        rm = t.process(v)
        for ds, name in ((rm.map, "map"), (rm.weight, "wgt"), (rm.dirty_beam_power, "dbp"), (rm.dirty_beam, "db")):
            r = g[f"c{i}_{name}"]
            assert np.abs(ds[:] - r).max() < 5e-6 * np.abs(r).max(), (i, name)


@pytest.mark.parametrize("mmax,oddra,nel,nfreq", [(64, False, 33, 2), (100, True, 7, 3), (512, False, 5, 1), (300, False, 4, 2), (1024, True, 2, 1)])
def test_analytic_beam_vs_oracle(mmax, oddra, nel, nfreq):
    """nra = 128 / 1024 (radix-2) and 201 / 600 / 2049 (Bluestein; the last at M = 8192) against the float64 oracle."""
    from draco_amd.analysis.ringmapmaker import TikhonovRingMapMakerAnalytical
    from draco_amd.core import containers

    freq = np.linspace(400.0, 800.0, nfreq, endpoint=False)
    ew = np.array([0.0, 22.0, 44.0, 66.0])
    el = np.linspace(-0.95, 0.95, nel)
    t = TikhonovRingMapMakerAnalytical()
    t.setup(_Tel(49.32))
    v = containers.HybridVisMModes(mmax=mmax, oddra=oddra, pol=_POL, freq=freq, ew=ew, el=el)
    got = t._get_beam_mmodes(v).vis[:]
    ref = orm.analytic_beam_mmodes(freq, ew, el, _POL, 49.32, mmax, oddra)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-7 * np.abs(ref).max()
    # structure: the -m half of m = 0 and (even nra) of the Nyquist row stays zero like _make_marray's
    assert np.all(got[0, 1] == 0)
    if not oddra:
        assert np.all(got[mmax, 1] == 0)


@pytest.mark.parametrize("nm,nel", [(257, 77), (1025, 37)])  # nra 512: all sixteen elevations' image in the LDS; nra 2048: the second eight parked
def test_single_pass_kernel_against_the_three_kernel_form(nm, nel):
    """The single-pass kernel (power-of-two nra, no RA-space dirty beam, own-row normalisation: reduce, inverse FFT and
    the [ra][el] store in one pass over the m-modes; 16 elevations per block, the second eight parked in a scratch image,
    and the 8-elevation form of round 3) against the three-kernel form on the same
    inputs ("ringmap_variant" = 1), at a shape with several m passes, partial elevation tiles and an odd row count."""
    import torch

    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    npol, nfreq, new = 2, 3, 4
    nra = 2 * (nm - 1)
    gen = torch.Generator(device=ctx.device).manual_seed(3)
    shp = (nm, 2, npol, nfreq, new, nel)
    hv = torch.randn(shp, dtype=torch.complex64, device=ctx.device, generator=gen)
    bv = torch.randn(shp, dtype=torch.complex64, device=ctx.device, generator=gen)
    hw = torch.rand(shp[:-1], dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
    hw[torch.rand(shp[:-1], device=ctx.device, generator=gen) < 0.1] = 0
    table = ctx.to_device(np.array([0.0, 1.0, 1.0, 1.0]), np.float64)  # first cylinder pair excluded
    eps = ctx.to_device(np.full((nfreq, nm), 1e-2), np.float64)
    win = torch.rand((nfreq, nm, nel), dtype=torch.float32, device=ctx.device, generator=gen)
    out = {}
    try:
        for variant in (0, 1, 2):  # 0: single pass, 16 elevations per block; 2: single pass, 8 per block; 1: three kernels
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ringmap_variant", variant))
            for mode in (0, 1, 2):
                rmap = ctx.zeros((1, npol, nfreq, nra, nel), np.float64)
                rwgt = ctx.zeros((npol, nfreq, nra, nel), np.float64)
                rdbp = ctx.zeros((1, npol, nfreq, nel), np.float64)
                _lib.check(_lib.lib.dmm_ringmap_deconvolve(ctx.handle, nm, nm, npol, nfreq, new, nel, nra, mode, 0, 0, ptr(hv), ptr(hw), ptr(bv),
                                                           ptr(table), ptr(eps), ptr(win), ptr(rmap), ptr(rwgt), ptr(rdbp), None))
                out[variant, mode] = [x.cpu().numpy() for x in (rmap, rwgt, rdbp)]
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ringmap_variant", 0))
    for mode in (0, 1, 2):
        for v in (0, 2):
            for a, b in zip(out[v, mode], out[1, mode]):
                assert np.all(np.isfinite(a))
                assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (v, mode)
        for a, b in zip(out[0, mode], out[2, mode]):  # the two single-pass forms differ in the order of the per-row sums over m only
            assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max(), mode

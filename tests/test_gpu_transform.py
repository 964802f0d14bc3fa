"""GPU parity: m-mode transform kernels vs the oracle and the reference's golden vectors.

Tolerances: the forward FFT is single precision in the reference too (complex64 FFT,
transform.py:689) so forward results agree to a few 1e-7 of the largest mode; we assert
< 2e-6 relative to max|F| (SURVEY.md section 0.7 budget: ~1e-6).  The inverse is computed in
float64 and rounded once to complex64: < 2e-7 relative.
"""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import transform as otr

FWD_TOL = 2e-6
INV_TOL = 2e-7


@pytest.fixture(scope="module")
def T():
    from draco_amd.analysis import transform

    return transform


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_make_marray_golden(T, golden_dir):
    g = np.load(os.path.join(golden_dir, "transform_make_marray.npz"))
    for i in range(int(g["ncase"])):
        ts, mmax = g[f"c{i}_ts"], int(g[f"c{i}_mmax"])
        ref = g[f"c{i}_out128"]
        out = T._make_marray(ts, mmax=mmax, dtype=np.complex128)
        assert out.shape == ref.shape
        assert _rel(out, ref) < FWD_TOL, f"case {i}"
        # unfilled slots are exact zeros, filled slots are not
        assert np.array_equal(out == 0, ref == 0), f"case {i} fill pattern"
        mm = np.full(ref.shape, 7 + 7j, np.complex128)  # must be fully overwritten
        T._make_marray(ts, mm)
        assert _rel(mm, ref) < FWD_TOL


def test_make_marray_errors(T):
    ts = np.zeros((2, 3, 8), np.complex64)
    with pytest.raises(ValueError, match="One of `mmodes` or `mmax`"):
        T._make_marray(ts)
    with pytest.raises(ValueError, match="mmax must be None"):
        T._make_marray(ts, np.zeros((5, 2, 2, 3), np.complex64), mmax=4)
    with pytest.raises(ValueError, match="incompatible shapes"):
        T._make_marray(ts, np.zeros((5, 2, 3, 2), np.complex64))


@pytest.mark.parametrize(
    "N,mmax,nrow",
    [(1, 0, 3), (2, 1, 5), (8, 4, 1), (64, 32, 37), (64, 10, 16), (512, 256, 33), (1024, 512, 50), (1024, 700, 9),
     (4096, 2048, 5), (8192, 100, 3), (15, 7, 21), (127, 63, 40), (127, 80, 7), (1000, 500, 11), (2047, 1023, 6), (4095, 64, 2), (3, 1, 4)],
)
def test_forward_vs_oracle(T, N, mmax, nrow):
    rng = np.random.default_rng(N * 7919 + mmax)
    ts = (rng.standard_normal((nrow, N)) + 1j * rng.standard_normal((nrow, N))).astype(np.complex64)
    ref = otr.make_marray(ts, mmax=mmax, dtype=np.complex128)
    exact = otr.make_marray(ts.astype(np.complex128), mmax=mmax, dtype=np.complex128)
    out = T._make_marray(ts, mmax=mmax, dtype=np.complex128)
    assert _rel(out, ref) < FWD_TOL
    # and we are not worse than ~the reference's own single-precision error vs exact arithmetic
    assert _rel(out, exact) < max(4 * _rel(ref, exact), 1e-6)
    assert np.array_equal(out == 0, ref == 0)


def test_mmode_task_golden(T, golden_dir):
    from draco_amd.core import containers
    from draco_amd.core.products import TransitTelescope

    g = np.load(os.path.join(golden_dir, "transform_mmode_task.npz"))
    for i in range(int(g["ncase"])):
        vis, w = g[f"c{i}_vis"], g[f"c{i}_weight"]
        mmax = int(g[f"c{i}_mmax"])
        ss = containers.SiderealStream(freq=np.arange(vis.shape[0]) + 400.0, ra=vis.shape[-1], stack=vis.shape[1])
        ss.vis[:] = vis
        ss.weight[:] = w
        ss.attrs["tag"] = "keep"
        task = T.MModeTransform(remove_integration_window=bool(g[f"c{i}_window"]))
        task.setup(None if mmax < 0 else TransitTelescope(np.arange(3.0), lmax=mmax))
        ma = task.process(ss)
        assert isinstance(ma, containers.MModes) and ma.attrs["tag"] == "keep"
        assert ma.attrs["oddra"] == bool(g[f"c{i}_oddra"])
        ref_v, ref_w = g[f"c{i}_mvis"], g[f"c{i}_mweight"]
        assert ma.vis.shape == ref_v.shape and ma.vis.dtype == np.complex128 and ma.weight.dtype == np.float64
        assert _rel(ma.vis[:], ref_v) < FWD_TOL, f"case {i}"
        # reference weights are float32 arithmetic stored as float64: relative 2e-6; zeros exact
        np.testing.assert_allclose(ma.weight[:], ref_w, rtol=2e-6, atol=0, err_msg=f"case {i}")
        assert np.array_equal(ma.weight[:] == 0, ref_w == 0)


def test_hybrid_stream_golden(T, golden_dir):
    """HybridVisStream -> HybridVisMModes (complex64 / float32 outputs, weight without the el axis)."""
    from draco_amd.core import containers
    from draco_amd.core.products import TransitTelescope

    g = np.load(os.path.join(golden_dir, "transform_hybrid_task.npz"))
    for i in range(int(g["ncase"])):
        vis, w = g[f"c{i}_vis"], g[f"c{i}_weight"]
        mmax = int(g[f"c{i}_mmax"])
        npol, nfreq, new, nel, nra = vis.shape
        hs = containers.HybridVisStream(pol=npol, freq=np.arange(nfreq) + 400.0, ew=new, el=nel, ra=nra)
        hs.vis[:] = vis
        hs.weight[:] = w
        task = T.MModeTransform(remove_integration_window=bool(g[f"c{i}_window"]))
        task.setup(None if mmax < 0 else TransitTelescope(np.arange(3.0), lmax=mmax))
        ma = task.process(hs)
        assert isinstance(ma, containers.HybridVisMModes)
        ref_v, ref_w = g[f"c{i}_mvis"], g[f"c{i}_mweight"]
        assert ma.vis.shape == ref_v.shape and ma.vis.dtype == np.complex64
        assert ma.weight.shape == ref_w.shape and ma.weight.dtype == np.float32
        assert _rel(ma.vis[:], ref_v) < FWD_TOL
        np.testing.assert_allclose(ma.weight[:], ref_w, rtol=2e-6)


def test_unsupported_container_keyerror(T):
    class Other:
        pass

    with pytest.raises(KeyError):
        T.MModeTransform().process(Other())


def test_inverse_golden(T, golden_dir):
    g = np.load(os.path.join(golden_dir, "transform_unpack.npz"))
    for i in range(int(g["ncase"])):
        n = int(g[f"c{i}_n"])
        ss = T._make_ssarray(g[f"c{i}_mm"], n=None if n < 0 else n)
        ref = g[f"c{i}_ss"]
        assert ss.shape == ref.shape
        assert _rel(ss, ref) < INV_TOL, f"case {i}"


def test_inverse_task_golden(T, golden_dir):
    from draco_amd.core import containers

    g = np.load(os.path.join(golden_dir, "transform_inverse_task.npz"))
    for i in range(int(g["ncase"])):
        mv, mw = g[f"c{i}_mvis"], g[f"c{i}_mweight"]
        nra = int(g[f"c{i}_nra"])
        mm = containers.MModes(mmax=mv.shape[0] - 1, oddra=bool(g[f"c{i}_oddra"]), freq=np.arange(mv.shape[2]) + 400.0, stack=mv.shape[3])
        mm.vis[:] = mv
        mm.weight[:] = mw
        task = T.MModeInverseTransform(apply_integration_window=bool(g[f"c{i}_window"]))
        task.nra = None if nra < 0 else nra
        ss = task.process(mm)
        ref_v, ref_w = g[f"c{i}_vis"], g[f"c{i}_weight"]
        assert ss.vis.shape == ref_v.shape and ss.vis.dtype == np.complex64 and ss.weight.dtype == np.float32
        assert _rel(ss.vis[:], ref_v) < INV_TOL, f"case {i}"
        np.testing.assert_allclose(ss.weight[:], ref_w, rtol=1e-6)
        # the input container is not modified (documented difference from the reference)
        assert np.array_equal(mm.vis[:], mv)


@pytest.mark.parametrize("N", [64, 127, 1000, 1024, 2047, 2049, 4095])  # the last two: Bluestein at M = 8192, twiddles read from memory
def test_inverse_vs_oracle(T, N):
    rng = np.random.default_rng(N)
    mmax = N // 2
    ts = rng.standard_normal((13, N)) + 1j * rng.standard_normal((13, N))
    mm = otr.make_marray(ts, mmax=mmax, dtype=np.complex128)
    ref = otr.make_ssarray(mm)
    out = T._make_ssarray(mm)
    assert out.shape == ref.shape
    assert _rel(out, ref) < INV_TOL
    assert _rel(out, ts) < INV_TOL


def test_roundtrip_full_size(T):
    """cfg-3 shaped round trip on the device: inverse(forward(x)) == x (size-independent property)."""
    import torch

    from draco_amd.device import Context

    ctx = Context.get()
    nfreq, npairs, nra = 64, 379, 1024  # a quarter of cfg 3's rows keeps the test quick
    gen = torch.Generator(device=ctx.device).manual_seed(5)
    vis = torch.randn((nfreq, npairs, nra), dtype=torch.complex64, device=ctx.device, generator=gen)
    w = torch.rand((nfreq, npairs, nra), dtype=torch.float32, device=ctx.device, generator=gen) + 0.5
    mv, mw = T.mmode_forward(ctx, vis, w, nra // 2)
    assert mv.shape == (nra // 2 + 1, 2, nfreq, npairs)
    back = T.mmode_inverse(ctx, mv, nra)
    err = (back - vis).abs().max().item() / vis.abs().max().item()
    assert err < 3e-6
    # Parseval with modes = F/N: sum_t |x|^2 / N == sum_m |modes|^2, per row, in float64
    p_t = (vis.to(torch.complex128).abs() ** 2).sum(-1) / nra
    p_m = (mv.abs() ** 2).sum((0, 1))
    assert ((p_t - p_m).abs().max() / p_t.max()).item() < 1e-5
    # weights: nra^2 / sum(1/w), identical for every (m, sign)
    ws = nra**2 / (1.0 / w.double()).sum(-1)
    assert ((mw[5, 1] - ws).abs().max() / ws.max()).item() < 1e-12
    assert torch.equal(mw[0, 0], mw[nra // 2, 1])


def test_linearity(T):
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((5, 127)) + 1j * rng.standard_normal((5, 127))).astype(np.complex64)
    b = (rng.standard_normal((5, 127)) + 1j * rng.standard_normal((5, 127))).astype(np.complex64)
    fa, fb = T._make_marray(a, mmax=63, dtype=np.complex128), T._make_marray(b, mmax=63, dtype=np.complex128)
    fab = T._make_marray((a + b).astype(np.complex64), mmax=63, dtype=np.complex128)
    assert _rel(fab, fa + fb) < 2e-6


@pytest.mark.parametrize("nra_in,nra_out,window", [(16, 24, False), (16, 10, False), (15, 15, True), (30, 64, True)])
def test_sidereal_mmode_resample(nra_in, nra_out, window):
    """SiderealMModeResample (transform.py:795-811) = forward then inverse transform; against the oracle chain."""
    from draco_amd.analysis.transform import SiderealMModeResample
    from draco_amd.core import containers
    from oracle import transform as otr

    rng = np.random.default_rng(nra_in * 100 + nra_out)
    nfreq, nstack = 3, 4
    vis = (rng.standard_normal((nfreq, nstack, nra_in)) + 1j * rng.standard_normal((nfreq, nstack, nra_in))).astype(np.complex64)
    w = rng.uniform(0.5, 1.5, vis.shape).astype(np.float32)
    ss = containers.SiderealStream(freq=400.0 + np.arange(nfreq), ra=nra_in, stack=nstack)
    ss.vis[:] = vis
    ss.weight[:] = w
    out = SiderealMModeResample(nra=nra_out, remove_integration_window=window, apply_integration_window=window).process(ss)
    assert out.vis[:].shape == (nfreq, nstack, nra_out)
    mv, mw = otr.mmode_transform(vis, w, mmax=nra_in // 2, remove_integration_window=window)
    ref_v, ref_w = otr.mmode_inverse_transform(mv, mw, bool(nra_in % 2), nra=nra_out, apply_integration_window=window)
    assert np.abs(out.vis[:] - ref_v).max() < 3e-6 * np.abs(ref_v).max()
    assert np.abs(out.weight[:] - ref_w).max() < 3e-6 * np.abs(ref_w).max()


@pytest.mark.parametrize("nrow,nra,mmax,window", [(64, 1024, 512, False), (65, 1024, 100, False), (7, 1022, 300, True), (130, 1001, 500, False), (1, 4, 2, True), (33, 8, 3, False)])
def test_weight_reduction_all_load_store_forms(nrow, nra, mmax, window):
    """`k_mmode_weight`: rows read four weights per load when ``nra % 4 == 0`` and one at a time otherwise; the (m, +/-)
    broadcast stored two rows per lane when ``nrow`` is even, one otherwise.  Each combination against
    transform.py:599-602,627,638-639 in float64: ``nra^2 / sum(1 / w)`` with zeros skipped, zero for an all-zero row."""
    import torch

    from draco_amd.analysis.transform import mmode_forward
    from draco_amd.device import Context

    ctx = Context.get()
    rng = np.random.default_rng(nrow * 131 + nra)
    w = rng.uniform(0.5, 40.0, (nrow, nra)).astype(np.float32)
    w[rng.uniform(size=w.shape) < 0.05] = 0.0
    w[nrow // 2] = 0.0  # one row without any valid sample
    vis = (rng.standard_normal((nrow, nra)) + 1j * rng.standard_normal((nrow, nra))).astype(np.complex64)
    _, mw = mmode_forward(ctx, torch.from_numpy(vis).to(ctx.device), torch.from_numpy(w).to(ctx.device), mmax, remove_integration_window=window)
    inv = np.where(w != 0, 1.0 / np.where(w != 0, w, 1.0).astype(np.float64), 0.0).sum(axis=1)
    ws = np.where(inv != 0, float(nra) ** 2 / np.where(inv != 0, inv, 1.0), 0.0)
    ref = np.broadcast_to(ws, (mmax + 1, 2, nrow)).copy()
    if window:
        ref *= (np.sinc(np.arange(mmax + 1) / nra) ** 2)[:, None, None]
    out = mw.cpu().numpy()
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, rtol=1e-13, atol=0)
    assert np.all(out[:, :, nrow // 2] == 0)

"""Host-side noise tasks vs vectors produced by the reference's own code (tests/golden/noise.npz).

The product draws from a NumPy Generator in the reference's order, so a seeded run must
reproduce the reference's stream exactly (float rounding aside for the task-level cases).
"""

import os

import numpy as np

from draco_amd.core import containers
from draco_amd.synthesis.noise import GaussianNoise, ReceiverTemperature, SampleNoise
from draco_amd.util import random


def test_random_draws(golden_dir):
    g = np.load(os.path.join(golden_dir, "noise.npz"))
    cn = random.complex_normal(size=(3, 4), scale=np.array([1.0, 2.0, 3.0, 4.0]), rng=np.random.default_rng(5))
    assert np.array_equal(cn, g["cn"])
    cn64 = random.complex_normal(size=(2, 5), dtype=np.complex64, loc=1 + 2j, rng=np.random.default_rng(6))
    assert cn64.dtype == np.complex64 and np.array_equal(cn64, g["cn64"])
    np.testing.assert_allclose(random.standard_complex_wishart(4, 50, rng=np.random.default_rng(7)), g["scw"], rtol=1e-14)
    np.testing.assert_allclose(random.complex_wishart(g["cw_C"], 100, rng=np.random.default_rng(8)), g["cw"], rtol=1e-13)


def _stream(vis, weight, ninput, width=0.390625):
    nfreq, nprod, nra = vis.shape
    prod = np.array([(i, j) for i in range(ninput) for j in range(i, ninput)], dtype=[("input_a", int), ("input_b", int)])
    fm = np.zeros(nfreq, dtype=[("centre", float), ("width", float)])
    fm["centre"] = 400.0 + np.arange(nfreq)
    fm["width"] = width
    ss = containers.SiderealStream(freq=fm, ra=nra, prod=prod, input=ninput)
    ss.vis[:] = vis
    ss.weight[:] = weight
    return ss


def test_gaussian_noise_task(golden_dir):
    g = np.load(os.path.join(golden_dir, "noise.npz"))
    vis = g["gn_vis_in"]
    ss = _stream(vis, np.ones(vis.shape, np.float32), 3)
    t = GaussianNoise(ndays=2.0)
    t.setup()
    t.rng = np.random.default_rng(9)
    out = t.process(ss)
    assert out is ss
    np.testing.assert_allclose(ss.vis[:], g["gn_vis"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ss.weight[:], g["gn_weight"], rtol=1e-6)
    autos = [0, 3, 5]
    assert np.all((ss.vis[:] - vis)[:, autos].imag == 0)  # autos get real noise only


def test_gaussian_noise_stacked_uses_redundancy():
    from draco_amd.core.products import TransitTelescope

    tel = TransitTelescope(np.array([400.0, 401.0]), lmax=3, ncyl=1, nfeed_cyl=2)
    fm = np.zeros(2, dtype=[("centre", float), ("width", float)])
    fm["centre"], fm["width"] = tel.frequencies, 0.39
    ss = containers.SiderealStream(freq=fm, ra=8, prod=tel.index_map_prod, stack=tel.index_map_stack, input=tel.nfeed)
    t = GaussianNoise(add_noise=False)
    t.setup(tel)
    t.process(ss)
    w = ss.weight[:][0, :, 0]
    assert np.allclose(w / w.min(), tel.redundancy / tel.redundancy.min())
    bad = containers.SiderealStream(freq=fm, ra=8, stack=5, input=tel.nfeed, prod=tel.index_map_prod[:5])
    import pytest

    with pytest.raises(ValueError, match="Unexpected number of products"):
        t.process(bad)


def test_sample_noise_task(golden_dir):
    g = np.load(os.path.join(golden_dir, "noise.npz"))
    exp = g["sn_vis_in"]
    ss = _stream(exp.astype(np.complex64), np.ones(exp.shape, np.float32), 3)
    ss.datasets["vis"] = containers.Dataset(host=exp.copy())  # keep float64 like the reference run
    ss.datasets["vis_weight"] = containers.Dataset(host=np.ones(exp.shape))
    t = SampleNoise(sample_frac=1e-4)
    t.rng = np.random.default_rng(10)
    t.process(ss)
    np.testing.assert_allclose(ss.vis[:], g["sn_vis"], rtol=1e-12)
    np.testing.assert_allclose(ss.weight[:], g["sn_weight"], rtol=1e-12)


def test_receiver_temperature_task(golden_dir):
    """ReceiverTemperature.process (noise.py:21-45) against the reference class run from source: the offset reaches the
    auto-correlations only."""
    g = np.load(os.path.join(golden_dir, "noise.npz"))
    ss = _stream(g["gn_vis_in"].copy(), np.ones(g["gn_vis_in"].shape, np.float32), 3)
    t = ReceiverTemperature(recv_temp=42.5)
    out = t.process(ss)
    assert out is ss
    assert np.array_equal(ss.vis[:], g["rt_vis"])

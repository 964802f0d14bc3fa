"""GPU parity: SVDModeProject / KLModeProject against the reference classes run from source (tests/golden/fgfilter.npz).

The basis matrices are inputs (driftscan's arithmetic is absent: unpinned); the task logic around them is the
reference's own.  Plus: the batched GEMV and the row median kernels against NumPy on awkward shapes.
"""

import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "fgfilter.npz"))


def _products(g, i):
    from draco_amd.core.products import KLTransform, ProductManager, SVDArrayProvider

    c = f"c{i}_"
    n_m, nfreq, npairs, ndofmax = (int(x) for x in g[c + "dims"])
    tel = types.SimpleNamespace(nfreq=nfreq, npairs=npairs, nfeed=3, lmax=4, mmax=n_m - 1, num_pol_sky=4,
                                frequencies=400.0 + 5.0 * np.arange(nfreq), uniquepairs=np.array([(0, d) for d in range(npairs)]))
    bt = SVDArrayProvider(tel, lambda m, f: None, lambda m, f: g[c + f"ut_{m}_{f}"], ndofmax, ut_inv=lambda m, f: g[c + f"uinv_{m}_{f}"])
    kl = KLTransform(lambda m: (g[c + f"kl_evals_{m}"], g[c + f"kl_evecs_{m}"], g[c + f"kl_inv_{m}"]))
    return c, n_m, nfreq, npairs, ndofmax, tel, bt, ProductManager(bt, {"kl_a": kl})


def test_svd_and_kl_projection_golden(golden_dir):
    from draco_amd.analysis.fgfilter import KLModeProject, SVDModeProject
    from draco_amd.core import containers

    g = _load(golden_dir)
    for i in range(int(g["ncase"])):
        c, n_m, nfreq, npairs, ndofmax, tel, bt, pm = _products(g, i)
        mm = containers.MModes(mmax=n_m - 1, freq=tel.frequencies, stack=npairs)
        mm.vis[:] = g[c + "mvis"]
        mm.weight[:] = g[c + "mweight"]
        t = SVDModeProject(mode="forward")
        t.setup(bt)
        sv = t.process(mm)
        assert isinstance(sv, containers.SVDModes) and sv.vis.shape == (n_m, ndofmax)
        np.testing.assert_array_equal(sv.nmode[:], g[c + "svd_nmode"])
        np.testing.assert_allclose(sv.vis[:], g[c + "svd_vis"], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(sv.weight[:], g[c + "svd_weight"])  # medians: exact
        # backward from the reference's SVD modes with fresh weights
        sin = containers.SVDModes(mode=ndofmax, mmax=n_m - 1)
        sin.vis[:] = g[c + "svd_vis"]
        sin.weight[:] = g[c + "back_in_weight"]
        sin.nmode[:] = g[c + "svd_nmode"]
        tb = SVDModeProject(mode="backward")
        tb.setup(bt)
        back = tb.process(sin)
        assert isinstance(back, containers.MModes) and back.vis.shape == (n_m, 2, nfreq, npairs)
        np.testing.assert_allclose(back.vis[:], g[c + "back_vis"], rtol=0, atol=1e-11)
        np.testing.assert_array_equal(back.weight[:], g[c + "back_weight"])
        np.testing.assert_array_equal(sin.nmode[:], g[c + "back_in_nmode_after"])
        np.testing.assert_array_equal(back.index_map["freq"]["centre"], g[c + "back_freq_centre"])
        np.testing.assert_array_equal(back.index_map["freq"]["width"], g[c + "back_freq_width"])
        assert len(back.index_map["input"]) == int(g[c + "back_input"])
        tf = SVDModeProject(mode="filter")
        tf.setup(bt)
        filt = tf.process(mm)
        np.testing.assert_allclose(filt.vis[:], g[c + "filter_vis"], rtol=0, atol=1e-10)
        np.testing.assert_array_equal(filt.weight[:], g[c + "filter_weight"])
        # the reference-visible per-m calls go through the same kernel
        tm = g[c + "mvis"][1].transpose(1, 0, 2).reshape(nfreq, 2 * npairs)
        one = bt.project_vector_telescope_to_svd(1, tm)
        np.testing.assert_allclose(one, g[c + "svd_vis"][1, : int(g[c + "svd_nmode"][1])], rtol=0, atol=1e-12)
        np.testing.assert_allclose(bt.project_vector_svd_to_telescope(1, g[c + "svd_vis"][1]).transpose(1, 0, 2), g[c + "back_vis"][1], rtol=0, atol=1e-11)
        for name, thr in (("none", None), ("thr", 4.0)):
            k = KLModeProject(mode="forward", klname="kl_a")
            k.threshold = thr
            k.setup(pm)
            km = k.process(sv)
            assert isinstance(km, containers.KLModes)
            np.testing.assert_array_equal(km.nmode[:], g[c + f"kl_{name}_nmode"])
            np.testing.assert_allclose(km.vis[:], g[c + f"kl_{name}_vis"], rtol=0, atol=1e-11)
            np.testing.assert_array_equal(km.weight[:], g[c + f"kl_{name}_weight"])
            kb = KLModeProject(mode="backward", klname="kl_a")
            kb.threshold = thr
            kb.setup(pm)
            sb = kb.process(km)
            np.testing.assert_array_equal(sb.nmode[:], g[c + f"klback_{name}_nmode"])
            np.testing.assert_allclose(sb.vis[:], g[c + f"klback_{name}_vis"], rtol=0, atol=1e-10)
            np.testing.assert_allclose(pm.kltransforms["kl_a"].project_vector_svd_to_kl(2, g[c + "svd_vis"][2, : int(g[c + "svd_nmode"][2])], thr),
                                       g[c + f"kl_{name}_vis"][2, : int(g[c + f"kl_{name}_nmode"][2])], rtol=0, atol=1e-11)
        bad = KLModeProject(mode="backward", klname="missing")
        bad.setup(pm)
        with pytest.raises(RuntimeError, match="Requested KL basis missing not available"):
            bad.process(sv)


def test_gemv_batch_and_row_median_against_numpy():
    import torch

    from draco_amd import _lib
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    rng = np.random.default_rng(5)
    shapes = [(1, 1), (3, 700), (65, 33), (0, 5), (40, 0), (130, 1500), (7, 64), (33, 65)]
    for dt, npdt, tol in ((_lib.DMM_C128, np.complex128, 1e-13), (_lib.DMM_C64, np.complex64, 1e-6)):
        mats = [(rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(npdt) for s in shapes]
        xs = [rng.standard_normal(s[1]) + 1j * rng.standard_normal(s[1]) for s in shapes]
        a_off = np.concatenate([[0], np.cumsum([m.size for m in mats])[:-1]])
        x_off = np.concatenate([[0], np.cumsum([len(x) for x in xs])[:-1]])
        y_off = np.concatenate([[0], np.cumsum([s[0] for s in shapes])[:-1]]) + 3
        A = ctx.to_device(np.concatenate([m.reshape(-1) for m in mats]), npdt)
        x = ctx.to_device(np.concatenate(xs), np.complex128)
        y = torch.full((sum(s[0] for s in shapes) + 6,), 7.0, dtype=torch.complex128, device=ctx.device)
        desc = _lib.gemv_desc_array(a_off, x_off, y_off, [s[0] for s in shapes], [s[1] for s in shapes])
        _lib.check(_lib.lib.dmm_gemv_batch(ctx.handle, ptr(A), dt, desc, len(shapes), ptr(x), ptr(y)))
        out = y.cpu().numpy()
        assert np.all(out[:3] == 7.0) and np.all(out[-3:] == 7.0)
        for m, xv, o in zip(mats, xs, y_off):
            ref = m.astype(np.complex128) @ xv
            if ref.size == 0:
                continue
            assert np.abs(out[o : o + m.shape[0]] - ref).max() <= tol * max(1.0, np.abs(ref).max() if ref.size else 1.0)
    for n in (1, 2, 5, 6, 1000, 4097):
        v = rng.standard_normal((4, n))
        v[1, : n // 2] = 0.0  # ties and signed zeros
        v[2] = np.round(v[2] * 2) / 2 * (-1.0) ** np.arange(n)
        v[3, ::3] = -0.0
        d = ctx.to_device(v, np.float64)
        o = ctx.empty((4,), np.float64)
        _lib.check(_lib.lib.dmm_row_median(ctx.handle, ptr(d), 4, n, ptr(o)))
        np.testing.assert_array_equal(o.cpu().numpy(), np.median(v, axis=1))

"""Oracle vs. the golden vectors produced by the reference's own functions.

The fixtures under tests/golden/ were written by oracle/gen_golden.py, which executes
the reference source (transform.py / mapmaker.py) in the build container.
"""

import os

import numpy as np
import pytest

from oracle import mapmaker as omm
from oracle import transform as otr


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_make_marray(golden_dir):
    g = _load(golden_dir, "transform_make_marray.npz")
    for i in range(int(g["ncase"])):
        ts, mmax = g[f"c{i}_ts"], int(g[f"c{i}_mmax"])
        dt = np.complex128 if ts.dtype == np.complex128 else None
        out = otr.make_marray(ts, mmax=mmax, dtype=dt)
        ref = g[f"c{i}_out64"]
        assert out.dtype == ref.dtype and out.shape == ref.shape
        # same algorithm, same FFT library, same precision: bit-exact
        np.testing.assert_array_equal(out, ref)
        mm = np.zeros((mmax + 1, 2, *ts.shape[:-1]), np.complex128)
        otr.make_marray(ts, mm)
        np.testing.assert_array_equal(mm, g[f"c{i}_out128"])


def test_make_marray_fill_pattern(golden_dir):
    """Which (m, sign) slots are filled: transform.py:678-679,701-703."""
    ts = np.ones((1, 16), np.complex128) + 1j
    for N, mmax, pos, neg in ((16, 8, 8, 7), (15, 7, 7, 7), (16, 5, 5, 5), (16, 12, 8, 7)):
        rng = np.random.default_rng(N * 100 + mmax)
        ts = rng.standard_normal((2, N)) + 1j * rng.standard_normal((2, N))
        mm = otr.make_marray(ts, mmax=mmax, dtype=np.complex128)
        filled_pos = np.nonzero(np.abs(mm[:, 0]).sum(axis=-1))[0]
        filled_neg = np.nonzero(np.abs(mm[:, 1]).sum(axis=-1))[0]
        assert filled_pos.max() == pos and filled_neg.max() == neg and filled_neg.min() == 1


def test_make_marray_errors():
    ts = np.zeros((2, 3, 8), np.complex64)
    with pytest.raises(ValueError):
        otr.make_marray(ts)
    with pytest.raises(ValueError):
        otr.make_marray(ts, np.zeros((5, 2, 2, 3), np.complex64), mmax=4)
    with pytest.raises(ValueError):
        otr.make_marray(ts, np.zeros((5, 2, 3, 2), np.complex64))


def test_unpack_and_ssarray(golden_dir):
    g = _load(golden_dir, "transform_unpack.npz")
    for i in range(int(g["ncase"])):
        n = int(g[f"c{i}_n"])
        n = None if n < 0 else n
        np.testing.assert_array_equal(otr.unpack_marray(g[f"c{i}_mm"], n=n), g[f"c{i}_unpack"])
        np.testing.assert_array_equal(otr.make_ssarray(g[f"c{i}_mm"], n=n), g[f"c{i}_ss"])


def test_mmode_task(golden_dir):
    g = _load(golden_dir, "transform_mmode_task.npz")
    for i in range(int(g["ncase"])):
        mmax = int(g[f"c{i}_mmax"])
        mv, mw = otr.mmode_transform(
            g[f"c{i}_vis"], g[f"c{i}_weight"], None if mmax < 0 else mmax, bool(g[f"c{i}_window"])
        )
        np.testing.assert_array_equal(mv, g[f"c{i}_mvis"])
        np.testing.assert_array_equal(mw, g[f"c{i}_mweight"])
        assert bool(g[f"c{i}_oddra"]) == bool(g[f"c{i}_vis"].shape[-1] % 2)


def test_hybrid_task(golden_dir):
    g = _load(golden_dir, "transform_hybrid_task.npz")
    for i in range(int(g["ncase"])):
        mmax = int(g[f"c{i}_mmax"])
        mv, mw = otr.mmode_transform(
            g[f"c{i}_vis"], g[f"c{i}_weight"], None if mmax < 0 else mmax, bool(g[f"c{i}_window"]), np.complex64, np.float32
        )
        assert mv.dtype == np.complex64 and mw.dtype == np.float32
        np.testing.assert_allclose(mv, g[f"c{i}_mvis"], rtol=0, atol=1e-7 * np.abs(g[f"c{i}_mvis"]).max())
        np.testing.assert_allclose(mw, g[f"c{i}_mweight"], rtol=1e-6)


def test_inverse_task(golden_dir):
    g = _load(golden_dir, "transform_inverse_task.npz")
    for i in range(int(g["ncase"])):
        nra = int(g[f"c{i}_nra"])
        vis, w = otr.mmode_inverse_transform(
            g[f"c{i}_mvis"], g[f"c{i}_mweight"], bool(g[f"c{i}_oddra"]), None if nra < 0 else nra, bool(g[f"c{i}_window"])
        )
        np.testing.assert_array_equal(vis, g[f"c{i}_vis"])
        np.testing.assert_array_equal(w, g[f"c{i}_weight"])


def test_pinv_svd(golden_dir):
    g = _load(golden_dir, "mapmaker_pinv_svd.npz")
    ranks = []
    for i in range(int(g["ncase"])):
        p = omm.pinv_svd(g[f"c{i}_M"])
        np.testing.assert_allclose(p, g[f"c{i}_pinv"], rtol=0, atol=1e-12 * np.abs(g[f"c{i}_pinv"]).max())
        ranks.append(np.linalg.matrix_rank(p, tol=1e-9 * np.abs(p).max()))
        # the rank the spectrum helper reports is the rank of the reference's own pseudo-inverse
        assert omm.pinv_svd_spectrum(g[f"c{i}_M"])[0] == np.linalg.matrix_rank(g[f"c{i}_pinv"], tol=1e-9 * np.abs(g[f"c{i}_pinv"]).max())
    assert ranks[0] == 7 and ranks[1] == 7 and ranks[2] == 8  # rcond and acond cuts bite


def test_solve_m(golden_dir):
    g = _load(golden_dir, "mapmaker_solve_m.npz")
    n = int(g["ncase"])
    assert n == 18
    for i in range(n):
        bm, m, v, Ni = g[f"c{i}_bm"], int(g[f"c{i}_m"]), g[f"c{i}_v"], g[f"c{i}_Ni"]
        # the one-decomposition helper of the GPU tests IS ml_solve + ml_spectrum
        a1, r1, s1 = omm.ml_solve_with_spectrum(bm, v, Ni)
        r0, s0 = omm.ml_spectrum(bm, Ni)
        np.testing.assert_array_equal(a1, omm.ml_solve(bm, v, Ni))
        assert r1 == r0
        np.testing.assert_allclose(s1, s0, rtol=1e-12, atol=1e-14 * s0.max())
        for name, got in (
            ("dirty", omm.dirty_solve(bm, v, Ni)),
            ("ml", omm.ml_solve(bm, v, Ni)),
            ("wiener", omm.wiener_solve(bm, m, v, Ni)),
            ("wiener_p", omm.wiener_solve(bm, m, v, Ni, 2.5, 1.25)),
        ):
            ref = g[f"c{i}_{name}"]
            assert got.shape == ref.shape
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12 * max(np.abs(ref).max(), 1e-30), err_msg=f"case {i} {name}")
            if name == "ml":  # the SVD leaves O(eps) dust in the structurally-zero l<m slots (so does the reference)
                assert np.abs(got[:, :m]).max(initial=0) < 1e-12 and np.abs(ref[:, :m]).max(initial=0) < 1e-12
            else:
                assert np.all(got[:, :m] == 0) and np.all(ref[:, :m] == 0)


def test_find_keys():
    assert omm.find_keys([400.0, 401.0, 402.0], [402.0, 400.0]) == [2, 0]
    assert omm.find_keys([400.0, 401.0], [399.0]) == [None]
    with pytest.raises(ValueError):
        omm.find_keys([400.0, 401.0], [399.0], require_match=True)


def test_mask_mmode(golden_dir):
    from oracle import flagging as ofl

    g = _load(golden_dir, "flagging_mask_mmode.npz")
    ps = list(zip(g["prod_a"], g["prod_b"]))
    for i in range(int(g["ncase"])):
        auto, mzero, pos, neg, low = (int(x) for x in g[f"c{i}_opts"])
        out = ofl.mask_mmode_weight(g[f"c{i}_w"], ps, bool(auto), bool(mzero), bool(pos), bool(neg), None if low < 0 else low)
        np.testing.assert_array_equal(out, g[f"c{i}_out"])


def _ringmap_case(g, i):
    from oracle import ringmap as orm

    kind = str(g[f"c{i}_kind"])
    skip, oddra = (bool(x) for x in g[f"c{i}_opts"])
    excl = [int(x) for x in g[f"c{i}_exclude"]]
    wtype = str(g[f"c{i}_window"])
    inv_SN, gal_amp, psrc_amp = (float(x) for x in g[f"c{i}_params"])
    hv = g[f"c{i}_hv"]
    window = None
    if wtype != "none":
        window = orm.get_window(g["freq"], np.arange(hv.shape[0]), g["el"], g["ew"], float(g["latitude"]), wtype, exclude_cyl=excl)
    reg = dict(inv_SN=inv_SN) if kind == "tikhonov" else dict(gal_amp=gal_amp, psrc_amp=psrc_amp)
    return dict(kind=kind, hv=hv, hw=g[f"c{i}_hw"], bv=g[f"c{i}_bv"], freq=g["freq"], el=g["el"], ew=g["ew"], oddra=oddra,
                exclude_cyl=excl, skip_deconvolution=skip, window=window, weight_ew=str(g[f"c{i}_weight_ew"]), **reg)


def test_ringmap_deconvolve(golden_dir):
    from oracle import ringmap as orm

    g = _load(golden_dir, "ringmap_deconvolve.npz")
    for i in range(int(g["ncase"])):
        rmm, rmw, rmbp, rmb = orm.deconvolve(**_ringmap_case(g, i))
        for got, name in ((rmm, "map"), (rmw, "wgt"), (rmbp, "dbp"), (rmb, "db")):
            ref = g[f"c{i}_{name}"]
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max(), err_msg=f"case {i} {name}")


def test_ringmap_analytic_beam(golden_dir):
    """Analytic beam m-modes + the ...Analytical makers' outputs (reference ringmapmaker.py:1004-1072, 1189-1190)."""
    from oracle import ringmap as orm

    g = _load(golden_dir, "ringmap_analytic.npz")
    for i in range(int(g["ncase"])):
        hv, el, odd = g[f"c{i}_hv"], g[f"c{i}_el"], int(g[f"c{i}_oddra"])
        bm = orm.analytic_beam_mmodes(g["freq"], g["ew"], el, g["pol"], float(g["latitude"]), hv.shape[0] - 1, odd)
        assert bm.dtype == np.complex64 and np.array_equal(bm, g[f"c{i}_beam_m"]), i
        out = orm.deconvolve(str(g[f"c{i}_kind"]), hv, g[f"c{i}_hw"], bm, g["freq"], el, g["ew"], odd, weight_ew="natural", inv_SN=1e-3)
        for got, name in zip(out, ("map", "wgt", "dbp", "db")):
            ref = g[f"c{i}_{name}"]
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max(), err_msg=f"case {i} {name}")


def _collate_tel(nfeed):
    feedmap = np.abs(np.arange(nfeed)[None, :] - np.arange(nfeed)[:, None])
    feedconj = np.arange(nfeed)[:, None] > np.arange(nfeed)[None, :]
    return feedmap, feedconj


def test_collate_products(golden_dir):
    from oracle import collate as oc

    g = _load(golden_dir, "transform_collate.npz")
    nfeed = int(g["nfeed_tel"])
    feedmap, feedconj = _collate_tel(nfeed)
    for i in range(int(g["ncase"])):
        ids = g[f"c{i}_file_ids"]
        ninp = len(ids)
        prod = np.array([(a, b) for a in range(ninp) for b in range(a, ninp)], dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        v, w, fl = oc.collate(g[f"c{i}_vis"], g[f"c{i}_w"], g[f"c{i}_flags"], ids, g[f"c{i}_ffreq"], prod, 100 + np.arange(nfeed), g["tel_freq"], feedmap, feedconj, str(g[f"c{i}_weight"]))
        np.testing.assert_allclose(v, g[f"c{i}_out_vis"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(w, g[f"c{i}_out_w"], rtol=1e-6)
        np.testing.assert_array_equal(fl, g[f"c{i}_out_flags"])
    # already stacked inputs: representatives re-derived (input 999 absent from the telescope, one pair masked)
    for i in range(int(g["nstacked"])):
        ids = g[f"s{i}_file_ids"]
        ninp = len(ids)
        prod = np.array([(a, b) for a in range(ninp) for b in range(a, ninp)], dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        stack = np.zeros(len(g[f"s{i}_stack_prod"]), dtype=[("prod", "<u4"), ("conjugate", "u1")])
        stack["prod"], stack["conjugate"] = g[f"s{i}_stack_prod"], g[f"s{i}_stack_conj"]
        rev = np.zeros(len(prod), dtype=[("stack", "<u4"), ("conjugate", "u1")])
        rev["stack"], rev["conjugate"] = g[f"s{i}_rev_stack"], g[f"s{i}_rev_conj"]
        feedmask = np.ones((nfeed, nfeed), dtype=bool)
        if int(g[f"s{i}_mask_pair"]):
            feedmask[0, 1] = feedmask[1, 0] = False
        v, w, _ = oc.collate(g[f"s{i}_vis"], g[f"s{i}_w"], g[f"s{i}_flags"], ids, g["tel_freq"], prod, 100 + np.arange(nfeed), g["tel_freq"], feedmap, feedconj,
                             str(g[f"s{i}_weight"]), stack=stack, reverse_stack=rev, feedmask=feedmask)
        assert np.abs(g[f"s{i}_out_vis"]).max() > 0
        np.testing.assert_allclose(v, g[f"s{i}_out_vis"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(w, g[f"s{i}_out_w"], rtol=1e-6)


def test_svd_em_and_tasks(golden_dir):
    """svdfilter.py: svd_em, SVDSpectrumEstimator.process, SVDFilter.process (outputs of the reference code)."""
    from oracle import svdfilter as osvd

    g = _load(golden_dir, "svdfilter.npz")
    for i in range(int(g["nem"])):
        niter, rank = (int(x) for x in g[f"e{i}_opts"])
        u, sig, vh = osvd.svd_em(g[f"e{i}_A"], g[f"e{i}_mask"], niter=niter, rank=rank)
        np.testing.assert_allclose(sig, g[f"e{i}_sig"], rtol=1e-12, atol=1e-13 * g[f"e{i}_sig"][0])
        np.testing.assert_allclose(np.dot(u * sig, vh), g[f"e{i}_recon"], rtol=1e-11, atol=1e-12 * g[f"e{i}_sig"][0])
    for i in range(int(g["ncase"])):
        niter, gthr, lthr = g[f"c{i}_opts"]
        vis, w = g[f"c{i}_vis"], g[f"c{i}_w"]
        spec = osvd.svd_spectrum(vis, w, niter=int(niter))
        np.testing.assert_allclose(spec, g[f"c{i}_spectrum"], rtol=1e-11, atol=1e-12 * spec.max())
        out = osvd.svd_filter(vis, w, niter=int(niter), global_threshold=gthr, local_threshold=lthr)
        np.testing.assert_allclose(out, g[f"c{i}_filtered"], rtol=0, atol=1e-11 * np.abs(vis).max())


def test_expand_products(golden_dir):
    """ExpandProducts.process (synthesis/stream.py:193-246): outputs of the reference class."""
    from oracle import expand as oe

    g = _load(golden_dir, "stream_expand.npz")
    for i in range(int(g["ncase"])):
        nfeed = int(g[f"c{i}_nfeed"])
        feedmap, feedconj = _collate_tel(nfeed)
        feedmap = feedmap.copy()
        if int(g[f"c{i}_mask"]):
            feedmap[1, 3] = feedmap[3, 1] = -1
        v, w = oe.expand_products(g[f"c{i}_vis"], feedmap, feedconj, nfeed)
        np.testing.assert_array_equal(v, g[f"c{i}_out_vis"])
        np.testing.assert_array_equal(w, g[f"c{i}_out_w"])
        assert np.abs(v).max() > 0


def test_simulate_sidereal_process(golden_dir):
    """oracle.stream.simulate_from_alm vs the reference's SimulateSidereal.process run from source
    (stream.py:48-178 under a one-rank MPIArray stand-in; the SHT output is part of the fixture)."""
    from oracle import stream as ostream

    g = _load(golden_dir, "stream_simulate.npz")
    assert str(g["mismatch_message"]) == "Frequencies in map do not match those in Beam Transfers."
    for i in range(int(g["ncase"])):
        nfeed, nfreq, lmax, mmax, npol, npairs = (int(x) for x in g[f"c{i}_dims"][:6])
        beam = g[f"c{i}_beam"]  # [m, f, 2, npairs, npol, lmax+1]
        vis = ostream.simulate_from_alm(g[f"c{i}_alm"], lambda m, f: beam[m, f], lmax, mmax, npairs, npol)
        ref = g[f"c{i}_vis"]
        assert vis.shape == ref.shape == (nfreq, npairs, 2 * mmax + 1) and vis.dtype == ref.dtype == np.complex64
        # same operations in a different association order, then one rounding to complex64
        assert np.abs(vis - ref).max() <= 2e-7 * np.abs(ref).max()
        np.testing.assert_array_equal(g[f"c{i}_weight"], 1.0)
        assert int(g[f"c{i}_ctor_ra"]) == 2 * mmax + 1


def test_mapmaker_process(golden_dir):
    """oracle.mapmaker.solve_alm vs the alm the reference's BaseMapMaker.process hands to the SHT
    (mapmaker.py:35-112 run from source: frequency matching, m trim, pol broadcast, square padding)."""
    g = _load(golden_dir, "mapmaker_process.npz")
    assert str(g["mismatch_message"]) == "Could not find all of the keys."
    for i in range(int(g["ncase"])):
        npairs, lmax, tel_mmax, n_m, npol = (int(x) for x in g[f"c{i}_dims"])
        beam = g[f"c{i}_beam"]  # [m, tel f, 2, npairs, npol, lmax+1]
        freq_ind = omm.find_keys(g[f"c{i}_tel_freq"], g[f"c{i}_freq"], require_match=True)
        for kind in ("dirty", "ml", "wiener"):
            if f"c{i}_{kind}_error" in g.files:
                assert npol != 4 and kind == "wiener" and str(g[f"c{i}_{kind}_error"]) == "ValueError"
                continue
            ref = g[f"c{i}_{kind}"]
            prior = {"prior_amp": 1.0, "prior_tilt": 0.5} if kind == "wiener" else {}
            out = omm.solve_alm(kind, lambda m, f: beam[m, f], g[f"c{i}_mvis"], g[f"c{i}_mweight"], lmax, tel_mmax, freq_ind, npol=npol, **prior)
            assert out.shape == ref.shape == (len(freq_ind), 4, lmax + 1, lmax + 1)
            assert np.abs(out - ref).max() <= 1e-9 * np.abs(ref).max(), (i, kind)
            mm_eff = min(tel_mmax, n_m - 1)
            assert not ref[..., mm_eff + 1 :].any()  # padded m columns stay zero (:108)


def _fg_case(g, i):
    c = f"c{i}_"
    n_m, nfreq, npairs, ndofmax = (int(x) for x in g[c + "dims"])
    ut = lambda m, f: g[c + f"ut_{m}_{f}"]  # noqa: E731
    uinv = lambda m, f: g[c + f"uinv_{m}_{f}"]  # noqa: E731
    return c, n_m, nfreq, npairs, ndofmax, ut, uinv


def test_fgfilter_projections(golden_dir):
    """oracle.fgfilter vs the reference's SVDModeProject / KLModeProject run from source (fgfilter.py:53-239)."""
    from oracle import fgfilter as ofg

    g = _load(golden_dir, "fgfilter.npz")
    for i in range(int(g["ncase"])):
        c, n_m, nfreq, npairs, ndofmax, ut, uinv = _fg_case(g, i)
        vis, weight, nmode = ofg.svd_forward(g[c + "mvis"], g[c + "mweight"], ut, ndofmax)
        np.testing.assert_array_equal(nmode, g[c + "svd_nmode"])
        np.testing.assert_allclose(vis, g[c + "svd_vis"], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(weight, g[c + "svd_weight"])
        mv, mw = ofg.svd_backward(g[c + "svd_vis"], g[c + "back_in_weight"], uinv, lambda m: g[c + "lens"][m], nfreq, npairs)
        np.testing.assert_allclose(mv, g[c + "back_vis"], rtol=0, atol=1e-11)
        np.testing.assert_array_equal(mw, g[c + "back_weight"])
        np.testing.assert_array_equal(g[c + "back_in_nmode_after"], ndofmax)  # the reference overwrites its input's nmode
        for name, thr in (("none", None), ("thr", 4.0)):
            keep = lambda m: (np.arange(len(g[c + f"kl_evals_{m}"])) if thr is None else np.flatnonzero(g[c + f"kl_evals_{m}"] >= thr))  # noqa: E731
            out, wout, nout = ofg.kl_apply(g[c + "svd_vis"], g[c + "svd_weight"], g[c + "svd_nmode"], lambda m: g[c + f"kl_evecs_{m}"][keep(m)], ndofmax)
            np.testing.assert_array_equal(nout, g[c + f"kl_{name}_nmode"])
            np.testing.assert_allclose(out, g[c + f"kl_{name}_vis"], rtol=0, atol=1e-11)
            np.testing.assert_array_equal(wout, g[c + f"kl_{name}_weight"])
            back, _, nb = ofg.kl_apply(g[c + f"kl_{name}_vis"], g[c + f"kl_{name}_weight"], g[c + f"kl_{name}_nmode"], lambda m: g[c + f"kl_inv_{m}"][:, keep(m)], ndofmax)
            np.testing.assert_array_equal(nb, g[c + f"klback_{name}_nmode"])
            np.testing.assert_allclose(back, g[c + f"klback_{name}_vis"], rtol=0, atol=1e-10)


def test_ringmap_chain_oracle_against_the_reference_classes(golden_dir):
    """MakeVisGrid.process, BeamformNS.process and BeamformEW.process (ringmapmaker.py:38-497) run from the reference source
    (tests/golden/ringmap_chain.npz): the oracle's restatements reproduce them -- the grid bit for bit, the beamformed
    streams to float32 rounding."""
    from oracle import ringmap as orm

    g = np.load(os.path.join(golden_dir, "ringmap_chain.npz"))
    fp = g["feedpositions"]
    up = g["uniquepairs"]
    baselines = fp[up[:, 0]] - fp[up[:, 1]]
    prod = [(int(a), int(b)) for a, b in g["prod"]]
    grids = {}
    for centered in (0, 1):
        gv, gw, gr, pol, ew, ns = orm.make_vis_grid(g["vis"], g["weight"], up, g["polarisation"], baselines, g["input_flags"], prod, g["rev_stack"], centered=bool(centered))
        k = f"grid{centered}"
        assert np.array_equal(gv, g[k + "_vis"]) and np.array_equal(gw, g[k + "_weight"]) and np.array_equal(gr, g[k + "_red"])
        assert list(pol) == list(g[k + "_pol"]) and np.allclose(ew, g[k + "_ew"]) and np.allclose(ns, g[k + "_ns"])
        grids[centered] = (gv, gw, gr, pol, ew, ns)
    gv, gw, gr, pol, ew, ns = grids[0]
    hybrids = []
    for i in range(int(g["n_ns"])):
        weight, scaled, auto, sdb, npix, span = g[f"ns{i}_opts"]
        hv, hw, hb, el, nsmax = orm.beamform_ns(gv, gw, gr, ns, g["freq"], npix=int(npix), span=float(span), weight=str(weight), scaled=bool(int(scaled)), include_auto=bool(int(auto)))
        ref = g[f"ns{i}_vis"]
        assert np.abs(hv - ref).max() <= 2e-6 * np.abs(ref).max(), i
        assert np.allclose(hw, g[f"ns{i}_weight"], rtol=2e-6, atol=0), i
        assert np.allclose(el, g[f"ns{i}_el"]) and abs(nsmax - float(g[f"ns{i}_nsmax"])) < 1e-12
        if int(sdb):
            assert np.abs(hb - g[f"ns{i}_db"]).max() <= 2e-6 * np.abs(g[f"ns{i}_db"]).max(), i
        hybrids.append((g[f"ns{i}_vis"], g[f"ns{i}_weight"]))
    assert str(g["ew_db_error"]) == "ValueError"  # the reference's dirty-beam branch of BeamformEW fails on its own broadcast
    for i in range(int(g["n_ew"])):
        hvi, excl, single, wew, flag = g[f"ew{i}_opts"]
        hv, hw = hybrids[int(hvi)]
        fl = None if str(flag) == "" else np.array([c == "1" for c in str(flag)])
        rmm, rmw, rmr, opol, _ = orm.beamform_ew(hv, hw, pol, exclude_intracyl=bool(int(excl)), single_beam=bool(int(single)), weight_ew=str(wew), flag_ew=fl)
        assert list(opol) == list(g[f"ew{i}_pol"])
        ref = g[f"ew{i}_map"]
        assert rmm.shape == ref.shape and np.abs(rmm - ref).max() <= 1e-6 * np.abs(ref).max(), i
        assert np.allclose(rmw, g[f"ew{i}_weight"], rtol=1e-6, atol=0), i
        assert np.allclose(rmr, g[f"ew{i}_rms"], rtol=1e-6, atol=0), i


def test_dirty_solve_many_is_the_per_day_solve():
    """The multi-day form of the Dirty solve (one matrix product per tile) column by column against the per-day
    restatement that the reference's golden `_solve_m` outputs pin."""
    rng = np.random.default_rng(5)
    bm = rng.standard_normal((2, 7, 4, 9)) + 1j * rng.standard_normal((2, 7, 4, 9))
    vs = rng.standard_normal((3, 2, 7)) + 1j * rng.standard_normal((3, 2, 7))
    Nis = rng.uniform(0.0, 2.0, (3, 2, 7))
    many = omm.dirty_solve_many(bm, vs, Nis)
    for d in range(3):
        np.testing.assert_allclose(many[d], omm.dirty_solve(bm, vs[d], Nis[d]), rtol=1e-13, atol=1e-13)

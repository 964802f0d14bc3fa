"""The dense map-makers at cfg-3 order on PHYSICALLY STRUCTURED tiles against the oracle (VERDICT r3 missing 2 / weak 2).

cfg 3 is the ML configuration of BASELINE.json and ``bench.py --maker ml`` times it on ``BeamScreenProvider`` tiles
(every tile eigen-decomposed, none certified) -- the regime where ``pinv_svd``'s cut (``mapmaker.py:287-300``:
``rcond = 1e-3``, ``acond = 1e-4``) really truncates a continuous singular spectrum and where the Gram route of the
GPU path (eigenvalues of ``D B B^H D`` / ``B^H N B``, rank decided on ``sqrt(lambda)``) squares the condition number.
Here one cfg-3 frequency pair (379 baselines, lmax 512, the bench's screen parameters) goes through the BATCHED default
pass of ``MaximumLikelihoodMapMaker`` and sampled (m, f) are compared with ``oracle.mapmaker.ml_solve`` (SVD of
``N^-1/2 B``) on the tiles read back with ``bt.beam_m(m, fi=f)``: the kept rank must be the oracle's and the solution
must agree.  The m whose spectrum comes closest to the cut is found from the library's own rank record
(``dmm_ctx_set_ml_diag``) and is part of the sample; what happens there is written to the report, not hidden.
"""

import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import synth as osyn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RCOND, ACOND = 1e-3, 1e-4  # pinv_svd's defaults, mapmaker.py:287


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _setup(nfreq=2, zero_frac=0.02, seed=11, chan0=0, **screen):
    import torch

    from draco_amd.core import containers
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context

    ctx = Context.get()
    c = osyn.CONFIGS[3]
    tel = TransitTelescope(osyn.frequencies(c["nfreq"])[chan0:chan0 + nfreq], lmax=c["lmax"], ncyl=c["ncyl"], nfeed_cyl=c["nfeed_cyl"])
    assert tel.npairs == 379 and tel.lmax == 512
    bt = BeamScreenProvider(tel, seed=3003, **screen)  # bench.py --maker ml's tile source and seed
    gen = torch.Generator(device=ctx.device).manual_seed(seed)
    shape = (tel.lmax + 1, 2, nfreq, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    # the weights bench.py's day produces: w_m = nra^2 / sum_ra (1 / w), w ~ 20 U(0.5, 1.5) -> ~ 2e4; 2 % exact zeros
    mw = (torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5) * 20.0 * 1024
    mw[torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) < zero_frac] = 0.0
    mm = containers.MModes(mmax=tel.lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
    mm.attach("vis", mv)
    mm.attach("vis_weight", mw)
    per_f = sum(2 * tel.npairs * 4 * (tel.lmax + 1 - m) for m in range(tel.lmax + 1)) * 16
    return ctx, tel, bt, mm, mv, mw, per_f


# (a) the bench's screen parameters: cylinders 22 m apart see the sky up to m ~ 300, so the truncating regime is the
#     telescope-side one (order 758) and every tile beyond is below the absolute cut (rank 0, like the oracle);
# (b) cylinders 44 m apart reach m ~ 500: the SKY-side systems (orders 4 (513 - m) < 758) are in the truncating regime
@pytest.mark.parametrize("name,screen,sample", [
    ("bench", {}, [(0, 0), (0, 40), (0, 180), (0, 240), (0, 280), (0, 300), (0, 324), (1, 7), (1, 262)]),
    ("wide", {"cyl_sep": 44.0}, [(0, 10), (0, 250), (0, 322), (0, 330), (0, 430), (0, 470), (1, 400), (1, 455)]),
    # (c) the TOP of the band (channels 254, 255: 797-798 MHz): the telescope reaches m ~ 600 > lmax -- no tile is null, the
    #     Gram matrices have twice the numerical rank they have at 400 MHz, and the sky-side systems (m > 323) truncate too
    ("top", {"chan0": 254}, [(0, 0), (0, 60), (0, 250), (0, 322), (0, 330), (0, 420), (1, 100), (1, 380)]),
])
def test_cfg3_ml_batched_pass_on_structured_tiles_against_the_oracle_svd(name, screen, sample):
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.device import ptr

    nfreq = 2
    screen = dict(screen)
    chan0 = screen.pop("chan0", 0)
    ctx, tel, bt, mm, mv, mw, per_f = _setup(nfreq, chan0=chan0, **screen)
    lmax, n_m = tel.lmax, tel.lmax + 1
    task = MaximumLikelihoodMapMaker(nside=64, pool_bytes=nfreq * per_f + (1 << 20))
    task.setup(bt)

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    diag = torch.full((nfreq, n_m, 4), -1.0, dtype=torch.float64, device=ctx.device)
    e0, d0, z0 = counter(b"ml_tiles_eigen"), counter(b"ml_tiles_direct"), counter(b"ml_tiles_null")
    st0, sc0 = counter(b"ml_tiles_stopped"), counter(b"ml_stop_cols")
    _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, ptr(diag)))
    try:
        alm = task.make_alm(mm)  # the batched default pass (certificate probe, deferred eigen pass, two-stage reduction with its rank stop)
        ctx.sync()
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, None))
    n_stop, stop_cols = counter(b"ml_tiles_stopped") - st0, counter(b"ml_stop_cols") - sc0
    n_eig, n_dir, n_null = counter(b"ml_tiles_eigen") - e0, counter(b"ml_tiles_direct") - d0, counter(b"ml_tiles_null") - z0
    assert n_eig + n_dir + n_null == nfreq * n_m
    # the null certificate is a shortcut, never a different answer: switched off, the same tiles are decomposed to the same zeros
    _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_null", 1))
    try:
        e1 = counter(b"ml_tiles_eigen")
        alm_off = task.make_alm(mm)
        ctx.sync()
        assert counter(b"ml_tiles_eigen") - e1 == n_eig + n_null
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_null", 0))
    # (not bit for bit everywhere: with fewer tiles the sky-side lists pair up differently and a tile may be reduced at
    # another padded order; the tiles the certificate answered are exact zeros either way)
    alm = alm.cpu().numpy()
    alm_off = alm_off.cpu().numpy()
    diag = diag.cpu().numpy()
    assert np.abs(alm - alm_off).max() < 1e-9 * np.abs(alm).max()
    answered = diag[..., 0] < 0
    assert answered.sum() == n_null + n_dir
    if n_dir == 0:
        for f, m in zip(*np.nonzero(answered)):
            assert not np.any(alm[f, :, m, :]) and not np.any(alm_off[f, :, m, :]), (f, m)
    del alm_off
    assert np.all(np.isfinite(alm))
    dec = diag[..., 0] >= 0  # tiles the eigen path decomposed (the certified ones leave -1)
    assert dec.sum() == n_eig
    # these tiles are ill-conditioned the way real products are: (almost) nothing passes the full-rank certificate; what
    # is not decomposed was answered by the NULL certificate (every singular value at or below acond -> a = 0)
    assert n_dir < 0.1 * nfreq * n_m, (n_eig, n_dir, n_null)

    # distance of every decomposed tile's spectrum from the cut, from the library's own record
    smax, kept_min, cut_max = diag[..., 1], diag[..., 2], diag[..., 3]
    cut = np.maximum(RCOND * smax, ACOND)
    gap = np.where(dec, np.minimum(np.where(kept_min < 1e299, kept_min / cut - 1.0, np.inf), 1.0 - cut_max / cut), np.inf)
    truncated = dec & (cut_max > 1e-2 * cut)  # a real mode is cut, not the rounding dust of zero modes (~1e-5 of the cut)
    assert truncated.sum() > 0.5 * dec.sum()  # the cut removes modes on most tiles (random tiles: on none)
    f_near, m_near = np.unravel_index(np.argmin(gap), gap.shape)
    sample = sorted(set(sample) | {(int(f_near), int(m_near))})

    mv_h, mw_h = mv.cpu().numpy(), mw.cpu().numpy()
    rows, worst, rank_mismatch = [], 0.0, []
    for f, m in sample:
        bm = bt.beam_m(m, fi=f)
        v, Ni = mv_h[m, :, f], mw_h[m, :, f]
        ref, rank_o, sig = omm.ml_solve_with_spectrum(bm, v, Ni)  # (one SVD for both)
        cut_o = max(RCOND * sig[0], ACOND)
        gap_o = float(min(sig[rank_o - 1] / cut_o - 1.0 if rank_o > 0 else np.inf, 1.0 - (sig[rank_o] / cut_o if rank_o < len(sig) else 0.0)))
        err = _rel(alm[f, :, m, :], ref)
        rank_g = int(diag[f, m, 0]) if dec[f, m] else None
        if rank_g is None and n_dir == 0:  # not decomposed, not certified full rank: the null certificate spoke
            assert rank_o == 0 and sig[0] <= ACOND and not np.any(alm[f, :, m, :]), (f, m, rank_o, sig[0])
            rank_g = 0
        rows.append({"f": f, "m": m, "order": int(min(2 * tel.npairs, 4 * (n_m - m))), "side": "telescope" if 4 * (n_m - m) >= 2 * tel.npairs else "sky",
                     "rank_oracle": rank_o, "rank_gpu": rank_g, "sigma_max_oracle": float(sig[0]), "sigma_max_gpu": float(diag[f, m, 1]),
                     "zero_weights": int((Ni == 0).sum()), "gap_to_cut_oracle": gap_o, "gap_to_cut_gpu": float(gap[f, m]), "rel_err": err,
                     "nearest_to_cut_of_all_tiles": bool((f, m) == (f_near, m_near))})
        if rank_g is not None and rank_g != rank_o:
            rank_mismatch.append(rows[-1])
        else:
            worst = max(worst, err)
        # l < m stays exactly zero, like the reference's output for a provider that zeroes those columns
        assert not np.any(alm[f, :, m, :m])
    report = {"screen": name, "screen_parameters": bt.model() and {k: v for k, v in bt.model().items() if np.isscalar(v)}, "cyl_sep": bt.cyl_sep, "tiles": nfreq * n_m, "eigen_decomposed": n_eig, "certified": n_dir, "null_certificate": n_null, "rank_stopped": n_stop, "rank_stop_mean_order": stop_cols / max(n_stop, 1), "tiles_the_cut_truncates": int(truncated.sum()),
              "smallest_gap_to_cut_over_all_tiles": float(gap.min()), "worst_rel_err_where_ranks_agree": worst, "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"r04_ml_structured_vs_oracle_{name}.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))
    # the sample really is in the truncating regime and carries zero weights
    assert sum(0 < r["rank_oracle"] < r["order"] for r in rows) >= len(rows) // 2
    assert any(r["side"] == "sky" and 0 < r["rank_oracle"] < r["order"] for r in rows) or name == "bench"
    assert min(abs(r["gap_to_cut_oracle"]) for r in rows) < 1e-2  # a singular value within 1 % of the cut is in the sample
    assert all(r["zero_weights"] >= 5 for r in rows)
    # rank: the Gram route resolves sigma to ~ eps sigma_max^2 / (2 sigma) -> 1e-10 relative at the cut; a tie closer
    # than 1e-8 of the cut may fall either way (and is then a different, equally valid, truncation): anything else must agree
    for r in rank_mismatch:
        assert abs(r["gap_to_cut_oracle"]) < 1e-8, r
    # solution: 1e-7 of its scale
    assert worst < 1e-7, report


def test_cfg3_wiener_batched_pass_on_structured_tiles_against_the_oracle():
    from draco_amd.analysis.mapmaker import WienerMapMaker

    nfreq = 2
    ctx, tel, bt, mm, mv, mw, per_f = _setup(nfreq, seed=12)
    task = WienerMapMaker(nside=64, pool_bytes=nfreq * per_f + (1 << 20))
    task.setup(bt)
    alm = task.make_alm(mm).cpu().numpy()
    mv_h, mw_h = mv.cpu().numpy(), mw.cpu().numpy()
    worst = 0.0
    for f, m in ((0, 0), (0, 150), (0, 323), (0, 324), (0, 500), (1, 60), (1, 322), (1, 400)):
        ref = omm.wiener_solve(bt.beam_m(m, fi=f), m, mv_h[m, :, f], mw_h[m, :, f], task.prior_amp, task.prior_tilt)
        worst = max(worst, _rel(alm[f, :, m, :], ref))
    assert worst < 1e-9, worst


def test_rank_stop_of_the_band_reduction_changes_no_rank_and_no_solution():
    """The rank stop (herm_band.h: trailing trace <= 1e-13 of lambda_max's lower bound) cuts the reduction of a Gram matrix
    off where what is left is below pinv_svd's cut by seven decades (mapmaker.py:296).  On and off must keep the same
    modes on every tile and give the same a_lm; on well-conditioned tiles it never fires and the pass is bit-identical."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import SyntheticProvider
    from draco_amd.device import ptr

    nfreq = 1
    ctx, tel, bt, mm, mv, mw, per_f = _setup(nfreq, seed=13)
    n_m = tel.lmax + 1

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    def run(provider, stop, shortcut=0):
        task = MaximumLikelihoodMapMaker(nside=64, pool_bytes=nfreq * per_f + (1 << 20))
        task.setup(provider)
        diag = torch.full((nfreq, n_m, 4), -1.0, dtype=torch.float64, device=ctx.device)
        s0, c0 = counter(b"ml_tiles_stopped"), counter(b"ml_stop_cols")
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_rank_stop", stop))
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", shortcut))
        _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, ptr(diag)))
        try:
            alm = task.make_alm(mm)
            ctx.sync()
        finally:
            _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, None))
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_rank_stop", 0))
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_shortcut", 0))
        return alm.cpu().numpy(), diag.cpu().numpy(), counter(b"ml_tiles_stopped") - s0, counter(b"ml_stop_cols") - c0

    a_on, d_on, n_on, cols_on = run(bt, 0)
    a_off, d_off, n_off, _ = run(bt, 1)
    a_loose, d_loose, n_loose, cols_loose = run(bt, 10)  # 1e-10: stops a few columns earlier, still far below the cut
    dec = d_off[..., 0] >= 0
    assert n_off == 0 and n_on > 0.9 * dec.sum() and n_loose >= n_on
    assert cols_on / n_on < 0.5 * 2 * tel.npairs and cols_loose / n_loose <= cols_on / n_on  # mean effective order: under half of 758
    assert np.array_equal(d_on[..., 0], d_off[..., 0]) and np.array_equal(d_loose[..., 0], d_off[..., 0])  # the same modes kept on every tile
    scale = np.abs(a_off).max()
    # (each pass is within ~2e-9 of the oracle's SVD on these tiles -- the Gram route's own resolution at the cut, see the test
    # above --, and the two reductions round differently: their distance is of that size, not smaller)
    assert np.abs(a_on - a_off).max() < 1e-8 * scale, np.abs(a_on - a_off).max() / scale
    assert np.abs(a_loose - a_off).max() < 1e-7 * scale, np.abs(a_loose - a_off).max() / scale
    kept = dec & (d_off[..., 0] > 0)
    assert np.abs(d_on[..., 2][kept] / d_off[..., 2][kept] - 1.0).max() < 1e-8  # the smallest kept sigma of every tile
    # well-conditioned tiles: all the stop can cut off are EXACT zeros -- the padding of a sky-side system decomposed at a
    # larger padded order, the rows of zero-weight baselines (2 % here), the empty half of the m = 0 tile --: only the tail
    # of a reduction is saved, and nothing changes
    syn = SyntheticProvider(tel, seed=77)
    s_on, ds_on, ns_on, cols_s = run(syn, 0, shortcut=2)
    s_off, ds_off, ns_off, _ = run(syn, 1, shortcut=2)
    assert ns_off == 0
    assert ns_on == 0 or cols_s / ns_on > 0.8 * (cols_on / n_on) * 3  # (mean effective order: most of the matrix, not a third of it)
    assert np.array_equal(ds_on[..., 0], ds_off[..., 0])
    assert np.abs(s_on - s_off).max() < 1e-11 * np.abs(s_off).max(), np.abs(s_on - s_off).max() / np.abs(s_off).max()


def test_resident_beam_gram_products_give_bit_identical_days():
    """``cache_beam_gram``: the telescope-side Gram matrix of a day is D (B B^H) D with the day's weights in D only
    (mapmaker.py:190-198 whitens B with them); with the products B B^H kept beside the resident B block the second day's
    matrices are formed by the Gram kernel's own scaling -- every a_lm must equal the uncached pass bit for bit, on the day
    that fills the cache and on the days that use it, also after the B block changed hands."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis import _solve
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers

    nfreq = 2
    ctx, tel, bt, mm, mv, mw, per_f = _setup(nfreq, seed=15)

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    gen = torch.Generator(device=ctx.device).manual_seed(99)
    days = [mm]
    for _ in range(2):  # two more days: other visibilities, other weights (other zero-weight baselines too)
        v = torch.randn(mv.shape, dtype=torch.complex128, device=ctx.device, generator=gen)
        w = (torch.rand(mw.shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5) * 20.0 * 1024
        w[torch.rand(mw.shape, dtype=torch.float64, device=ctx.device, generator=gen) < 0.02] = 0.0
        d = containers.MModes(mmax=tel.lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
        d.attach("vis", v)
        d.attach("vis_weight", w)
        days.append(d)
    plain = MaximumLikelihoodMapMaker(nside=64, pool_bytes=nfreq * per_f + (1 << 20))
    plain.setup(bt)
    ref = [plain.make_alm(d).cpu().numpy() for d in days]
    cached = MaximumLikelihoodMapMaker(nside=64, pool_bytes=nfreq * per_f + (1 << 20), cache_beam_gram=True)
    cached.setup(bt)
    c0, e0 = counter(b"ml_gram_cached"), counter(b"ml_tiles_eigen")
    a0 = cached.make_alm(days[0]).cpu().numpy()  # fills the cache
    assert counter(b"ml_gram_cached") == c0
    n_eig = counter(b"ml_tiles_eigen") - e0
    a1 = cached.make_alm(days[1]).cpu().numpy()  # every telescope-side Gram matrix from its resident product
    n_cached = counter(b"ml_gram_cached") - c0
    n_tel = sum(1 for m in range(tel.lmax + 1) if 4 * (tel.lmax + 1 - m) >= 2 * tel.npairs) * nfreq
    assert 0 < n_cached <= n_tel and n_cached > 0.5 * min(n_eig, n_tel)
    assert np.array_equal(a0, ref[0]) and np.array_equal(a1, ref[1])
    # the B block is given back and filled again: the engine starts the cache over (no stale product survives)
    _solve.release_pools()
    c1 = counter(b"ml_gram_cached")
    a2 = cached.make_alm(days[2]).cpu().numpy()
    assert counter(b"ml_gram_cached") == c1
    assert np.array_equal(a2, ref[2])
    a2b = cached.make_alm(days[2]).cpu().numpy()
    assert counter(b"ml_gram_cached") > c1 and np.array_equal(a2b, ref[2])
    # D days from one pass over B (`make_alm_many`): the products computed for the first day of a group serve the others
    _solve.release_pools()
    c2 = counter(b"ml_gram_cached")
    many = [a.cpu().numpy() for a in cached.make_alm_many(days)]
    assert counter(b"ml_gram_cached") > c2
    for a, r in zip(many, ref):
        assert np.array_equal(a, r)
    # ... and with the singular bases instead (`cache_beam_basis`): the solver's own resolution, not bit for bit
    based = MaximumLikelihoodMapMaker(nside=64, pool_bytes=nfreq * per_f + (1 << 20), cache_beam_basis=True)
    based.setup(bt)
    b0 = counter(b"ml_tiles_basis")
    many_b = [a.cpu().numpy() for a in based.make_alm_many(days)]
    assert counter(b"ml_tiles_basis") > b0 and based._engine.basis_builds == 1
    for a, r in zip(many_b, ref):
        assert np.abs(a - r).max() < 2e-8 * np.abs(r).max()


def test_resident_beam_gram_products_for_wiener_too():
    """The Wiener maker's telescope-side systems ``I + D (B S B^H) D`` (mapmaker.py:267-272) from resident products
    ``B S B^H``: bit-identical a_lm on the filling day and after; another prior starts the cache over."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import WienerMapMaker
    from draco_amd.core import containers

    nfreq = 2
    ctx, tel, bt, mm, mv, mw, per_f = _setup(nfreq, seed=16)

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    gen = torch.Generator(device=ctx.device).manual_seed(7)
    v = torch.randn(mv.shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    w = (torch.rand(mw.shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5) * 20.0 * 1024
    w[torch.rand(mw.shape, dtype=torch.float64, device=ctx.device, generator=gen) < 0.02] = 0.0
    day2 = containers.MModes(mmax=tel.lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
    day2.attach("vis", v)
    day2.attach("vis_weight", w)
    pool = nfreq * per_f + (1 << 20)
    plain = WienerMapMaker(nside=64, pool_bytes=pool)
    plain.setup(bt)
    ref = [plain.make_alm(d).cpu().numpy() for d in (mm, day2)]
    cached = WienerMapMaker(nside=64, pool_bytes=pool, cache_beam_gram=True)
    cached.setup(bt)
    c0 = counter(b"ml_gram_cached")
    a0 = cached.make_alm(mm).cpu().numpy()
    assert counter(b"ml_gram_cached") == c0
    a1 = cached.make_alm(day2).cpu().numpy()
    n_tel = sum(1 for m in range(tel.lmax + 1) if 4 * (tel.lmax + 1 - m) >= 2 * tel.npairs) * nfreq
    assert counter(b"ml_gram_cached") - c0 == n_tel
    assert np.array_equal(a0, ref[0]) and np.array_equal(a1, ref[1])
    # another prior: other products -- the cache starts over, and the answer is that prior's
    other = WienerMapMaker(nside=64, pool_bytes=pool, prior_tilt=1.0)
    other.setup(bt)
    ref_o = other.make_alm(day2).cpu().numpy()
    cached.prior_tilt = 1.0
    c1 = counter(b"ml_gram_cached")
    a2 = cached.make_alm(day2).cpu().numpy()
    assert counter(b"ml_gram_cached") == c1 and np.array_equal(a2, ref_o)


@pytest.mark.parametrize("chan0", [0, 254])
def test_resident_beam_bases_keep_the_same_modes_and_agree_with_the_oracle_svd(chan0):
    """``cache_beam_basis``: with the singular basis ``B = U Sigma V^H`` of a telescope-side tile resident, the day's
    pseudo-inverse (mapmaker.py:190-201, 287-300) is that of the r x r matrix ``Sigma U^H N^-1 U Sigma``.  Against the
    full-order pass on the same structured tiles: the same modes kept on every tile, a_lm within the solver's resolution;
    sampled tiles against the oracle's SVD; a second day (other weights) through the same resident bases."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.device import ptr

    nfreq = 1
    ctx, tel, bt, mm, mv, mw, per_f = _setup(nfreq, seed=17, chan0=chan0)
    n_m = tel.lmax + 1

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    gen = torch.Generator(device=ctx.device).manual_seed(5)
    v2 = torch.randn(mv.shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    w2 = (torch.rand(mw.shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5) * 20.0 * 1024
    w2[torch.rand(mw.shape, dtype=torch.float64, device=ctx.device, generator=gen) < 0.02] = 0.0
    day2 = containers.MModes(mmax=tel.lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
    day2.attach("vis", v2)
    day2.attach("vis_weight", w2)

    def run(task, day):
        diag = torch.full((nfreq, n_m, 4), -1.0, dtype=torch.float64, device=ctx.device)
        _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, ptr(diag)))
        try:
            alm = task.make_alm(day)
            ctx.sync()
        finally:
            _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, None))
        return alm.cpu().numpy(), diag.cpu().numpy()

    pool = nfreq * per_f + (1 << 20)
    plain = MaximumLikelihoodMapMaker(nside=64, pool_bytes=pool)
    plain.setup(bt)
    based = MaximumLikelihoodMapMaker(nside=64, pool_bytes=pool, cache_beam_basis=True)
    based.setup(bt)
    worst = 0.0
    for day, vh, wh in ((mm, mv, mw), (day2, v2, w2)):
        a_ref, d_ref = run(plain, day)
        b0 = counter(b"ml_tiles_basis")
        a_bs, d_bs = run(based, day)
        n_bs = counter(b"ml_tiles_basis") - b0
        dec = d_ref[..., 0] >= 0
        n_tel_dec = int(dec[0, : sum(1 for m in range(n_m) if 4 * (n_m - m) >= 2 * tel.npairs)].sum())
        assert n_bs >= 0.9 * n_tel_dec > 0, (n_bs, n_tel_dec)
        assert np.array_equal(d_bs[..., 0], d_ref[..., 0])  # the same modes kept on every tile
        scale = np.abs(a_ref).max()
        assert np.abs(a_bs - a_ref).max() < 2e-8 * scale, np.abs(a_bs - a_ref).max() / scale
        kept = dec & (d_ref[..., 0] > 0)
        assert np.abs(d_bs[..., 2][kept] / d_ref[..., 2][kept] - 1.0).max() < 1e-8  # the smallest kept sigma of every tile
        vh_h, wh_h = vh.cpu().numpy(), wh.cpu().numpy()
        for m in ((0, 15, 113, 280) if day is mm else (60, 200)):  # (the oracle's SVD of a 758 x 2052 tile takes seconds)
            bm = bt.beam_m(m, fi=0)
            ref, rank_o, _ = omm.ml_solve_with_spectrum(bm, vh_h[m, :, 0], wh_h[m, :, 0])
            assert int(d_bs[0, m, 0]) == rank_o, (m, d_bs[0, m, 0], rank_o)
            worst = max(worst, _rel(a_bs[0, :, m, :], ref))
    assert based._engine.basis_builds == 1  # built once, used by both days
    assert worst < 1e-7, worst
    print("basis route against the oracle's SVD, worst over two days:", worst)


def test_resident_beam_bases_small_telescope_where_the_basis_would_meet_the_reduction_tail():
    """on the basis route ``X = Sigma U^H D`` (nr x ntel) sits in the second half of a matrix's log region
    and the order-nr reduction keeps its T factors and reflector log at that region's tail; for a telescope of padded order
    576 (ntel 513..576) the two would overlap from nr = 448 on -- the default ``basis_rmax``.  A 2 x 24-feed telescope
    (ntel = 566) whose ranks reach ~400 at the top of the band: the chunks whose largest rank lies above 384 must take the
    full-order path (``solve_dense.hip``: the fit check), every tile keeps the modes of the plain pass, a_lm agrees, and the
    tiles of highest rank agree with the oracle's SVD."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    lmax = 512
    tel = TransitTelescope(np.array([798.0]), lmax=lmax, ncyl=2, nfeed_cyl=24)
    assert 513 <= 2 * tel.npairs <= 576
    bt = BeamScreenProvider(tel, seed=3003, feed_sep=1.0, sigma_n=1.2)
    gen = torch.Generator(device=ctx.device).manual_seed(23)
    shape = (lmax + 1, 2, 1, tel.npairs)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = (torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5) * 20.0 * 1024
    mw[torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) < 0.02] = 0.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
    mm.attach("vis", mv)
    mm.attach("vis_weight", mw)
    per_f = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    def run(task):
        diag = torch.full((1, lmax + 1, 4), -1.0, dtype=torch.float64, device=ctx.device)
        _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, ptr(diag)))
        try:
            alm = task.make_alm(mm)
            ctx.sync()
        finally:
            _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, None))
        return alm.cpu().numpy(), diag.cpu().numpy()

    plain = MaximumLikelihoodMapMaker(nside=64, pool_bytes=per_f + (1 << 20))
    plain.setup(bt)
    a_ref, d_ref = run(plain)
    ranks = d_ref[0, :, 0]
    assert ranks.max() > 384, ranks.max()  # the regime the finding is about: a chunk whose basis order would be 448
    based = MaximumLikelihoodMapMaker(nside=64, pool_bytes=per_f + (1 << 20), cache_beam_basis=True)
    based.setup(bt)
    eng = based._get_engine()
    assert eng.basis_rmax == 448
    # A workspace of 1 GiB: the frequency's telescope-side tiles go through a dozen chunks of ~30 matrices, sorted by rank
    # (with the default offer they share ONE chunk, which falls back as a whole -- VERDICT r5 item 7: then `a_bs` below was
    # the full-order path compared with itself).
    offer = eng._offer_workspace
    eng._offer_workspace = lambda option, cap_mib: offer(option, min(cap_mib, 1024))
    run(based)  # (builds the bases)
    b0 = counter(b"ml_tiles_basis")
    a_bs, d_bs = run(based)
    n_bs = counter(b"ml_tiles_basis") - b0
    n_tel = sum(1 for m in range(lmax + 1) if 4 * (lmax + 1 - m) >= 2 * tel.npairs and ranks[m] >= 0)
    # the chunks that hold the high-rank tiles keep the full-order path -- without the fit check they took the basis route and
    # the top ranks came out wrong --, the chunks of lower rank take the basis route: the fit check is tested from both sides
    assert 0 < n_bs < n_tel, (n_bs, n_tel)
    assert np.array_equal(d_bs[..., 0], d_ref[..., 0])
    scale = np.abs(a_ref).max()
    assert np.abs(a_bs - a_ref).max() < 2e-8 * scale, np.abs(a_bs - a_ref).max() / scale
    vh, wh = mv.cpu().numpy(), mw.cpu().numpy()
    top = [int(i) for i in np.argsort(ranks)[-2:]] + [int(np.argmin(np.abs(ranks - 380)))]
    for m in top:
        ref, rank_o, _ = omm.ml_solve_with_spectrum(bt.beam_m(m, fi=0), vh[m, :, 0], wh[m, :, 0])
        assert int(d_bs[0, m, 0]) == rank_o, (m, d_bs[0, m, 0], rank_o)
        assert _rel(a_bs[0, :, m, :], ref) < 1e-7, (m, _rel(a_bs[0, :, m, :], ref))


def test_resident_beam_bases_weight_guard_is_taken_per_day():
    """the weight-range guard of the resident bases (``SolveEngine.basis_max_weight_ratio``) must be
    evaluated for every day.  Day 2's weights live at the same address with the same version counter as day 1's (what
    the caching allocator normally produces for a new day's tensor; staged here by writing through ``.data``, which does
    not bump the counter), but the non-zero ones span eight decades: the day must take the full-order path
    (``ml_tiles_basis`` does not grow) and agree with the plain maker."""
    import ctypes as C

    import torch

    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context

    ctx = Context.get()
    lmax = 256
    tel = TransitTelescope(np.array([450.0]), lmax=lmax, ncyl=2, nfeed_cyl=16)  # (order 384: room for a basis of order <= 256 beside it)
    bt = BeamScreenProvider(tel, seed=3003, feed_sep=1.0, sigma_n=1.2)
    shape = (lmax + 1, 2, 1, tel.npairs)
    gen = torch.Generator(device=ctx.device).manual_seed(29)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    per_f = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16

    g = torch.Generator(device=ctx.device).manual_seed(31)
    w = (torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=g) + 0.5) * 2e4

    def counter(name):
        v = C.c_int64()
        _lib.check(_lib.lib.dmm_ctx_get_counter(ctx.handle, name, C.byref(v)))
        return int(v.value)

    def day(task, w):
        mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
        mm.attach("vis", mv)
        mm.attach("vis_weight", w)
        out = task.make_alm(mm).cpu().numpy()
        ctx.sync()
        return out

    based = MaximumLikelihoodMapMaker(nside=64, pool_bytes=per_f + (1 << 20), cache_beam_basis=True)
    based.setup(bt)
    key1 = (w.data_ptr(), w._version)
    day(based, w)  # builds the bases
    b0 = counter(b"ml_tiles_basis")
    day(based, w)
    assert counter(b"ml_tiles_basis") > b0  # an ordinary day takes the basis route
    w.data[..., :3] *= 1e-8
    assert (w.data_ptr(), w._version) == key1
    b1 = counter(b"ml_tiles_basis")
    a2 = day(based, w)
    assert counter(b"ml_tiles_basis") == b1, "a day whose weights span 1e8 kept the truncated-basis route"
    plain = MaximumLikelihoodMapMaker(nside=64, pool_bytes=per_f + (1 << 20))
    plain.setup(bt)
    a_ref = day(plain, w)
    assert np.abs(a2 - a_ref).max() <= 1e-9 * np.abs(a_ref).max()


def test_stage1_forms_keep_the_same_modes_on_structured_tiles():
    """Round 6: the three forms of stage 1 of the band reduction -- every other two-sided update deferred (default), the same
    with the reading sweeps as one block per matrix ("ml_reduce" = 3), no update deferred ("ml_reduce" = 2: rounds 3-5) --
    on structured, cut-truncated tiles where the rank stop engages (the running diagonal of the deferred forms against the
    stored one of the undeferred): the same modes kept on every tile, a_lm within the solver's own resolution, and the
    highest-rank tile against the oracle's SVD."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import MaximumLikelihoodMapMaker
    from draco_amd.core import containers
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context, ptr

    ctx = Context.get()
    lmax = 256
    tel = TransitTelescope(np.array([600.0]), lmax=lmax, ncyl=2, nfeed_cyl=16)
    bt = BeamScreenProvider(tel, seed=3003, feed_sep=1.0, sigma_n=1.2)
    shape = (lmax + 1, 2, 1, tel.npairs)
    gen = torch.Generator(device=ctx.device).manual_seed(37)
    mv = torch.randn(shape, dtype=torch.complex128, device=ctx.device, generator=gen)
    mw = (torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) + 0.5) * 2e4
    mw[torch.rand(shape, dtype=torch.float64, device=ctx.device, generator=gen) < 0.02] = 0.0
    mm = containers.MModes(mmax=lmax, freq=tel.frequencies, stack=tel.npairs, allocate=False)
    mm.attach("vis", mv)
    mm.attach("vis_weight", mw)
    per_f = sum(2 * tel.npairs * 4 * (lmax + 1 - m) for m in range(lmax + 1)) * 16
    task = MaximumLikelihoodMapMaker(nside=64, pool_bytes=per_f + (1 << 20))
    task.setup(bt)
    out, ranks = {}, {}
    try:
        for red in (0, 3, 5, 2):
            _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_reduce", red))
            diag = torch.full((1, lmax + 1, 4), -1.0, dtype=torch.float64, device=ctx.device)
            _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, ptr(diag)))
            try:
                out[red] = task.make_alm(mm).cpu().numpy()
                ctx.sync()
            finally:
                _lib.check(_lib.lib.dmm_ctx_set_ml_diag(ctx.handle, None))
            ranks[red] = diag.cpu().numpy()[0, :, 0]
    finally:
        _lib.check(_lib.lib.dmm_ctx_set_option(ctx.handle, b"ml_reduce", 0))
    assert ranks[0].max() > 32  # (tiles that are really decomposed and truncated)
    scale = np.abs(out[2]).max()
    for red in (0, 3, 5):
        assert np.array_equal(ranks[red], ranks[2]), red
        assert np.abs(out[red] - out[2]).max() < 2e-8 * scale, (red, np.abs(out[red] - out[2]).max() / scale)
    m = int(np.argmax(ranks[0]))
    ref, rank_o, _ = omm.ml_solve_with_spectrum(bt.beam_m(m, fi=0), mv.cpu().numpy()[m, :, 0], mw.cpu().numpy()[m, :, 0])
    assert int(ranks[0][m]) == rank_o
    assert _rel(out[0][0, :, m, :], ref) < 1e-7

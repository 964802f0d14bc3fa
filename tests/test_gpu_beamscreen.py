"""Physically structured synthetic beam transfers (`BeamScreenProvider`, csrc/beamscreen.hip).

driftscan's beam transfers are third-party inputs to the path; this generator is the library's stand-in
with the same STRUCTURE: tiles = spherical-harmonic analysis of per-pair response maps built from per-polarisation
Jones screens.  Checked here: the tiles against the oracle's NumPy twin (oracle/synth.py::screen_tile, the oracle's own
SHT), the property the construction is for -- the full feed x feed matrix of a simulated stream is positive semi-definite
at every RA for a physical sky, so `SampleNoise` can draw from it (noise.py:311-374) -- and the BASELINE config-5 chain
with the Wishart step in it at real tile size.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapmaker as omm
from oracle import synth as osyn


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _small(npol=4, nfreq=2, lmax=12, nside=8):
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope

    tel = TransitTelescope(np.array([450.0, 700.0])[:nfreq], lmax=lmax, ncyl=2, nfeed_cyl=3, num_pol_sky=npol)
    # (a wide east-west beam: the 8-pixel-per-ring grid of this test could not resolve the default one)
    return tel, BeamScreenProvider(tel, seed=77, nside=nside, sigma_e=0.35, cyl_sep=2.0, feed_sep=0.4)


@pytest.mark.parametrize("npol", [4, 1])
def test_tiles_equal_the_oracle_twin(npol):
    tel, bt = _small(npol)
    model = bt.model()
    for f in range(tel.nfreq):
        for m in (0, 1, 5, 12):
            ref = osyn.screen_tile(model, tel.frequencies[f], m, tel.lmax, npol)
            got = bt.beam_m(m, fi=f)
            assert got.shape == (2, tel.npairs, npol, tel.lmax + 1)
            assert np.all(got[..., :m] == 0) and (m > 0 or np.all(got[1] == 0))
            assert _rel(got, ref) < 1e-10, (f, m, _rel(got, ref))


def test_pool_fill_layouts_and_storage_types_agree_with_beam_m():
    """`fill_pool` (what the engine calls) in both layouts and both storage types, chunked over pairs, against `beam_m`."""
    import torch

    from draco_amd import _lib
    from draco_amd.device import Context

    tel, bt = _small()
    bt.chunk_bytes = 1  # one pair per chunk: every chunk boundary is exercised
    ctx = Context.get()
    ms = np.array([0, 3, 12, 7], dtype=np.int32)
    fs = np.array([1, 0, 1, 1], dtype=np.int32)
    for layout in (_lib.DMM_B_PACKED, _lib.DMM_B_FULL):
        sizes = np.array([bt.tile_elems(int(m), layout) for m in ms], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        for dt, tdt, tol in ((_lib.DMM_C128, torch.complex128, 1e-14), (_lib.DMM_C64, torch.complex64, 1e-6)):
            pool = torch.zeros(int(sizes.sum()), dtype=tdt, device=ctx.device)
            bt.fill_pool(ctx, pool, _lib.tile_array(ms, fs, offs), dt, layout)
            host = pool.cpu().numpy()
            for m, f, o, n in zip(ms, fs, offs, sizes):
                ref = bt.beam_m(int(m), fi=int(f)).reshape(2 * tel.npairs, 4, tel.lmax + 1)
                if layout == _lib.DMM_B_PACKED:
                    ref = ref[..., m:]
                assert _rel(host[o : o + n].reshape(ref.shape), ref) < tol


def test_simulated_stream_is_positive_semi_definite_feed_by_feed():
    """The property the construction is for: Map -> SimulateSidereal -> ExpandProducts gives, at every (frequency, RA),
    a Hermitian feed x feed matrix that is positive semi-definite for a physical sky (I >= sqrt(Q^2 + U^2 + V^2))."""
    import torch

    from draco_amd import _lib
    from draco_amd.core import containers
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context, ptr
    from draco_amd.synthesis.stream import ExpandProducts, SimulateSidereal

    ctx = Context.get()
    lmax, nside = 40, 32
    tel = TransitTelescope(np.array([500.0, 640.0]), lmax=lmax, ncyl=2, nfeed_cyl=4)
    bt = BeamScreenProvider(tel, seed=5, nside=nside, sigma_e=0.2, cyl_sep=3.0, feed_sep=0.4)
    gen = torch.Generator(device=ctx.device).manual_seed(11)
    alm = torch.randn((2, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)
    ll = torch.arange(lmax + 1, device=ctx.device)
    alm = alm * ((ll.double() + 1.0) ** -1.5)[None, None, None, :] * (ll[None, :] >= ll[:, None]).to(alm.dtype)[None, None]
    alm[:, :, 0] = alm[:, :, 0].real.to(alm.dtype)
    alm[:, 1:3, :, :2] = 0
    alm[:, 1:] *= 0.05  # weakly polarised
    alm = alm.contiguous()
    sky = ctx.empty((2, 4, 12 * nside * nside), np.float64)
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), 2, 4, lmax, lmax, nside, ptr(sky)))
    sky[:, 0] += 2.0 * float(sky.abs().max())  # a bright positive monopole: the sky is physical at every pixel
    s = sky.cpu().numpy()
    assert np.all(s[:, 0] > np.sqrt((s[:, 1:] ** 2).sum(axis=1)))
    mp = containers.Map(nside=nside, freq=tel.frequencies, allocate=False)
    mp.attach("map", sky)
    sim = SimulateSidereal()
    sim.setup(bt)
    ss = sim.process(mp)
    ex = ExpandProducts()
    ex.setup(tel)
    full = ex.process(ss)
    v = full.vis[:].astype(np.complex128)  # [freq, nprod, ra]
    n = tel.nfeed
    iu = np.triu_indices(n)
    worst = 0.0
    for f in range(2):
        for t in range(0, v.shape[2], 7):
            mat = np.zeros((n, n), dtype=np.complex128)
            mat[iu] = v[f, :, t]
            mat = mat + np.triu(mat, 1).T.conj()
            ev = np.linalg.eigvalsh(mat)
            worst = min(worst, ev[0] / ev[-1])
    # (the stream is stored in complex64 and the input map goes through a 3-iteration map2alm: not exact, but tiny)
    assert worst > -2e-5, worst


def test_noise_free_chain_returns_the_wiener_filtered_sky():
    """Map -> SimulateSidereal -> ExpandProducts -> CollateProducts -> MModeTransform -> WienerMapMaker on a small
    telescope, no noise: the m-modes are B a of the input a_lm (tiles read back through `beam_m`), the full-triangle
    round trip is bit exact, and per m the Wiener a_lm are (S^-1 + B^H N B)^-1 B^H N B a_true (mapmaker.py:260-272)."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis.mapmaker import WienerMapMaker
    from draco_amd.analysis.transform import CollateProducts, MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context, ptr
    from draco_amd.synthesis.stream import ExpandProducts, SimulateSidereal

    rel = _rel
    ctx = Context.get()
    lmax, nside = 40, 32
    tel = TransitTelescope(np.array([500.0]), lmax=lmax, ncyl=2, nfeed_cyl=4)
    bt = BeamScreenProvider(tel, seed=5, nside=nside, sigma_e=0.2, cyl_sep=3.0, feed_sep=0.4)
    npairs = tel.npairs
    gen = torch.Generator(device=ctx.device).manual_seed(11)
    alm = torch.randn((1, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)
    ll = torch.arange(lmax + 1, device=ctx.device)
    alm = alm * ((ll.double() + 1.0) ** -1.5)[None, None, None, :] * (ll[None, :] >= ll[:, None]).to(alm.dtype)[None, None]
    alm[:, :, 0] = alm[:, :, 0].real.to(alm.dtype)
    alm[:, 1:3, :, :2] = 0
    alm[:, 1:] *= 0.05
    alm = alm.contiguous()
    sky = ctx.empty((1, 4, 12 * nside * nside), np.float64)
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), 1, 4, lmax, lmax, nside, ptr(sky)))
    mp = containers.Map(nside=nside, freq=tel.frequencies, allocate=False)
    mp.attach("map", sky)
    sim = SimulateSidereal(); sim.setup(bt)
    ss = sim.process(mp)
    tr = MModeTransform(); tr.setup(bt)
    mm = tr.process(ss)
    mv = mm.vis[:]
    a_true = alm.cpu().numpy()[0]
    for m in (0, 1, 5, 20, 40):
        B = bt.beam_m(m, fi=0).reshape(2 * npairs, -1)
        v_ref = (B @ a_true[:, m, :].reshape(-1)).reshape(2, npairs)
        assert rel(mv[m, :, 0], v_ref) < 5e-5, m  # (complex64 stream; the 3-iteration map2alm of the input map)
    ex = ExpandProducts(); ex.setup(tel)
    full = ex.process(ss)
    col = CollateProducts(); col.setup(tel)
    ss2 = col.process(full)
    assert np.array_equal(ss2.vis[:], ss.vis[:])
    mm2 = tr.process(ss2)
    assert np.array_equal(mm2.vis[:], mv)
    mm2.weight[:] = mm2.weight[:] * 1e4
    mw = mm2.weight[:]
    wm = WienerMapMaker(nside=nside, prior_amp=10.0, prior_tilt=0.5); wm.setup(bt)
    a_w = wm.make_alm(mm2).cpu().numpy()[0]
    for m in (1, 5, 20):
        B = bt.beam_m(m, fi=0)[..., m:].reshape(2 * npairs, -1)
        nv = mw[m, :, 0].reshape(-1)
        S = omm.wiener_prior(lmax, m, 10.0, 0.5)
        A = (B.conj().T * nv) @ B
        expect = np.linalg.solve(np.diag(1.0 / S) + A, A @ a_true[:, m, m:].reshape(-1))
        got = a_w[:, m, m:].reshape(-1)
        corr = np.vdot(expect, got).real / np.sqrt(np.vdot(expect, expect).real * np.vdot(got, got).real)
        assert corr > 1 - 1e-6 and abs(np.linalg.norm(got) / np.linalg.norm(expect) - 1) < 1e-4, (m, corr)


def test_cfg5_chain_with_the_wishart_step_at_real_tile_size():
    """BASELINE config 5 with its noise model in the chain (VERDICT r2 missing 2), one frequency at the real sizes
    (256 feeds -> 763 baselines, lmax 1023, nra 2047, nside 512):
    Map -> SimulateSidereal -> ExpandProducts -> ReceiverTemperature -> SampleNoise (complex Wishart per RA, on the
    host like the reference, noise.py:311-374) -> CollateProducts -> MModeTransform -> WienerMapMaker.
    Checked: the Wishart step accepts every RA (positive-definite expectations), the Wiener a_lm satisfy the normal
    equations of mapmaker.py:260-272 on sampled m against tiles read back from the provider, and the map correlates
    with the sky that went in."""
    import torch

    from draco_amd import _lib
    from draco_amd.analysis import _solve
    from draco_amd.analysis.flagging import MaskMModeData
    from draco_amd.analysis.mapmaker import WienerMapMaker
    from draco_amd.analysis.transform import CollateProducts, MModeTransform
    from draco_amd.core import containers
    from draco_amd.core.products import BeamScreenProvider, TransitTelescope
    from draco_amd.device import Context, ptr
    from draco_amd.synthesis.noise import ReceiverTemperature, SampleNoise
    from draco_amd.synthesis.stream import ExpandProducts, SimulateSidereal

    _solve.release_pools()
    ctx = Context.get()
    c = osyn.CONFIGS[5]
    tel = TransitTelescope(osyn.frequencies(1), lmax=c["lmax"], ncyl=c["ncyl"], nfeed_cyl=c["nfeed_cyl"])
    npairs, lmax, nside, nra = tel.npairs, tel.lmax, c["nside"], c["nra"]
    assert (tel.nfeed, npairs, lmax, nra, nside) == (256, 763, 1023, 2047, 512)
    bt = BeamScreenProvider(tel, seed=3005)
    assert bt.nside == 512
    gen = torch.Generator(device=ctx.device).manual_seed(4005)
    alm = torch.randn((1, 4, lmax + 1, lmax + 1), dtype=torch.complex128, device=ctx.device, generator=gen)
    ll = torch.arange(lmax + 1, device=ctx.device)
    alm = alm * ((ll.double() + 1.0) ** -1.0)[None, None, None, :] * (ll[None, :] >= ll[:, None]).to(alm.dtype)[None, None]
    alm[:, :, 0] = alm[:, :, 0].real.to(alm.dtype)
    alm[:, 1:3, :, :2] = 0
    alm[:, 1:] *= 0.05
    alm = alm.contiguous()
    sky = ctx.empty((1, 4, 12 * nside * nside), np.float64)
    _lib.check(_lib.lib.dmm_alm2map(ctx.handle, ptr(alm), 1, 4, lmax, lmax, nside, ptr(sky)))
    sky[:, 0] += 2.0 * float(sky.abs().max())
    sky_scale = 10.0 / float(sky[:, 0].mean())  # ~10 K of sky
    sky *= sky_scale
    mp = containers.Map(nside=nside, freq=tel.frequencies, allocate=False)
    mp.attach("map", sky)

    sim = SimulateSidereal()
    sim.setup(bt)
    ss = sim.process(mp)
    assert ss.vis.shape == (1, npairs, nra)
    clean = ss.vis[:].copy()
    ex = ExpandProducts()
    ex.setup(tel)
    full = ex.process(ss)
    assert full.vis.shape == (1, 256 * 257 // 2, nra)
    ReceiverTemperature(recv_temp=50.0).process(full)
    sn = SampleNoise(seed=12, sample_frac=25.0)  # (25 sidereal days' worth of samples)
    full = sn.process(full)  # raises LinAlgError if any RA's expectation is not positive definite
    assert np.all(np.isfinite(full.vis[:])) and np.all(full.weight[:] > 0)
    col = CollateProducts()
    col.setup(tel)
    ss_n = col.process(full)
    assert ss_n.vis.shape == (1, npairs, nra)
    noisy = ss_n.vis[:]
    autos = np.flatnonzero(tel.uniquepairs[:, 0] == tel.uniquepairs[:, 1])
    cross = np.setdiff1d(np.arange(npairs), autos)
    # the receiver temperature sits on the autos; the cross-correlations scatter about the clean stream
    assert abs(np.mean(noisy[0, autos].real - clean[0, autos].real) - 50.0) < 1.0
    assert np.abs(noisy[0, cross] - clean[0, cross]).std() < 0.1 * 60.0

    tr = MModeTransform()
    tr.setup(bt)
    mm = tr.process(ss_n)
    # as in the reference's real-data pipeline (test/pipe_config.yaml:100-131): the auto-correlations (they carry the
    # receiver temperature) and m = 0 are masked ahead of map making
    mm = MaskMModeData().process(mm)
    mv, mw = mm.vis[:], mm.weight[:]
    # the prior matched to the sky that went in, C_l = (scale / (l + 1))^2: with a loose prior the filter is the ML
    # estimator, and the noise it amplifies in the poorly measured modes of these ill-conditioned tiles swamps the rest
    wm = WienerMapMaker(nside=nside, prior_amp=sky_scale, prior_tilt=2.0)
    wm.setup(bt)
    a_w = wm.make_alm(mm).cpu().numpy()[0]
    assert np.all(np.isfinite(a_w))
    for m in (0, 300, 1023):
        B = bt.beam_m(m, fi=0)[..., m:].reshape(2 * npairs, -1)
        nv, vv = mw[m, :, 0].reshape(-1), mv[m, :, 0].reshape(-1)
        x = a_w[:, m, m:].reshape(-1)
        S = omm.wiener_prior(lmax, m, sky_scale, 2.0)
        lhs = x / S + B.conj().T @ (nv * (B @ x))
        rhs = B.conj().T @ (nv * vv)
        # (structured tiles + weights of ~1e4: the normal matrix is ill-conditioned, unlike a random tile set's; the
        # residual of a backward-stable solve scales with eps * |A| |x|, not with |rhs|)
        assert _rel(lhs, rhs) < 1e-6, (m, _rel(lhs, rhs))
        assert np.all(a_w[:, m, :m] == 0)
    out = wm.process(mm)
    rec = out.map[:][0, 0]
    assert out.map.shape == (1, 4, 12 * nside * nside) and np.all(np.isfinite(rec))
    # The estimate against the sky that went in.  A two-cylinder array sees, at a given m, only the declinations where
    # an east-west baseline's fringe rate matches (plus |m| <~ 1 / sigma_e from the single-cylinder baselines): the
    # a_lm themselves are not recoverable mode by mode.  What the chain must deliver is the WIENER-FILTERED true sky:
    # per m,  E[a_hat] = (S^-1 + B^H N B)^-1 B^H N B a_true,  a_true = the m-row of the input a_lm (times the map's scale).
    a_true = alm.cpu().numpy()[0] * sky_scale  # [pol, m, l]
    for m in (40, 150):
        B = bt.beam_m(m, fi=0)[..., m:].reshape(2 * npairs, -1)
        nv = mw[m, :, 0].reshape(-1)
        S = omm.wiener_prior(lmax, m, sky_scale, 2.0)
        A = (B.conj().T * nv) @ B
        expect = np.linalg.solve(np.diag(1.0 / S) + A, A @ a_true[:, m, m:].reshape(-1))
        got = a_w[:, m, m:].reshape(-1)
        nI = lmax + 1 - m  # intensity: the polarised sky is twenty times fainter under the same prior, i.e. noise
        corr_all = np.vdot(expect, got).real / np.sqrt(np.vdot(expect, expect).real * np.vdot(got, got).real)
        expect, got = expect[:nI], got[:nI]
        corr = np.vdot(expect, got).real / np.sqrt(np.vdot(expect, expect).real * np.vdot(got, got).real)
        print(f"cfg5 Wishart chain, m = {m}: correlation of the Wiener a_lm with the Wiener-filtered input sky: {corr:.4f} (all four Stokes: {corr_all:.4f}), |got| / |expect| = {np.linalg.norm(got) / np.linalg.norm(expect):.4f}")
        assert corr > 0.9, (m, corr)
    del out, mp, sky
    torch.cuda.empty_cache()
    _solve.release_pools()

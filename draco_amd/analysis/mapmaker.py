"""Map making from driftscan data using the m-mode formalism, on the GPU.

Drop-in for ``draco/analysis/mapmaker.py``: :class:`BaseMapMaker` (``:11-140``),
:class:`DirtyMapMaker` (``:143-168``), :class:`MaximumLikelihoodMapMaker` (``:171-201``),
:class:`WienerMapMaker` (``:204-284``), :func:`pinv_svd` (``:287-300``), with the same
config attributes (``nside``, ``prior_amp``, ``prior_tilt``) and ``setup(bt)`` /
``process(mmodes) -> Map`` signatures.

Where the reference runs a Python double loop of ``_solve_m`` calls with one HDF5
``beam_m`` read each (``mapmaker.py:79-94``), this runs every (m, freq) solve of a slab in
one launch (``csrc/solve_*.hip``), then the inverse spherical-harmonic transform
(``csrc/sht.hip``) replaces ``hputil.sphtrans_inv_sky`` (``mapmaker.py:112``).  Frequency is
never transposed to m and back (``mapmaker.py:62-67,99``): every stage is per-frequency.
"""

from __future__ import annotations

import collections
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..core import containers, io
from ..core.task import ContainerTask
from ..device import Context, StreamDone, ptr
from ..util import tools
from . import _solve
from .transform import _dev_dataset


# Days queued on each device and not yet known to have finished (oldest first): see `BaseMapMaker.days_in_flight`.
_IN_FLIGHT: dict = {}


def _alm2map_neighbourly(c, alm, nfreq, lmax, mmax, nside, maps):
    """``dmm_alm2map`` of a map-maker: always with the FIRST MFMA form of the Legendre synthesis (option ``sht_synth_form`` = 1).

    The map-makers run this transform beside the HBM-bound solve kernel by default (side stream), where the pipelined
    synthesis kernel of round 5 -- 256 / 512 registers per lane -- starves the solve kernel of wave slots (2000 -> 1076
    m-modes/s on the headline day, ``profiles/r05_cu_split_ab.txt``); the 208-register form does not.  The sequential
    path (``overlap_sht = False``) uses the same form so that a map never depends on that switch: same bits either way.
    An explicit ``sht_variant`` set on the context by the caller (bench.py's DMM_OPTS A/B) is left alone.
    """
    lib = _lib.lib
    pinned = getattr(c, "sht_variant_pin", None)
    if pinned is not None:  # (whatever `sht_variant` the caller has pinned on the context stays as it is)
        _lib.check(lib.dmm_alm2map(c.handle, ptr(alm), nfreq, 4, lmax, mmax, nside, ptr(maps)))
        return
    before = C.c_int64()  # the option's value as the caller left it: restored, not reset
    _lib.check(lib.dmm_ctx_get_counter(c.handle, b"opt_sht_synth_form", C.byref(before)))
    _lib.check(lib.dmm_ctx_set_option(c.handle, b"sht_synth_form", 1))
    try:
        _lib.check(lib.dmm_alm2map(c.handle, ptr(alm), nfreq, 4, lmax, mmax, nside, ptr(maps)))
    finally:
        _lib.check(lib.dmm_ctx_set_option(c.handle, b"sht_synth_form", int(before.value)))


def _bound_run_ahead(ctx, depth):
    """Host-wait until at most ``depth - 1`` earlier days are still running on the device, so that with the day about
    to be queued there are ``depth``."""
    q = _IN_FLIGHT.get(ctx.device_index)
    while q and (len(q) >= depth or q[0].done()):
        d = q.popleft()
        if not d.done():
            d.host_wait()


class BaseMapMaker(ContainerTask):
    """Rudimentary m-mode map maker (``mapmaker.py:11-140``).

    Attributes
    ----------
    nside : int
        Resolution of output Healpix map.
    b_dtype : str
        Storage type of the beam-transfer pool on the GPU: ``"complex128"`` (the reference's
        precision, default) or ``"complex64"`` (half the HBM traffic; accumulation stays
        float64).  Not a reference attribute.
    pool_bytes : int or None
        HBM budget for B tiles per slab (default: 60 % of free memory).
    overlap_sht : bool or None
        Run the inverse SHT of a finished slab on a side stream beside the next slab's solves (True) or on the
        caller's stream between them (False).  None (default): beside them for complex128 B, between them for
        complex64 B -- measured (profiles/r03_c64_overlap_ab.json): with complex128 the solves wait for HBM and the
        SHT's FP64 work fills the gaps (+3.5 % per day); with complex64 the Dirty kernel is issue-bound (conversion +
        FMA per 8 bytes), the SHT's waves take from it exactly what they get (0.69 instead of 0.81 of the HBM peak in
        the step) and the day is 1.3 % slower.  Confining the SHT to a CU subset (16 / 32 / 48 CUs, round 2) lost
        in every setting.  Not a reference attribute.
    days_in_flight : int
        How many ``process`` calls (sidereal days) the host may have queued on the GPU at once.  ``process`` never
        waits for its own day, but before it queues day ``d`` it waits -- on the host -- for day
        ``d - days_in_flight`` to have finished (the GPU still has the days in between queued, so it never idles).
        Without a bound a pipeline that only issues work gets as far ahead as the HBM lets it (a day's launches
        take 2 ms to issue and 250 ms to run); every day in flight holds its own a_lm and maps (10.7 GB at cfg 3),
        and when the caching allocator runs out it synchronises, frees its cache and allocates again while the GPU
        idles (measured: 344 instead of 258 ms per day over 20 days, profiles/r03_runahead_before_20steps.json).  Not a
        reference attribute.
    """

    nside = 256
    b_dtype = "complex128"
    pool_bytes = None
    overlap_sht = None
    days_in_flight = 2
    _config_names = ("nside", "b_dtype", "pool_bytes", "overlap_sht", "days_in_flight")

    bt_cache = None
    _kind = None
    _engine = None

    def setup(self, bt):
        """Set the beamtransfer matrices to use (``mapmaker.py:24-33``)."""
        self.beamtransfer = io.get_beamtransfer(bt)
        self._engine = None

    # ---- device pipeline
    def _get_engine(self):
        if self._engine is None:
            dt = {"complex128": _lib.DMM_C128, "complex64": _lib.DMM_C64}[str(self.b_dtype)]
            self._engine = _solve.SolveEngine(self.beamtransfer, Context.get(), dt, _lib.DMM_B_PACKED, self.pool_bytes,
                                              gram_cache=bool(getattr(self, "cache_beam_gram", False)))
        return self._engine

    def _solve_params(self):
        return {}

    def make_alm(self, mmodes, on_freqs_done=None):
        """All (m, f) solves: device ``alm [nfreq, npol, mmax+1, lmax+1]`` (``mapmaker.py:50-94``)."""
        bt = self.beamtransfer
        mmax = min(bt.telescope.mmax, len(mmodes.index_map["m"]) - 1)
        bt_freq = bt.telescope.frequencies
        mm_freq = mmodes.index_map["freq"]["centre"]
        freq_ind = tools.find_keys(bt_freq, mm_freq, require_match=True)  # ValueError on a miss (:59)

        mmodes.redistribute("freq")
        eng = self._get_engine()
        ctx = eng.ctx
        mvis = _dev_dataset(mmodes.vis, ctx, np.complex128)
        mweight = _dev_dataset(mmodes.weight, ctx, np.float64)
        return eng.solve(self._kind, mvis, mweight, freq_ind, mmax, on_freqs_done=on_freqs_done, **self._solve_params())

    def alm_square(self, alm_d):
        """Device alm -> the reference's square ``[nfreq, 4, lmax+1, lmax+1]`` ndarray (``mapmaker.py:102-109``)."""
        nfreq, npol, n_m, nl = alm_d.shape
        out = np.zeros((nfreq, 4, nl, nl), dtype=np.complex128)
        a = alm_d.permute(0, 1, 3, 2).cpu().numpy()  # [f, pol, l, m]
        out[:, :, :, :n_m] = a  # npol == 1 broadcasts over the 4 slots exactly like mapmaker.py:94
        return out

    def process(self, mmodes):
        """Make a map from the given m-modes (``mapmaker.py:35-118``)."""
        bt = self.beamtransfer
        lmax = bt.telescope.lmax
        user_hook = type(self)._solve_m not in _BUILTIN_SOLVERS
        ctx = Context.get()
        nside = int(self.nside)
        npix = 12 * nside**2
        pending = None
        _bound_run_ahead(ctx, max(int(self.days_in_flight), 1))
        if user_hook or bt.telescope.num_pol_sky != 4:
            alm_d = self._host_loop(mmodes) if user_hook else self.make_alm(mmodes)
            nfreq, npol, n_m, nl = alm_d.shape
            if npol == 1:  # the reference's alm always has 4 pol slots and broadcasts into them (:71,:94)
                alm_d = alm_d.expand(nfreq, 4, n_m, nl).contiguous()
            maps = ctx.empty((nfreq, 4, npix), np.float64)
            _alm2map_neighbourly(ctx, alm_d, nfreq, lmax, n_m - 1, nside, maps)
        else:
            # the inverse SHT (:112) of the frequencies a slab has finished runs on a side stream
            # beside the next slab's fill + solves: it is compute-bound, they are HBM/PCIe-bound
            overlap = (str(self.b_dtype) != "complex64") if self.overlap_sht is None else bool(self.overlap_sht)
            side = Context.side(ctx.device_index) if overlap else ctx
            main = torch.cuda.current_stream(ctx.device)
            out = {}

            def sht_of(alm, f0, f1):
                nfreq, _, n_m, _ = alm.shape  # the local frequencies
                if "maps" not in out:
                    out["maps"] = ctx.empty((nfreq, 4, npix), np.float64)
                if not overlap:
                    _alm2map_neighbourly(ctx, alm[f0:f1], f1 - f0, lmax, n_m - 1, nside, out["maps"][f0:f1])
                    return
                side.wait_for(main)
                side.uses(alm, out["maps"])  # read / written on the side stream: held until side.sync()
                _alm2map_neighbourly(side, alm[f0:f1], f1 - f0, lmax, n_m - 1, nside, out["maps"][f0:f1])

            alm_d = self.make_alm(mmodes, on_freqs_done=sht_of)
            maps = out.get("maps")
            if maps is None:  # no frequencies on this rank
                maps = ctx.empty((alm_d.shape[0], 4, npix), np.float64)
            elif overlap:
                # the last slab's SHT is still running on the side stream.  Whoever reads the map is ordered behind
                # it at that moment (Dataset's `pending`); work that does not -- the next day's transform and
                # solves -- is not held up.  (`record_stream` guards the allocator meanwhile.)
                pending = StreamDone(side.stream, ctx.device)
                side.release_held()
        # the day's last work, for the run-ahead bound of the days that follow
        _IN_FLIGHT.setdefault(ctx.device_index, collections.deque()).append(
            pending if pending is not None else StreamDone(torch.cuda.current_stream(ctx.device), ctx.device))
        m = containers.Map(nside=self.nside, axes_from=mmodes, comm=mmodes.comm, allocate=False)
        m.attach("map", maps, pending=pending)
        return m

    def process_many(self, mmodes_list):
        """Maps of D sidereal days from ONE pass over the beam transfers: ``[process(mm) for mm in mmodes_list]``, with
        every slab of B brought in once for all of them.  Not a reference method: the reference's pipeline calls
        ``process`` once per item (``doc/tutorial.rst:110-120``, the loop at ``mapmaker.py:79-94``) and reads every
        ``beam_m`` again each time; real processing applies one set of beam transfers to many days.

        With B streamed from the host the PCIe crossing (56 GB/s: 28 s per cfg-3 day) is shared by the D days; with B
        resident ``DirtyMapMaker`` also shares every tile READ between up to eight days (``dmm_dirty_run_multi``).
        All days must have the same frequencies and m range.  Each day's a_lm equals its single-day ``process`` bit for
        bit.  HBM: every day in the group holds its own a_lm and maps (10.7 GB at cfg 3).
        """
        mmodes_list = list(mmodes_list)
        if not mmodes_list:
            return []
        bt = self.beamtransfer
        if type(self)._solve_m not in _BUILTIN_SOLVERS or bt.telescope.num_pol_sky != 4:
            return [self.process(mm) for mm in mmodes_list]
        first = mmodes_list[0]
        for mm in mmodes_list[1:]:
            if len(mm.index_map["m"]) != len(first.index_map["m"]) or not np.array_equal(mm.index_map["freq"]["centre"], first.index_map["freq"]["centre"]):
                raise ValueError("process_many: every day must cover the same frequencies and m range")
        lmax = bt.telescope.lmax
        ctx = Context.get()
        nside = int(self.nside)
        npix = 12 * nside**2
        # run-ahead in DAYS, not calls: a group of D days holds D sets of a_lm and maps, so at most
        # days_in_flight // D earlier groups (none for D > days_in_flight / 2) may still be running when this one is queued
        _bound_run_ahead(ctx, max(int(self.days_in_flight) // len(mmodes_list), 1))
        overlap = (str(self.b_dtype) != "complex64") if self.overlap_sht is None else bool(self.overlap_sht)
        side = Context.side(ctx.device_index) if overlap else ctx
        main = torch.cuda.current_stream(ctx.device)
        maps = {}

        ndays = len(mmodes_list)

        def sht_of(d, alm, f0, f1):
            nfreq, _, n_m, _ = alm.shape
            if not maps:  # one allocation for the whole group (see solve_many)
                all_maps = ctx.empty((ndays, nfreq, 4, npix), np.float64)
                for dd in range(ndays):
                    maps[dd] = all_maps[dd]
            if overlap:
                if d == 0:
                    side.wait_for(main)
                side.uses(alm, maps[d])
            _alm2map_neighbourly(side, alm[f0:f1], f1 - f0, lmax, n_m - 1, nside, maps[d][f0:f1])

        alms = self.make_alm_many(mmodes_list, on_freqs_done=sht_of)
        pending = None
        if overlap and maps:
            pending = StreamDone(side.stream, ctx.device)
            side.release_held()
        _IN_FLIGHT.setdefault(ctx.device_index, collections.deque()).append(
            pending if pending is not None else StreamDone(torch.cuda.current_stream(ctx.device), ctx.device))
        out = []
        for d, mm in enumerate(mmodes_list):
            mp = maps.get(d)
            if mp is None:  # no frequencies on this rank
                mp = ctx.empty((alms[d].shape[0], 4, npix), np.float64)
            m = containers.Map(nside=self.nside, axes_from=mm, comm=mm.comm, allocate=False)
            m.attach("map", mp, pending=pending)
            out.append(m)
        return out

    def make_alm_many(self, mmodes_list, on_freqs_done=None):
        """The a_lm of D days from one pass over B (see :meth:`process_many`): a list of device arrays as
        :meth:`make_alm` returns them."""
        bt = self.beamtransfer
        first = mmodes_list[0]
        mmax = min(bt.telescope.mmax, len(first.index_map["m"]) - 1)
        freq_ind = tools.find_keys(bt.telescope.frequencies, first.index_map["freq"]["centre"], require_match=True)
        eng = self._get_engine()
        ctx = eng.ctx
        mv, mw = [], []
        for mm in mmodes_list:
            mm.redistribute("freq")
            mv.append(_dev_dataset(mm.vis, ctx, np.complex128))
            mw.append(_dev_dataset(mm.weight, ctx, np.float64))
        return eng.solve_many(self._kind, mv, mw, freq_ind, mmax, on_freqs_done=on_freqs_done, **self._solve_params())

    def _host_loop(self, mmodes):
        """A subclass overrode ``_solve_m``: honour the hook with the reference's loop (:79-94)."""
        bt = self.beamtransfer
        tel = bt.telescope
        mmax = min(tel.mmax, len(mmodes.index_map["m"]) - 1)
        freq_ind = tools.find_keys(tel.frequencies, mmodes.index_map["freq"]["centre"], require_match=True)
        vis, wgt = mmodes.vis[:], mmodes.weight[:]
        nfreq = vis.shape[2]
        alm = np.zeros((nfreq, tel.num_pol_sky, mmax + 1, tel.lmax + 1), dtype=np.complex128)
        for m in range(mmax + 1):
            for fi in range(nfreq):
                alm[fi, :, m, :] = self._solve_m(m, freq_ind[fi], vis[m, :, fi], wgt[m, :, fi])
        return Context.get().to_device(alm)

    def _solve_m(self, m, f, v, Ni):
        """Solve one (m, f) on the GPU; same contract as ``mapmaker.py:120-140``.

        ``v``, ``Ni``: ``[2, nbase]``; returns ``a [npol, lmax+1]``.
        """
        tel = self.beamtransfer.telescope
        ctx = Context.get()
        v = np.asarray(v, dtype=np.complex128).reshape(2, 1, tel.npairs)
        Ni = np.asarray(Ni, dtype=np.float64).reshape(2, 1, tel.npairs)
        mvis = torch.zeros((m + 1, 2, 1, tel.npairs), dtype=torch.complex128, device=ctx.device)
        mw = torch.zeros((m + 1, 2, 1, tel.npairs), dtype=torch.float64, device=ctx.device)
        mvis[m] = ctx.to_device(v)
        mw[m] = ctx.to_device(Ni)
        dt = {"complex128": _lib.DMM_C128, "complex64": _lib.DMM_C64}[str(self.b_dtype)]
        slab = _solve.Slab(ctx, self.beamtransfer, np.array([m], np.int32), np.array([0], np.int32), np.array([f], np.int32), dt, _lib.DMM_B_PACKED, 1, m + 1)
        eng = _solve.SolveEngine(self.beamtransfer, ctx, dt, _lib.DMM_B_PACKED, cache=True)
        eng._cached_key = ((int(f),), int(m), 1, m + 1, dt, _lib.DMM_B_PACKED)
        eng._cached_slabs = [_OneTile(slab)]
        alm = eng.solve(self._kind, mvis, mw, [f], m, **self._solve_params())
        out = alm[0, :, m, :].cpu().numpy()
        slab.close()
        return out


class _OneTile:
    """Adapter so that a single-tile slab (only m, not 0..m) can be fed through SolveEngine.solve."""

    def __init__(self, slab):
        self.plan, self.pool, self.b_bytes, self.ntile = slab.plan, slab.pool, slab.b_bytes, slab.ntile


class DirtyMapMaker(BaseMapMaker):
    r"""Generate a dirty map: :math:`\hat a = B^\dagger N^{-1} v` (``mapmaker.py:143-168``)."""

    _kind = "dirty"


class MaximumLikelihoodMapMaker(BaseMapMaker):
    r"""Maximum-likelihood map: :math:`\hat a = (N^{-1/2} B)^+ N^{-1/2} v` (``mapmaker.py:171-201``).

    The pseudo-inverse keeps singular values ``> rcond * max`` and ``> acond`` with the
    reference's ``acond=1e-4, rcond=1e-3`` (``mapmaker.py:287,296``).
    """

    _kind = "ml"
    # Multi-day processing: keep the products B B^H of the resident telescope-side tiles beside the B block.  The day's Gram
    # matrix is D (B B^H) D with the day's weights in D only, so from the second day on it is formed by scaling the kept
    # product -- bit-identical to computing it (`dmm_ctx_set_ml_gram_cache`).  Costs 5 MB of HBM per resident tile at cfg 3.
    cache_beam_gram = False
    # One step further: keep the singular bases B = U Sigma V^H of the resident telescope-side tiles (U, Sigma: what driftscan's
    # SVD-compressed products hold).  The day's problem is then the r x r matrix Sigma U^H N^-1 U Sigma, r the beam transfer's
    # numerical rank (150-400 of 758 at cfg 3): no Gram product of B, an eigenproblem of a third of the order.  Same modes
    # kept, a_lm within the solver's own resolution of the full-order pass (not bit-identical: another, equally valid,
    # rounding).  The first pass over a B block builds the bases (one decomposition per tile).
    cache_beam_basis = False
    _config_names = ("cache_beam_gram", "cache_beam_basis")

    def _get_engine(self):
        eng = super()._get_engine()
        eng.basis_cache = bool(self.cache_beam_basis)
        return eng

    def _solve_params(self):
        return {"acond": 1e-4, "rcond": 1e-3}


class WienerMapMaker(BaseMapMaker):
    r"""Wiener-filtered map (``mapmaker.py:204-284``).

    :math:`\hat a = (S^{-1} + B^\dagger N^{-1} B)^{-1} B^\dagger N^{-1} v` with a power-law
    prior ``S = prior_amp**2 * l**(-prior_tilt)`` (``l[0] := 1``).

    The reference's two branches (``ntel > nsky`` at :267, block-inverse form at :275) are
    the same estimator; the kernel factors whichever Hermitian system is smaller.  The
    reference passes ``sym_pos=True`` to ``scipy.linalg.solve`` (removed from SciPy, so the
    reference raises ``TypeError`` as written); the intended Hermitian-PD solve is built.

    Attributes
    ----------
    prior_amp : float
        Amplitude prior, in Kelvin.
    prior_tilt : float
        Power law index prior for the power spectrum.
    """

    prior_amp = 1.0
    prior_tilt = 0.5
    # multi-day processing, as for MaximumLikelihoodMapMaker: the telescope-side system is I + D (B S B^H) D (mapmaker.py:267-272)
    # with the day's weights in D only -- B S B^H is kept beside the resident B block (per prior) and scaled: bit-identical maps
    cache_beam_gram = False
    _config_names = ("prior_amp", "prior_tilt", "cache_beam_gram")
    _kind = "wiener"

    def _solve_params(self):
        # the reference builds the prior for exactly four sky polarisations (`np.concatenate([cl_TT] * 4)`,
        # mapmaker.py:264) and fails on its first product with any other telescope: same error class here
        npol = self.beamtransfer.telescope.num_pol_sky
        if npol != 4:
            raise ValueError(f"operands could not be broadcast together: the Wiener prior covers 4 sky polarisations, the telescope has {npol}")
        return {"prior_amp": self.prior_amp, "prior_tilt": self.prior_tilt}


_BUILTIN_SOLVERS = (BaseMapMaker._solve_m,)


def pinv_svd(M, acond=1e-4, rcond=1e-3):
    """Pseudo-inverse with the reference's rank rule (``mapmaker.py:287-300``), on the GPU.

    Realised through the ML kernel (``dmm_ml_run``): column ``i`` of ``pinv(M)`` is the ML
    solve of the unit vector ``e_i`` with unit weights, all ``nrow`` solves in one batch.
    """
    from ..core.products import ArrayProvider, TransitTelescope

    M = np.asarray(M, dtype=np.complex128)
    nrow, ncol = M.shape
    npairs = (nrow + 1) // 2
    Mp = np.zeros((2 * npairs, ncol), dtype=np.complex128)  # an odd row count is padded with a zero row
    Mp[:nrow] = M
    tel = TransitTelescope(np.arange(float(2 * npairs)), lmax=ncol - 1, mmax=0, num_pol_sky=1, npairs=npairs)
    bt = ArrayProvider(tel, lambda m, f: Mp)
    ctx = Context.get()
    eng = _solve.SolveEngine(bt, ctx, _lib.DMM_C128, _lib.DMM_B_PACKED, cache=False)
    eye = np.eye(2 * npairs, dtype=np.complex128).reshape(2 * npairs, 2, npairs).transpose(1, 0, 2)  # [sign, f, pair]
    mvis = ctx.to_device(eye[np.newaxis], np.complex128)
    mw = torch.ones(mvis.shape, dtype=torch.float64, device=ctx.device)
    alm = eng.solve("ml", mvis, mw, list(range(2 * npairs)), 0, acond=acond, rcond=rcond)  # [f, 1, 1, ncol]
    return alm[:nrow, 0, 0, :].cpu().numpy().T.copy()

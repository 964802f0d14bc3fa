"""Telescope + beam-transfer *provider protocol* (replaces the driftscan objects).

The reference tasks receive a ``drift.core.beamtransfer.BeamTransfer`` (or a
``ProductManager``) [3P, not in this container] and touch exactly these names
(SURVEY.md section 8b):

``bt.telescope.{lmax, mmax, nfreq, frequencies, num_pol_sky, npairs, nfeed, uniquepairs,
input_index, index_map_prod, index_map_stack, reverse_map_stack}``  (``mapmaker.py:50-56``,
``stream.py:68-71,144-162``), ``bt.ntel``, ``bt.nsky`` (``mapmaker.py:160-162``),
``bt.beam_m(m, fi=f)`` (``mapmaker.py:162``) and
``bt.project_vector_sky_to_telescope(m, alm)`` (``stream.py:110``).

Anything exposing those names is a provider.  On top of that this module adds the one
method the reference lacks and a GPU needs -- a *bulk* hand-over of many tiles into an
HBM pool: :meth:`BeamTransferProvider.fill_pool`.

* :class:`SyntheticProvider`  seeded procedural tiles (a counter hash, bit-identical
  on host and device) with the structural ``l < m`` zeros; fills the pool on the GPU.
* :class:`ArrayProvider`      user-supplied tiles (ndarray / memmap / callable).
* Any foreign object with ``beam_m`` (e.g. a real driftscan ``BeamTransfer``) is wrapped by
  :func:`draco_amd.core.io.get_beamtransfer` into :class:`ForeignProvider`.
"""

from __future__ import annotations

import itertools

import numpy as np

from .. import _lib

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = np.asarray(z, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _tile_key(seed, m, f):
    m, f = int(m), int(f)
    with np.errstate(over="ignore"):
        a = _mix64(np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(m + 1))
        return _mix64(a ^ (np.uint64(0xD1B54A32D192ED03) * np.uint64(f + 1)))


def synth_beam_tile(seed, m, f, npairs, npol, lmax):
    """Host twin of ``k_synth_fill`` (``csrc/synth.hip``): complex128 ``[2, npairs, npol, lmax+1]``."""
    m = int(m)
    ntel = 2 * npairs
    scale = np.sqrt(3.0 / (2.0 * ntel))
    key = _tile_key(seed, m, f)
    ctr = np.arange(ntel * npol * (lmax + 1), dtype=np.uint64)
    with np.errstate(over="ignore"):
        h1 = _mix64(key + np.uint64(2) * ctr)
        h2 = _mix64(key + np.uint64(2) * ctr + np.uint64(1))
    re = ((h1 >> np.uint64(11)).astype(np.float64) * 2.0**-53 * 2.0 - 1.0) * scale
    im = ((h2 >> np.uint64(11)).astype(np.float64) * 2.0**-53 * 2.0 - 1.0) * scale
    b = (re + 1j * im).reshape(2, npairs, npol, lmax + 1)
    b[..., :m] = 0.0
    return b


class TransitTelescope:
    """The telescope attributes the path reads (driftscan ``TransitTelescope`` [3P])."""

    cyl_sep = 22.0      # metres between cylinder centres (east-west)
    feed_sep = 0.3048   # metres between feed positions along a cylinder (north-south)

    def __init__(self, frequencies, lmax, mmax=None, num_pol_sky=4, ncyl=1, nfeed_cyl=8, npol_feed=2, npairs=None, pair_rule="canonical"):
        #: how a redundancy class picks its representative pair.  "canonical" (default): the smaller of the pair's key and
        #: its conjugate's -- XY at separation d and YX at -d are one class.  "halfplane": the member whose separation
        #: pos_a - pos_b lies in the half plane x > 0 (x = 0: y > 0; same position: feed order) -- every class then has
        #: x >= 0 and all four polarisation pairs occur, the layout MakeVisGrid (ringmapmaker.py:81-176) expects of
        #: driftscan's `uniquepairs` / `baselines`
        self.pair_rule = str(pair_rule)
        self.frequencies = np.asarray(frequencies, dtype=np.float64)
        self.nfreq = len(self.frequencies)
        self.lmax = int(lmax)
        self.mmax = int(lmax if mmax is None else mmax)
        self.num_pol_sky = int(num_pol_sky)
        self.ncyl, self.nfeed_cyl, self.npol_feed = int(ncyl), int(nfeed_cyl), int(npol_feed)
        self.nfeed = self.ncyl * self.nfeed_cyl * self.npol_feed
        self.input_index = np.array([(i, i) for i in range(self.nfeed)], dtype=[("chan_id", "<u2"), ("correlator_input", "<u2")])
        self._build_pairs()
        if npairs is not None and int(npairs) != self.npairs:
            # free-form pair count (tests): treat every pair as unique, no stacking information
            self.npairs = int(npairs)
            self.uniquepairs = np.stack([np.zeros(self.npairs, int), np.arange(self.npairs)], axis=1)
            self._free = True

    def _build_pairs(self):
        """Redundancy of a regular cylinder grid, autos included."""
        self._free = False
        pos = []
        for c in range(self.ncyl):
            for y in range(self.nfeed_cyl):
                for pl in range(self.npol_feed):
                    pos.append((c, y, pl))
        n = len(pos)
        groups = {}
        prod = []
        for i in range(n):
            for j in range(i, n):
                (ci, yi, pi), (cj, yj, pj) = pos[i], pos[j]
                key = (pi, pj, cj - ci, yj - yi)
                ckey = (pj, pi, ci - cj, yi - yj)
                conj = 0
                if self.pair_rule == "halfplane":
                    # separation of the oriented pair (a, b) = pos_a - pos_b: (i, j) has (ci - cj, yi - yj)
                    sx, sy = ci - cj, yi - yj
                    if sx < 0 or (sx == 0 and sy < 0):
                        key, conj = ckey, 1
                elif ckey < key:
                    key, conj = ckey, 1
                groups.setdefault(key, []).append((len(prod), conj))
                prod.append((i, j))
        self.index_map_prod = np.array(prod, dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        keys = sorted(groups)
        self.npairs = len(keys)
        self.index_map_stack = np.array([groups[k][0] for k in keys], dtype=[("prod", "<u4"), ("conjugate", "u1")])
        rev = np.zeros(len(prod), dtype=[("stack", "<u4"), ("conjugate", "u1")])
        for s, k in enumerate(keys):
            for pidx, conj in groups[k]:
                rev[pidx] = (s, conj)
        self.reverse_map_stack = rev
        # pair -> unique baseline maps in driftscan's shape [3P]: feedmap[i, j] = stack index (-1: not
        # measured), feedconj[i, j] = the pair is the conjugate of the stack's representative
        self.feedmap = np.full((n, n), -1, dtype=np.int64)
        self.feedconj = np.zeros((n, n), dtype=bool)
        for pidx, (i, j) in enumerate(prod):
            s_, c_ = int(rev[pidx]["stack"]), bool(rev[pidx]["conjugate"])
            self.feedmap[i, j], self.feedconj[i, j] = s_, c_
            self.feedmap[j, i], self.feedconj[j, i] = s_, (not c_) if i != j else c_
        self.feedmask = self.feedmap >= 0
        self.nbase = self.npairs
        self.redundancy = np.array([len(groups[k]) for k in keys], dtype=np.float64)
        up = self.index_map_prod[self.index_map_stack["prod"]]
        self.uniquepairs = np.stack([up["input_a"].astype(int), up["input_b"].astype(int)], axis=1)
        # geometry of the regular grid (driftscan's `feedpositions`, `polarisation`, `baselines`, `prodstack` [3P], as
        # MakeVisGrid reads them, ringmapmaker.py:81-106): cylinders `cyl_sep` metres apart east-west, feeds `feed_sep`
        # apart north-south, polarisation X / Y alternating per position
        self.feedpositions = np.array([(c * self.cyl_sep, y * self.feed_sep) for c, y, _ in pos], dtype=np.float64)
        self.polarisation = np.array(["XY"[pl % 2] for _, _, pl in pos])
        # `prodstack`: the pair each stack entry's DATA belongs to -- the representative product with its inputs swapped
        # where the stack entry is the conjugate (containers.py:211-229 applies the same rule to a stream's prodstack);
        # `baselines` follows it, so that polarisation pair, separation and data of a stack entry agree
        ps = up.copy()
        cj = self.index_map_stack["conjugate"].astype(bool)
        ps["input_a"] = np.where(cj, up["input_b"], up["input_a"])
        ps["input_b"] = np.where(cj, up["input_a"], up["input_b"])
        self.prodstack = ps
        self.baselines = self.feedpositions[ps["input_a"].astype(int)] - self.feedpositions[ps["input_b"].astype(int)]


_NPDT = {_lib.DMM_C128: np.complex128, _lib.DMM_C64: np.complex64}


def tile_runs(tiles):
    """Group a ``dmm_tile`` table into runs of consecutive m at one frequency whose pool offsets are back to back.

    Returns a list of ``(f, m_lo, m_hi, b_off)`` (``m_hi`` exclusive).  Slabs are f-major / m-minor, so a slab is a
    few runs: a partial first frequency, whole frequencies, a partial last one.
    """
    rec = np.frombuffer(tiles, dtype=_lib._TILE_DTYPE) if not isinstance(tiles, np.ndarray) else tiles
    runs = []
    n = len(rec)
    if n == 0:
        return runs
    m, f, off = rec["m"].astype(np.int64), rec["f"].astype(np.int64), rec["b_off"]
    brk = np.flatnonzero((np.diff(f) != 0) | (np.diff(m) != 1)) + 1
    starts = np.concatenate([[0], brk])
    stops = np.concatenate([brk, [n]])
    for a, b in zip(starts, stops):
        runs.append((int(f[a]), int(m[a]), int(m[b - 1]) + 1, int(off[a])))
    return runs


class BeamTransferProvider:
    """Base provider: the reference-visible protocol + the bulk hand-over (``beam_block`` / ``fill_pool``)."""

    #: "device": ``fill_pool`` writes the pool with kernels (nothing crosses PCIe); "host": tiles come from host memory
    fill_mode = "host"
    #: tiles of frequency f are those of ``f % alias_period`` (the hbm-pool residency policy of SURVEY 8d); None: all distinct
    alias_period = None
    #: True when ``beam_block`` hands out views of memory the GPU can copy from directly (pinned)
    block_is_pinned = False
    #: worker threads that may call ``beam_block`` / ``beam_m`` at once (None: the host's share; 1: not thread-safe)
    stage_workers = None

    _uids = itertools.count(1)

    def __init__(self, telescope):
        self.telescope = telescope
        self._uid = next(BeamTransferProvider._uids)  # never reused, unlike id(): a stale pool must not match a new provider

    # ---- names the reference tasks read
    @property
    def ntel(self):
        return 2 * self.telescope.npairs

    @property
    def nsky(self):
        return self.telescope.num_pol_sky * (self.telescope.lmax + 1)

    def beam_m(self, m, fi=None):
        """complex128 ``[2, npairs, npol, lmax+1]`` (``fi=None``: leading ``nfreq`` axis)."""
        raise NotImplementedError

    def project_vector_sky_to_telescope(self, mi, vec):
        """``[nfreq, npol, lmax+1] -> [nfreq, ntel]`` (driftscan semantics as used at ``stream.py:110``).

        Served by the device kernel (``dmm_project_run``); kept for protocol completeness.
        """
        from ..analysis import _solve

        return _solve.project_single_m(self, mi, vec)

    # ---- identity of the tile contents (what may stay resident in a pool between passes)
    def content_key(self):
        """Identity of the tile CONTENTS: equal keys mean a pool filled from one serves the other.  Tiles are taken
        to be immutable for the provider's lifetime (``beam_m`` of the same (m, f) always returns the same numbers)."""
        return ("provider", self._uid)

    def canonical_freq(self, f):
        """The frequency index whose tiles frequency ``f`` shares (identity unless the provider aliases)."""
        return f if self.alias_period is None else np.asarray(f) % int(self.alias_period)

    # ---- bulk hand-over
    def tile_elems(self, m, layout):
        tel = self.telescope
        w = tel.lmax + 1 if layout == _lib.DMM_B_FULL else tel.lmax + 1 - m
        return self.ntel * tel.num_pol_sky * w

    def beam_block(self, m_lo, m_hi, f_lo, f_hi, dtype=np.complex128, layout=_lib.DMM_B_PACKED, out=None):
        """Tiles ``(m, f)`` for ``f_lo <= f < f_hi`` (outer), ``m_lo <= m < m_hi`` (inner) back to back in the pool's
        wire format: each tile ``[ntel, npol, lmax+1-m]`` (packed) or ``[ntel, npol, lmax+1]`` (full), C order,
        ``dtype``.  The bulk method SURVEY 8b asks of a provider; the reference has only the per-tile ``beam_m``
        (``mapmaker.py:162``), which this generic version calls once per tile.  Returns a 1-D array (``out`` when
        given); providers that hold their tiles in this format return views instead of copies.
        """
        tel = self.telescope
        npdt = np.dtype(dtype)
        sizes = [self.tile_elems(m, layout) for m in range(m_lo, m_hi)]
        total = sum(sizes) * (f_hi - f_lo)
        if out is None:
            out = np.empty(total, dtype=npdt)
        elif out.size != total or out.dtype != npdt:
            raise ValueError(f"beam_block: out has {out.size} x {out.dtype}, need {total} x {npdt}")
        pos = 0
        for f in range(f_lo, f_hi):
            for m, n in zip(range(m_lo, m_hi), sizes):
                b = np.asarray(self.beam_m(m, fi=f)).reshape(self.ntel, tel.num_pol_sky, tel.lmax + 1)
                if layout == _lib.DMM_B_PACKED:
                    b = b[..., m:]
                np.copyto(out[pos : pos + n].reshape(b.shape), b, casting="same_kind")
                pos += n
        return out

    def fill_pool(self, ctx, pool, tiles, dtype, layout, stager=None):
        """Write the tiles described by ``tiles`` (ctypes ``dmm_tile`` array) into the device ``pool``.

        Enqueues on ``ctx``'s stream and returns once everything is enqueued.  Generic host path: the table is cut
        into chunks of whole tiles; each chunk is one :meth:`beam_block` call, packed into a pinned staging slot by a
        worker thread and copied up asynchronously (``core/hoststage.py``), or copied straight from the provider's
        own memory when that is pinned.  Providers that can do better (procedural, already resident) override this.
        """
        import torch

        from .hoststage import HostStager

        npdt = np.dtype(_NPDT[dtype])
        es = npdt.itemsize
        if stager is None:
            stager = HostStager.get(ctx.device, **({"workers": int(self.stage_workers)} if self.stage_workers else {}))
        stream = ctx.stream if ctx.stream is not None else torch.cuda.current_stream(ctx.device)
        pool_u8 = pool.view(-1).view(torch.uint8)
        per_m = None
        jobs = []
        for f, m_lo, m_hi, b_off in tile_runs(tiles):
            if per_m is None or len(per_m) < m_hi:
                per_m = np.array([self.tile_elems(m, layout) for m in range(max(m_hi, self.telescope.mmax + 1))], dtype=np.int64)
                per_m += per_m & 1  # the slab pads tiles to an even element count (a no-op: ntel is even)
            # cut the run into chunks of whole tiles that fit a slot
            m0 = m_lo
            while m0 < m_hi:
                cum = np.cumsum(per_m[m0:m_hi]) * es
                k = int(np.searchsorted(cum, stager.slot_bytes, side="right"))
                if k == 0:
                    if not self.block_is_pinned:
                        raise MemoryError(f"one tile ({int(cum[0])} bytes) exceeds the staging slot ({stager.slot_bytes})")
                    k = 1
                if self.block_is_pinned:
                    k = m_hi - m0  # no staging: the whole run in one copy
                m1 = m0 + k
                nbytes = int(per_m[m0:m1].sum()) * es
                dst = (b_off + int(per_m[m_lo:m0].sum())) * es
                if self.block_is_pinned:
                    src = self.beam_block(m0, m1, f, f + 1, npdt, layout)
                    jobs.append((dst, nbytes, torch.from_numpy(src.view(np.uint8))))
                else:
                    jobs.append((dst, nbytes, _Producer(self, m0, m1, f, npdt, layout)))
                m0 = m1
        stager.upload(jobs, pool_u8, stream)


class _Producer:
    """Packs one chunk (tiles m0..m1-1 of frequency f) into a staging slot; runs on a worker thread."""

    __slots__ = ("p", "m0", "m1", "f", "dt", "layout")

    def __init__(self, p, m0, m1, f, dt, layout):
        self.p, self.m0, self.m1, self.f, self.dt, self.layout = p, m0, m1, f, dt, layout

    def __call__(self, out_u8):
        out = out_u8.view(self.dt)
        blk = self.p.beam_block(self.m0, self.m1, self.f, self.f + 1, self.dt, self.layout, out=out)
        if blk is not out:  # the provider returned a view of its own (pageable) memory: one GIL-free memcpy
            np.copyto(out, blk, casting="same_kind")


class SyntheticProvider(BeamTransferProvider):
    """Seeded procedural B tiles, ``B ~ U``-complex with variance ``1/ntel``, zeros for ``l < m``."""

    fill_mode = "device"

    def __init__(self, telescope, seed=3000):
        super().__init__(telescope)
        self.seed = int(seed)

    def content_key(self):
        tel = self.telescope
        return ("synthetic", self.seed, tel.npairs, tel.num_pol_sky, tel.lmax)

    def beam_m(self, m, fi=None):
        tel = self.telescope
        if fi is None:
            return np.stack([self.beam_m(m, fi=f) for f in range(tel.nfreq)])
        return synth_beam_tile(self.seed, m, fi, tel.npairs, tel.num_pol_sky, tel.lmax)

    def fill_pool(self, ctx, pool, tiles, dtype, layout, stager=None):
        from ..device import ptr

        tel = self.telescope
        _lib.check(
            _lib.lib.dmm_synth_beam_fill(
                ctx.handle, tiles, len(tiles), tel.npairs, tel.num_pol_sky, tel.lmax, dtype, layout, self.seed, ptr(pool)
            )
        )


class BeamScreenProvider(BeamTransferProvider):
    """Physically structured synthetic beam transfers (``csrc/beamscreen.hip``), generated on the GPU.

    What :class:`SyntheticProvider` cannot give: tiles with the structure of real driftscan products.  A transit
    telescope at ``latitude`` whose feeds sit on the regular cylinder grid of :class:`TransitTelescope` (cylinders
    ``cyl_sep`` metres apart east-west, feeds ``feed_sep`` apart north-south, two polarisations per position), one
    complex Jones screen per polarisation type (gain ripple ``eps_gain``, cross-polar leakage ``eps_leak``; feeds of
    one type share it, so redundant baselines are exactly redundant) and a primary beam that is narrow east-west
    (``sigma_e`` at 600 MHz, scaling with wavelength) and wide north-south.  Tile ``(m, f)`` is the spherical-harmonic
    analysis of the pair's response maps by this library's own ``dmm_map2alm`` (iteration 0, the exact adjoint of
    ``dmm_alm2map``), so that

    * the stream ``SimulateSidereal`` makes from it is, for every RA, ``sum_p w_p e_i(p)^H C(p) e_j(p)`` with ``C`` the
      coherency of the band-limited sky at the pixel centres: after ``ExpandProducts`` the full feed x feed matrix is
      positive semi-definite for any physical sky, which is what ``SampleNoise`` (``noise.py:311-374``) needs and a
      random tile set cannot offer (VERDICT r2 missing 2);
    * the Gram matrices of the ML / Wiener solvers are ill-conditioned the way real ones are: a beam-limited patch
      of sky behind 4 (lmax + 1 - m) columns.

    ``beam_m`` serves single tiles to host code (tests, oracle comparisons) from a per-frequency cache; the engine
    uses ``fill_pool``.
    """

    fill_mode = "device"
    C_LIGHT = 299.792458  # m MHz

    def __init__(self, telescope, seed=5000, nside=None, latitude=49.3, cyl_sep=22.0, feed_sep=0.3048, sigma_e=0.04,
                 sigma_n=0.7, eps_gain=0.05, eps_leak=0.03, chunk_bytes=2 << 30):
        super().__init__(telescope)
        tel = telescope
        if getattr(tel, "_free", False):
            raise ValueError("BeamScreenProvider needs the telescope's grid (a free-form pair count has no baselines)")
        if tel.num_pol_sky not in (1, 4):
            raise ValueError("BeamScreenProvider: num_pol_sky must be 1 or 4")
        self.seed = int(seed)
        # lmax <= 2 nside keeps the quadrature of the analysis well behaved (the property above holds for any nside)
        self.nside = int(nside) if nside else max(8, 1 << int(np.ceil(np.log2(max(tel.lmax, 2) / 2.0))))
        self.latitude = float(latitude)
        self.sigma_e, self.sigma_n = float(sigma_e), float(sigma_n)
        self.eps_gain, self.eps_leak = float(eps_gain), float(eps_leak)
        self.cyl_sep, self.feed_sep = float(cyl_sep), float(feed_sep)
        self.chunk_bytes = int(chunk_bytes)
        # feed k = (cylinder, position, polarisation) in TransitTelescope._build_pairs' order
        k = np.arange(tel.nfeed)
        self.feed_pol = (k % tel.npol_feed).astype(np.int32)
        self.feed_east = (k // (tel.npol_feed * tel.nfeed_cyl)) * self.cyl_sep
        self.feed_north = ((k // tel.npol_feed) % tel.nfeed_cyl) * self.feed_sep
        ia, ib = tel.uniquepairs[:, 0], tel.uniquepairs[:, 1]
        self.sep_e = np.ascontiguousarray(self.feed_east[ib] - self.feed_east[ia], dtype=np.float64)
        self.sep_n = np.ascontiguousarray(self.feed_north[ib] - self.feed_north[ia], dtype=np.float64)
        self.pol_a = np.ascontiguousarray(self.feed_pol[ia] % 2, dtype=np.int32)
        self.pol_b = np.ascontiguousarray(self.feed_pol[ib] % 2, dtype=np.int32)
        self._host_cache: dict = {}

    def model(self):
        """The numbers that define the tiles (for the oracle's twin, ``oracle/synth.py::screen_tile``)."""
        return {"seed": self.seed, "nside": self.nside, "latitude": self.latitude, "sigma_e": self.sigma_e, "sigma_n": self.sigma_n,
                "eps_gain": self.eps_gain, "eps_leak": self.eps_leak, "sep_e": self.sep_e, "sep_n": self.sep_n, "pol_a": self.pol_a,
                "pol_b": self.pol_b}

    def content_key(self):
        tel = self.telescope
        return ("beam-screen", self.seed, self.nside, self.latitude, self.cyl_sep, self.feed_sep, self.sigma_e, self.sigma_n, self.eps_gain,
                self.eps_leak, tel.ncyl, tel.nfeed_cyl, tel.npol_feed, tel.num_pol_sky, tel.lmax, tel.mmax, tel.frequencies.tobytes())

    def wavelength(self, f):
        return self.C_LIGHT / float(self.telescope.frequencies[int(f)])

    def fill_pool(self, ctx, pool, tiles, dtype, layout, stager=None):
        import ctypes as C

        import torch

        from ..device import ptr

        tel = self.telescope
        rec = np.frombuffer(tiles, dtype=_lib._TILE_DTYPE) if not isinstance(tiles, np.ndarray) else tiles
        if len(rec) == 0:
            return
        npol, lmax = tel.num_pol_sky, tel.lmax
        npix = 12 * self.nside**2
        m_top = int(rec["m"].max())
        per_pair = 2 * npol * (npix * 8 + (m_top + 1) * (lmax + 1) * 16)
        nc_max = int(max(1, min(tel.npairs, self.chunk_bytes // per_pair)))
        maps = torch.empty((2, nc_max, npol, npix), dtype=torch.float64, device=ctx.device)
        alm = torch.empty((2, nc_max, npol, m_top + 1, lmax + 1), dtype=torch.complex128, device=ctx.device)
        lib = _lib.lib
        vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        for f in np.unique(rec["f"]):
            sel = rec[rec["f"] == f]
            tl = _lib.tile_array(sel["m"], sel["f"], sel["b_off"])
            sigma_e = self.sigma_e * self.wavelength(f) / (self.C_LIGHT / 600.0)  # the aperture is fixed: width ~ wavelength
            for s0 in range(0, tel.npairs, nc_max):
                nc = min(nc_max, tel.npairs - s0)
                mp = maps.view(-1)[: 2 * nc * npol * npix]
                al = alm.view(-1)[: 2 * nc * npol * (m_top + 1) * (lmax + 1)]
                _lib.check(lib.dmm_beam_screen_maps(ctx.handle, self.nside, npol, self.wavelength(f), np.deg2rad(self.latitude), self.seed, sigma_e,
                                                    self.sigma_n, self.eps_gain, self.eps_leak, vp(self.sep_e[s0:]), vp(self.sep_n[s0:]),
                                                    vp(self.pol_a[s0:]), vp(self.pol_b[s0:]), nc, ptr(mp)))
                _lib.check(lib.dmm_map2alm(ctx.handle, ptr(mp), 2 * nc, npol, lmax, m_top, self.nside, 0, ptr(al)))
                _lib.check(lib.dmm_beam_screen_pack(ctx.handle, ptr(al), nc, s0, tl, len(sel), tel.npairs, npol, lmax, m_top, dtype, layout, ptr(pool)))

    def _freq_tiles(self, f):
        """All tiles of one frequency on the host, full layout ``[mmax+1, 2, npairs, npol, lmax+1]`` (cached)."""
        f = int(f)
        if f not in self._host_cache:
            import torch

            from ..device import Context

            tel = self.telescope
            ctx = Context.get()
            n_m = tel.mmax + 1
            per = self.ntel * tel.num_pol_sky * (tel.lmax + 1)
            pool = torch.empty(n_m * per, dtype=torch.complex128, device=ctx.device)
            tl = _lib.tile_array(np.arange(n_m, dtype=np.int32), np.full(n_m, f, np.int32), np.arange(n_m, dtype=np.int64) * per)
            self.fill_pool(ctx, pool, tl, _lib.DMM_C128, _lib.DMM_B_FULL)
            if len(self._host_cache) >= 4:
                self._host_cache.pop(next(iter(self._host_cache)))
            self._host_cache[f] = pool.cpu().numpy().reshape(n_m, 2, tel.npairs, tel.num_pol_sky, tel.lmax + 1)
        return self._host_cache[f]

    def beam_m(self, m, fi=None):
        tel = self.telescope
        if fi is None:
            return np.stack([self.beam_m(m, fi=f) for f in range(tel.nfreq)])
        per = self.ntel * tel.num_pol_sky * (tel.lmax + 1)
        if (tel.mmax + 1) * per * 16 <= 1 << 30:  # small telescopes: one generation per frequency serves every m
            return self._freq_tiles(fi)[int(m)].copy()
        import torch

        from ..device import Context

        ctx = Context.get()
        pool = torch.empty(per, dtype=torch.complex128, device=ctx.device)
        tl = _lib.tile_array(np.array([m], np.int32), np.array([fi], np.int32), np.zeros(1, np.int64))
        self.fill_pool(ctx, pool, tl, _lib.DMM_C128, _lib.DMM_B_FULL)
        return pool.cpu().numpy().reshape(2, tel.npairs, tel.num_pol_sky, tel.lmax + 1)


class ArrayProvider(BeamTransferProvider):
    """Tiles from memory: ``beams[m][f]`` array-likes or a callable ``(m, f) -> ndarray``."""

    def __init__(self, telescope, beams):
        super().__init__(telescope)
        self._beams = beams

    def beam_m(self, m, fi=None):
        tel = self.telescope
        if fi is None:
            return np.stack([self.beam_m(m, fi=f) for f in range(tel.nfreq)])
        b = self._beams(m, fi) if callable(self._beams) else self._beams[m][fi]
        return np.asarray(b).reshape(2, tel.npairs, tel.num_pol_sky, tel.lmax + 1)


class PackedStoreProvider(BeamTransferProvider):
    """Tiles held in host memory ALREADY in the pool's wire format.

    ``store`` is a flat array (ndarray, ``np.memmap`` or the NumPy view of a pinned torch tensor) holding, for every
    frequency ``f`` of the telescope (outer) and every ``m = 0 .. mmax`` (inner), the packed tile
    ``[ntel, npol, lmax+1-m]`` in C order -- what ``bt.beam_m(m, fi=f)[..., m:]`` flattens to.  ``beam_block`` then is
    a slice: whole runs go to the GPU with no host-side packing; if the memory is pinned (``pinned=True``) the copy
    engine reads it directly, otherwise worker threads memcpy it through the pinned staging ring.
    A complex64 store halves the PCIe bytes (use it with ``b_dtype = "complex64"``).
    """

    def __init__(self, telescope, store, pinned=False, keepalive=None):
        super().__init__(telescope)
        tel = telescope
        self.store = store
        self.block_is_pinned = bool(pinned)
        self._keepalive = keepalive  # e.g. the pinned torch tensor `store` is a view of
        per_m = np.array([self.tile_elems(m, _lib.DMM_B_PACKED) for m in range(tel.mmax + 1)], dtype=np.int64)
        self._m_off = np.concatenate([[0], np.cumsum(per_m)])
        self.per_freq = int(self._m_off[-1])
        if store.ndim != 1 or store.size != self.per_freq * tel.nfreq:
            raise ValueError(f"store must be flat with {self.per_freq * tel.nfreq} elements, got shape {store.shape}")

    @staticmethod
    def elements(telescope):
        """Number of store elements the telescope's tiles take (all frequencies, all m, packed)."""
        tel = telescope
        ntel = 2 * tel.npairs
        return int(sum(ntel * tel.num_pol_sky * (tel.lmax + 1 - m) for m in range(tel.mmax + 1))) * tel.nfreq

    def content_key(self):
        return ("packed-store", self._uid)

    def beam_m(self, m, fi=None):
        tel = self.telescope
        if fi is None:
            return np.stack([self.beam_m(m, fi=f) for f in range(tel.nfreq)])
        a = fi * self.per_freq + self._m_off[m]
        out = np.zeros((self.ntel, tel.num_pol_sky, tel.lmax + 1), dtype=np.complex128)
        out[..., m:] = np.asarray(self.store[a : a + self._m_off[m + 1] - self._m_off[m]]).reshape(self.ntel, tel.num_pol_sky, tel.lmax + 1 - m)
        return out.reshape(2, tel.npairs, tel.num_pol_sky, tel.lmax + 1)

    def beam_block(self, m_lo, m_hi, f_lo, f_hi, dtype=np.complex128, layout=_lib.DMM_B_PACKED, out=None):
        whole = m_lo == 0 and m_hi == self.telescope.mmax + 1
        if layout != _lib.DMM_B_PACKED or np.dtype(dtype) != self.store.dtype or (f_hi - f_lo > 1 and not whole):
            return super().beam_block(m_lo, m_hi, f_lo, f_hi, dtype, layout, out)
        a = f_lo * self.per_freq + self._m_off[m_lo]
        b = (f_hi - 1) * self.per_freq + self._m_off[m_hi]
        return self.store[a:b]  # a view: no packing, no copy

    @classmethod
    def from_provider(cls, provider, ctx, dtype=np.complex128, pin=True):
        """Materialise ``provider``'s tiles into a host store (filled on the GPU frequency by frequency when the
        provider generates on the device, so large stores do not take minutes of host hashing)."""
        import torch

        tel = provider.telescope
        npdt = np.dtype(dtype)
        n = cls.elements(tel)
        tdt = torch.complex128 if npdt == np.complex128 else torch.complex64
        host = torch.empty(n, dtype=tdt, pin_memory=bool(pin))
        per_freq = n // tel.nfreq
        b_dtype = _lib.DMM_C128 if npdt == np.complex128 else _lib.DMM_C64
        ms = np.arange(tel.mmax + 1, dtype=np.int32)
        sizes = np.array([provider.tile_elems(int(m), _lib.DMM_B_PACKED) for m in ms], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        dev = torch.empty(per_freq, dtype=tdt, device=ctx.device)
        for f in range(tel.nfreq):
            tiles = _lib.tile_array(ms, np.full(len(ms), f, np.int32), offs)
            provider.fill_pool(ctx, dev, tiles, b_dtype, _lib.DMM_B_PACKED)
            ctx.sync()
            host[f * per_freq : (f + 1) * per_freq].copy_(dev)
        del dev
        return cls(tel, host.numpy(), pinned=bool(pin), keepalive=host)


    # ---- one-time packing of a provider that only has per-tile ``beam_m`` (a real driftscan ``BeamTransfer``)
    @classmethod
    def open(cls, telescope, path, pinned=False):
        """A store packed earlier (``pack``): the ``.npy`` file memory-mapped read-only."""
        return cls(telescope, np.load(path, mmap_mode="r"), pinned=pinned)

    @classmethod
    def pack(cls, provider, path, dtype=np.complex128, processes=None, factory=None, chunk_bytes=256 << 20):
        """Write ``provider``'s tiles ONCE into a ``.npy`` file in the pool's wire format and return the store over it.

        For providers that only offer per-tile ``beam_m(m, fi=f)`` -- what a driftscan ``BeamTransfer`` offers
        (``mapmaker.py:160-162``): streamed tile by tile they are host-bound (14 GB/s through the staging ring, one
        interpreter packing under the GIL); packed once, every later day streams at the rate of a plain memory copy.
        The packing itself runs in ``processes`` worker PROCESSES (default: the host's share, at most 16), each
        writing its (frequency, m-range) chunks straight into the memory-mapped file.  How the workers get the
        provider: ``factory`` (a picklable callable that builds it inside a SPAWNED worker) if given; otherwise a
        pickled copy of ``provider`` handed to spawned workers when this process has already initialised the GPU (a
        forked child of a process that holds a HIP context and runtime threads is undefined: it can hang or crash);
        plain ``fork`` (workers inherit ``provider`` as it is) only while the GPU is untouched.  A provider that
        makes its tiles ON the GPU (``fill_mode == "device"``) cannot be packed by worker processes at all: that is
        refused unless ``processes == 1`` (packed by this process).
        """
        import multiprocessing as mp
        import pickle

        tel = provider.telescope
        npdt = np.dtype(dtype)
        n = cls.elements(tel)
        np.lib.format.open_memmap(path, mode="w+", dtype=npdt, shape=(n,)).flush()
        ntel = 2 * tel.npairs
        per_m = np.array([ntel * tel.num_pol_sky * (tel.lmax + 1 - m) for m in range(tel.mmax + 1)], dtype=np.int64)
        m_off = np.concatenate([[0], np.cumsum(per_m)])
        per_freq = int(m_off[-1])
        jobs = []
        for f in range(tel.nfreq):
            m0 = 0
            while m0 <= tel.mmax:  # chunks of whole tiles of about chunk_bytes
                m1 = int(np.searchsorted(m_off, m_off[m0] + chunk_bytes // npdt.itemsize, side="right")) - 1
                m1 = min(max(m1, m0 + 1), tel.mmax + 1)
                jobs.append((f, m0, m1, int(f * per_freq + m_off[m0])))
                m0 = m1
        if processes is None:
            from .hoststage import default_workers

            processes = default_workers()
        processes = max(1, min(int(processes), len(jobs)))
        if processes == 1:
            _pack_init(provider if factory is None else factory(), path)
            for j in jobs:
                _pack_job(j)
        else:
            if factory is None:
                if getattr(provider, "fill_mode", "host") == "device":
                    raise ValueError(f"PackedStoreProvider.pack: {type(provider).__name__} generates its tiles on the GPU (fill_mode 'device'); "
                                     "worker processes must not touch the GPU -- pack with processes=1, or pass a host-side factory")
                import torch

                if torch.cuda.is_initialized():
                    try:
                        factory = _Unpickle(pickle.dumps(provider, protocol=pickle.HIGHEST_PROTOCOL))
                    except Exception as exc:  # noqa: BLE001
                        raise RuntimeError("PackedStoreProvider.pack: this process has initialised the GPU, so its workers are spawned, not "
                                           f"forked, and need a picklable provider ({type(provider).__name__} is not: {exc!r}); pass factory=, "
                                           "or pack before the first GPU call") from exc
            ctx = mp.get_context("fork" if factory is None else "spawn")
            global _PACK_STATE
            _PACK_STATE = (provider, path) if factory is None else None  # forked workers inherit it
            try:
                with ctx.Pool(processes, initializer=_pack_init_worker, initargs=(factory, path)) as pool:
                    for _ in pool.imap_unordered(_pack_job, jobs, chunksize=1):
                        pass
            finally:
                _PACK_STATE = None
        return cls.open(tel, path)


class _Unpickle:
    """Picklable factory: the provider rebuilt from its pickle inside a spawned worker."""

    def __init__(self, blob):
        self.blob = blob

    def __call__(self):
        import pickle

        return pickle.loads(self.blob)


_PACK_STATE = None  # (provider, path) of the packing run in this process / inherited by its forked workers
_PACK_OUT = None


def _pack_init(provider, path):
    global _PACK_STATE, _PACK_OUT
    _PACK_STATE = (provider, path)
    _PACK_OUT = np.load(path, mmap_mode="r+")


def _pack_init_worker(factory, path):
    _pack_init(_PACK_STATE[0] if factory is None else factory(), path)


def _pack_job(job):
    """Tiles m0 .. m1-1 of frequency f, packed, into the file at element offset ``off``."""
    f, m0, m1, off = job
    prov = _PACK_STATE[0]
    tel = prov.telescope
    ntel = 2 * tel.npairs
    out = _PACK_OUT
    pos = off
    for m in range(m0, m1):
        b = np.asarray(prov.beam_m(m, fi=f)).reshape(ntel, tel.num_pol_sky, tel.lmax + 1)[..., m:]
        cnt = b.size
        np.copyto(out[pos : pos + cnt].reshape(b.shape), b, casting="same_kind")
        pos += cnt
    return pos - off


class PoolCycledProvider(BeamTransferProvider):
    """The *hbm-pool* residency policy of SURVEY 8d as a provider: frequency ``f`` uses the tiles of ``f % period``.

    The beam transfers of the metric configuration (1.64 TB at cfg 3) exceed one GPU; throughput runs therefore
    keep ``period`` frequencies' worth of DISTINCT tiles resident (far more than the 256 MiB Infinity Cache) and
    cycle the job's frequencies through them: every byte of B is read from HBM once per solve, tile contents repeat
    every ``period`` frequencies.  With this provider the policy runs through ``DirtyMapMaker.process`` itself: the
    engine sees that consecutive slabs hold the same contents and fills the pool once -- also across days.
    """

    def __init__(self, base, period):
        super().__init__(base.telescope)
        self.base = base
        self.alias_period = int(period)
        self.fill_mode = base.fill_mode
        self.block_is_pinned = base.block_is_pinned

    def content_key(self):
        return ("cycled", self.alias_period, self.base.content_key())

    def beam_m(self, m, fi=None):
        if fi is None:
            return np.stack([self.beam_m(m, fi=f) for f in range(self.telescope.nfreq)])
        return self.base.beam_m(m, fi=int(fi) % self.alias_period)

    def beam_block(self, m_lo, m_hi, f_lo, f_hi, dtype=np.complex128, layout=_lib.DMM_B_PACKED, out=None):
        if f_hi - f_lo == 1:
            f = f_lo % self.alias_period
            return self.base.beam_block(m_lo, m_hi, f, f + 1, dtype, layout, out)
        return super().beam_block(m_lo, m_hi, f_lo, f_hi, dtype, layout, out)

    def fill_pool(self, ctx, pool, tiles, dtype, layout, stager=None):
        rec = np.frombuffer(tiles, dtype=_lib._TILE_DTYPE)
        canon = _lib.tile_array(rec["m"], rec["f"] % self.alias_period, rec["b_off"])
        self.base.fill_pool(ctx, pool, canon, dtype, layout, stager)


class ForeignProvider(BeamTransferProvider):
    """Wrap any object with ``telescope`` + ``beam_m`` (e.g. driftscan's ``BeamTransfer``)."""

    stage_workers = 1  # a foreign object's ``beam_m`` (HDF5 reads) is not assumed to be thread-safe

    def __init__(self, bt):
        super().__init__(bt.telescope)
        self._bt = bt

    @property
    def ntel(self):
        return int(self._bt.ntel)

    @property
    def nsky(self):
        return int(self._bt.nsky)

    def beam_m(self, m, fi=None):
        return self._bt.beam_m(m, fi=fi)


class SVDBasisMixin:
    """The SVD-basis part of the provider protocol (what ``SVDModeProject`` needs, ``fgfilter.py:53-146``).

    driftscan's ``BeamTransfer`` [3P] keeps, per (m, frequency), the matrix that takes telescope-space m-modes to its
    reduced "SVD" degrees of freedom and one that takes them back; the reference reaches them through
    ``bt.project_vector_telescope_to_svd(mi, vec[nfreq, ntel]) -> packed modes`` (:87) and
    ``bt.project_vector_svd_to_telescope(mi, svec) -> [nfreq, 2, npairs]`` (:132), and sizes its containers with
    ``bt.ndofmax`` (:80).  Here the matrices themselves are the hand-over (bulk, like ``beam_block`` for B):

    * ``svd_len(m) -> int[nfreq]``            modes kept per frequency (their sum is ``<= ndofmax``)
    * ``beam_ut(m, f) -> [n_f, ntel]``        telescope -> SVD modes of frequency f
    * ``beam_ut_inv(m, f) -> [ntel, n_f]``    back (default: the conjugate transpose, exact for orthonormal rows)

    The two reference-visible methods are served from them by the batched GEMV kernel (``dmm_gemv_batch``).  What the
    matrices CONTAIN is driftscan's business: parity of that arithmetic is unpinned (driftscan absent).
    """

    ndofmax = 0

    def svd_len(self, m):
        raise NotImplementedError

    def beam_ut(self, m, f):
        raise NotImplementedError

    def beam_ut_inv(self, m, f):
        return np.conj(np.asarray(self.beam_ut(m, f))).T

    def project_vector_telescope_to_svd(self, mi, vec):
        from ..analysis import fgfilter

        return fgfilter.project_one_m(self, "forward", int(mi), np.asarray(vec))

    def project_vector_svd_to_telescope(self, mi, svec):
        from ..analysis import fgfilter

        return fgfilter.project_one_m(self, "backward", int(mi), np.asarray(svec))


class SVDArrayProvider(SVDBasisMixin, ArrayProvider):
    """:class:`ArrayProvider` + an SVD basis from memory: ``ut(m, f) -> [n_f, ntel]`` (and optionally ``ut_inv``)."""

    def __init__(self, telescope, beams, ut, ndofmax, ut_inv=None):
        ArrayProvider.__init__(self, telescope, beams)
        self._ut, self._ut_inv = ut, ut_inv
        self.ndofmax = int(ndofmax)

    def beam_ut(self, m, f):
        return np.asarray(self._ut(m, f))

    def beam_ut_inv(self, m, f):
        if self._ut_inv is None:
            return super().beam_ut_inv(m, f)
        return np.asarray(self._ut_inv(m, f))

    def svd_len(self, m):
        return np.array([self.beam_ut(m, f).shape[0] for f in range(self.telescope.nfreq)], dtype=np.int64)


class KLTransform:
    """A KL basis over the SVD modes, per m (driftscan ``KLTransform`` [3P] as the reference uses it,
    ``fgfilter.py:193,229``): ``modes(m) -> (evals [nkl], evecs [nkl, nsvd_m], inv [nsvd_m, nkl])``.

    ``project_vector_svd_to_kl(mi, vec, threshold)`` keeps the modes whose eigenvalue (signal-to-noise) is at least
    ``threshold`` (all of them for ``None``) and applies their rows of ``evecs``; ``project_vector_kl_to_svd`` applies
    the matching columns of ``inv``.  The selection rule and the matrices are driftscan's arithmetic [3P, recalled]:
    parity unpinned.
    """

    def __init__(self, modes):
        self._modes = modes

    def modes(self, m):
        ev, evecs, inv = self._modes(m)
        return np.asarray(ev, dtype=np.float64), np.asarray(evecs), np.asarray(inv)

    def kept(self, m, threshold):
        ev = self.modes(m)[0]
        return np.arange(ev.size) if threshold is None else np.flatnonzero(ev >= threshold)

    def project_vector_svd_to_kl(self, mi, vec, threshold=None):
        from ..analysis import fgfilter

        return fgfilter.kl_one_m(self, "forward", int(mi), np.asarray(vec), threshold)

    def project_vector_kl_to_svd(self, mi, vec, threshold=None):
        from ..analysis import fgfilter

        return fgfilter.kl_one_m(self, "backward", int(mi), np.asarray(vec), threshold)


class ProductManager:
    """Stand-in for ``drift.core.manager.ProductManager``: holds ``beamtransfer`` / ``telescope`` / ``kltransforms``."""

    def __init__(self, beamtransfer, kltransforms=None):
        self.beamtransfer = beamtransfer
        self.telescope = beamtransfer.telescope
        self.kltransforms = dict(kltransforms or {})

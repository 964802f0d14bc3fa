"""Driver for the batched (m, freq) solves: tile lists, the HBM pool of B, plans.

The reference walks ``for m: for f: _solve_m(m, f, v, Ni)`` and reads one ``beam_m`` tile
from disk per call (``mapmaker.py:79-94``).  Here the whole double loop is ONE kernel
launch per *slab*: a slab is as many (m, f) tiles as fit the HBM budget for B, filled in
bulk by the provider, described by a ``dmm_tile`` table (``include/draco_amd.h``).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..device import Context, ptr

_ELEM = {_lib.DMM_C64: 8, _lib.DMM_C128: 16}
_TORCH = {_lib.DMM_C64: torch.complex64, _lib.DMM_C128: torch.complex128}


# One block of HBM per device for B, kept between passes (a day's map-making is several passes: simulate, dirty,
# Wiener ...): it is a large fraction of the HBM, and handing it back to the caching allocator only to ask for a
# slightly different size a moment later ends in a second pool-sized allocation.  The block is used whole (one buffer)
# or as two halves (double buffering for providers whose tiles come over PCIe).  What each buffer currently HOLDS is
# remembered (`content`), so a slab whose tiles are already resident -- the same day again, or the hbm-pool policy's
# repeating frequencies -- is not filled a second time.
_POOLS: dict[int, torch.Tensor] = {}


class _Buffer:
    """One buffer of the device's B block: its memory, what it holds, and the events that order its reuse."""

    def __init__(self):
        self.mem = None  # uint8 view
        self.content = None  # key of the tiles it holds (None: garbage)
        self.filled = None  # event: the fill that produced `content` has finished (on the fill stream)
        self.last_use = None  # event: the last solve that read it has finished (on the caller's stream)
        # resident beam Gram products of the buffer's telescope-side tiles (ML with `cache_beam_gram`): the arrays, and the
        # buffer content they were computed from (anything else: stale, the library is told to start over)
        self.gram = None
        self.gram_valid = None
        self.gram_content = None
        # resident singular bases of the buffer's telescope-side tiles (ML with `cache_beam_basis`): U^H, sigma, rank per slot
        self.basis = None
        self.basis_content = None


_BUFFERS: dict[int, list[_Buffer]] = {}


def _take_block(ctx, nbytes):
    """The kept uint8 block of the context's device, grown if needed (growing discards what the buffers hold)."""
    kept = _POOLS.get(ctx.device_index)
    if kept is None or kept.numel() < nbytes:
        _POOLS.pop(ctx.device_index, None)
        _BUFFERS.pop(ctx.device_index, None)
        del kept
        torch.cuda.synchronize(ctx.device)
        torch.cuda.empty_cache()  # give the old block back before asking for a larger one
        kept = torch.empty(int(nbytes), dtype=torch.uint8, device=ctx.device)
        _POOLS[ctx.device_index] = kept
    return kept


def _buffers(ctx, nbuf, buf_bytes):
    """``nbuf`` buffers of ``buf_bytes`` carved from the device's block; contents survive while the carving does."""
    buf_bytes = (int(buf_bytes) + 255) & ~255
    block = _take_block(ctx, nbuf * buf_bytes)
    bufs = _BUFFERS.get(ctx.device_index)
    # a carving made for larger buffers still serves (its contents stay valid); anything else is re-carved
    if bufs is None or len(bufs) != nbuf or bufs[0].mem.numel() < buf_bytes:
        if bufs is not None:
            torch.cuda.synchronize(ctx.device)  # re-carving moves buffer boundaries under work that may be in flight
        size = block.numel() // nbuf & ~255
        bufs = []
        for k in range(nbuf):
            b = _Buffer()
            b.mem = block[k * size : (k + 1) * size]
            bufs.append(b)
        _BUFFERS[ctx.device_index] = bufs
    return bufs


def _tile_sizes(provider, ms, b_layout):
    """Elements of every tile of the list, padded to an even count (every tile stays 16-byte aligned for complex64
    too); the size depends on m only."""
    ms = np.asarray(ms, dtype=np.int64)
    if ms.size == 0:
        return np.zeros(0, dtype=np.int64)
    per_m = np.array([provider.tile_elems(int(m), b_layout) for m in range(int(ms.max()) + 1)], dtype=np.int64)
    per_m += per_m & 1
    return per_m[ms]


def release_pools():
    """Give the kept block back (to the caching allocator, and on to the device)."""
    torch.cuda.synchronize()
    _BUFFERS.clear()
    _POOLS.clear()
    torch.cuda.empty_cache()


class Slab:
    """A resident pool of B tiles + the plan that solves them."""

    def __init__(self, ctx, provider, ms, fs_data, fs_bt, b_dtype, b_layout, nfreq_data, n_m, pool=None, fill=True):
        tel = provider.telescope
        self.ctx = ctx
        self.ntile = len(ms)
        sizes = _tile_sizes(provider, ms, b_layout)
        offs = np.zeros(self.ntile, dtype=np.int64)
        if self.ntile:
            np.cumsum(sizes[:-1], out=offs[1:])
        self.nelem = int(sizes.sum())
        self.tiles = _lib.tile_array(ms, fs_data, offs)
        self.fill_tiles = _lib.tile_array(ms, fs_bt, offs)
        if pool is None or pool.numel() < self.nelem or pool.dtype != _TORCH[b_dtype]:
            pool = torch.empty(max(self.nelem, 1), dtype=_TORCH[b_dtype], device=ctx.device)
        self.pool = pool
        if fill:
            provider.fill_pool(ctx, pool, self.fill_tiles, b_dtype, b_layout)
        h = C.c_void_p()
        _lib.check(
            _lib.lib.dmm_solve_plan_create(
                ctx.handle, self.tiles, self.ntile, tel.npairs, tel.num_pol_sky, tel.lmax, nfreq_data, n_m, b_dtype, b_layout, C.byref(h)
            )
        )
        self.plan = h
        self.b_bytes = int(_lib.lib.dmm_plan_b_bytes(h))

    def close(self):
        if getattr(self, "plan", None):
            _lib.lib.dmm_plan_destroy(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the binding module may already be gone
            pass


class SolveEngine:
    """Runs one of {dirty, wiener, ml, project} over all (m <= mmax, f) with slabbed B.

    Slabs are consecutive ranges of the f-major / m-minor tile list.  Per slab: make sure its tiles are in a buffer
    of the device's B block (skipped when they already are), then hand the buffer + plan to the caller, who launches
    the solves on the current stream.  Fills run on a separate *fill context* (own stream) and are ordered against
    the solves by events only -- the host never waits for the GPU here:

    * a buffer is refilled only after the last solve that read it (``last_use``, recorded on the caller's stream);
    * a slab is solved only after its fill (``filled``, recorded on the fill stream).

    Providers whose tiles come from the host (``fill_mode == "host"``) get two buffers, so the upload of slab k+1
    (worker threads packing into pinned slots, copies on the fill stream) runs under the solves of slab k; providers
    that generate on the device get one (their fill is as HBM-bound as the solve: nothing to hide).
    Plans are kept for the engine's lifetime (``dmm_plan_destroy`` frees device memory, which synchronises).
    """

    _MAX_PLAN_KEYS = 4

    def __init__(self, provider, ctx=None, b_dtype=_lib.DMM_C128, b_layout=_lib.DMM_B_PACKED, pool_bytes=None, cache=True, gram_cache=False):
        self.provider = provider
        self.gram_cache = bool(gram_cache)  # ML: keep B B^H of the resident telescope-side tiles beside the B block (multi-day processing)
        self.basis_cache = False            # ML: keep their singular bases instead (`cache_beam_basis`)
        self.basis_rmax = 448
        self.ctx = ctx or Context.get()
        self.b_dtype = b_dtype
        self.b_layout = b_layout
        self.pool_bytes = pool_bytes
        self.cache = cache
        self._cached_key = None
        self._cached_slabs = None
        self._plans: dict = {}
        self._plan_keys: list = []  # keys of `_plans` entries, least recently used first
        self.last_b_bytes = 0
        self.fills = 0  # slabs actually filled (the others were resident)
        self.launch_events = None  # set to a list to collect (start, stop, b_bytes, ntile) per Dirty launch
        self._ws_offer = {}
        self._basis_wkey = None
        self._ws = None

    def close(self):
        for s in self._plans.values():
            s.close()
        self._plans.clear()
        self._plan_keys.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nbuf(self):
        return 2 if getattr(self.provider, "fill_mode", "host") == "host" else 1

    def _budget(self):
        """Bytes one buffer may take."""
        if self.pool_bytes is not None:
            return int(self.pool_bytes) // self.nbuf
        free, _total = torch.cuda.mem_get_info(self.ctx.device)
        kept = _POOLS.get(self.ctx.device_index)  # the previous pass's block is ours to take again
        return int((free + (kept.numel() if kept is not None else 0)) * 0.6) // self.nbuf

    def _slab_ranges(self, ms, n_m):
        """Split the tile list into consecutive ranges whose tiles fit one buffer; also the largest range's size in
        elements.  With an aliasing provider (hbm-pool policy) slabs are whole multiples of frequencies that divide
        the alias period, so that consecutive slabs hold identical contents."""
        budget = self._budget() // _ELEM[self.b_dtype]
        sizes = _tile_sizes(self.provider, ms, self.b_layout)
        if len(sizes) and int(sizes.max()) > budget:
            raise MemoryError(f"one B tile ({int(sizes.max())} elements) exceeds the pool budget ({budget})")
        cum = np.concatenate([[0], np.cumsum(sizes)])
        period = getattr(self.provider, "alias_period", None)
        ranges, start, largest = [], 0, 0
        if period and len(ms) % n_m == 0 and len(ms) // n_m > 1:
            per_freq = int(cum[n_m])
            g = max((d for d in range(1, int(period) + 1) if period % d == 0 and d * per_freq <= budget), default=0)
            if g:
                step = g * n_m
                while start < len(ms):
                    stop = min(start + step, len(ms))
                    ranges.append((start, stop))
                    largest = max(largest, int(cum[stop] - cum[start]))
                    start = stop
        if not ranges:
            while start < len(ms):  # greedy: as many consecutive tiles as fit (one searchsorted per slab)
                stop = int(np.searchsorted(cum, cum[start] + budget, side="right")) - 1
                stop = max(stop, start + 1)
                ranges.append((start, stop))
                largest = max(largest, int(cum[stop] - cum[start]))
                start = stop
        if not ranges:
            ranges.append((0, 0))
        self._pool_elems = largest
        return ranges

    def _content_key(self, ms, fs_bt):
        if not self.cache:
            return None  # never matches: every slab is filled
        canon = np.asarray(self.provider.canonical_freq(fs_bt), dtype=np.int32) if hasattr(self.provider, "canonical_freq") else fs_bt
        key_of = getattr(self.provider, "content_key", None)
        if key_of is None:
            return None  # a foreign provider says nothing about its contents: no reuse
        return (key_of(), self.b_dtype, self.b_layout, len(ms), ms.tobytes(), canon.tobytes())

    def slabs(self, freq_ind, mmax, nfreq_data, n_m):
        """Yield :class:`Slab` objects covering f-major, m-minor order; each is ready (on the current stream) when
        yielded and must have its solves launched on the current stream before the generator is advanced."""
        nf = len(freq_ind)
        ms = np.tile(np.arange(mmax + 1, dtype=np.int32), nf)
        fs_data = np.repeat(np.arange(nf, dtype=np.int32), mmax + 1)
        fs_bt = np.repeat(np.asarray(freq_ind, dtype=np.int32), mmax + 1)
        key = (tuple(int(f) for f in freq_ind), int(mmax), int(nfreq_data), int(n_m), self.b_dtype, self.b_layout)
        if self._cached_key == key and self._cached_slabs is not None:  # a slab installed by hand (single-tile solves)
            yield from self._cached_slabs
            return
        ctx = self.ctx
        ranges = self._slab_ranges(ms, mmax + 1)
        # plans are keyed by slab range; the ranges follow the HBM budget, which may drift between passes (free memory):
        # plans of a carving that is not this pass's are dropped instead of piling up with their device tables
        # -- only plans of THIS key: another frequency selection or mmax used alternately through the same engine keeps
        # its plans (dropping them costs a device-synchronising hipFree each and a rebuild every pass); at most
        # `_MAX_PLAN_KEYS` keys are kept, the least recently used one goes first
        live = {(key, a, b) for a, b in ranges}
        for stale in [k for k in self._plans if k[0] == key and k not in live]:
            self._plans.pop(stale).close()
        if key in self._plan_keys:
            self._plan_keys.remove(key)
        self._plan_keys.append(key)
        while len(self._plan_keys) > self._MAX_PLAN_KEYS:
            old = self._plan_keys.pop(0)
            for stale in [k for k in self._plans if k[0] == old]:
                self._plans.pop(stale).close()
        es = _ELEM[self.b_dtype]
        bufs = _buffers(ctx, self.nbuf, max(self._pool_elems, 1) * es)
        main = torch.cuda.current_stream(ctx.device)
        fill_ctx = Context.fill(ctx.device_index) if self.nbuf > 1 else ctx
        fill_stream = fill_ctx.stream if fill_ctx.stream is not None else main
        for k, (a, b) in enumerate(ranges):
            pkey = (key, a, b)
            s = self._plans.get(pkey)
            if s is None:
                s = Slab(ctx, self.provider, ms[a:b], fs_data[a:b], fs_bt[a:b], self.b_dtype, self.b_layout, nfreq_data, n_m,
                         pool=bufs[0].mem.view(_TORCH[self.b_dtype]), fill=False)
                s.pool = None
                self._plans[pkey] = s
            content = self._content_key(ms[a:b], fs_bt[a:b])
            buf = next((x for x in bufs if content is not None and x.content == content), None)
            if buf is None:
                buf = bufs[k % len(bufs)]
                buf.content = None
                if buf.last_use is not None and fill_stream != main:
                    fill_stream.wait_event(buf.last_use)  # the solves still reading the buffer
                pool = buf.mem.view(_TORCH[self.b_dtype])
                self.provider.fill_pool(fill_ctx, pool, s.fill_tiles, self.b_dtype, self.b_layout)
                buf.filled = torch.cuda.Event()
                buf.filled.record(fill_stream)
                buf.content = content
                self.fills += 1
            if buf.filled is not None and fill_stream != main:
                main.wait_event(buf.filled)
            s.pool = buf.mem.view(_TORCH[self.b_dtype])
            s.buf = buf
            try:
                yield s
            finally:
                s.pool = None  # plans outlive the pass; they must not keep the device's B block alive (release_pools)
                s.buf = None
                # also when the consumer raised or closed the generator after launching: whatever it did enqueue on
                # `main` reads the buffer, and a later pass must not refill it underneath (recording is free)
                buf.last_use = torch.cuda.Event()
                buf.last_use.record(main)

    #: largest ratio of a day's non-zero noise weights for which the resident bases are used (they are truncated at 1e-15
    #: of lambda_max of the UNWEIGHTED B B^H: the dropped modes enter the day's Gram matrix at up to 1e-15 ratio^2 of its
    #: lambda_max, and pinv_svd cuts at 1e-6 -- ADVICE r4); beyond it the day takes the full-order path
    basis_max_weight_ratio = 1e6

    def _basis_weights_ok(self, weights):
        """One reduction and one host read per CALL of solve / solve_many (not per slab): the verdict is remembered for
        the weight arrays it was taken on and forgotten when the next call starts -- the next day's weights are normally
        handed the same block by the caching allocator, with the same version counter."""
        key = tuple((int(w.data_ptr()), int(w._version), tuple(w.shape)) for w in weights)
        if getattr(self, "_basis_wkey", None) != key:
            hi = max(float(w.max()) for w in weights)
            lo = min(float(torch.where(w > 0, w, torch.full_like(w, float("inf"))).min()) for w in weights)
            # (D = sqrt(w): (d_max / d_min)^2 = w_max / w_min -> dropped modes at <= 1e-15 * 1e6 = 1e-9 of lambda_max)
            self._basis_wok = bool(hi <= 0 or lo == float("inf") or hi / lo <= self.basis_max_weight_ratio)
            self._basis_wkey = key
        return self._basis_wok

    def _basis_on(self, slab, mvis_d, mweight_d, params, ws, alm, weights=None):
        """Hand the library the resident singular bases of the slab's buffer (``basis_cache``), building them first -- one
        decomposition of B B^H per telescope-side tile, with unit weights -- when the buffer holds other tiles than the
        ones they were computed from.  Not for days whose weights span too many decades (``basis_max_weight_ratio``)."""
        buf = getattr(slab, "buf", None)
        if not self.basis_cache or buf is None or buf.content is None:
            return False
        if not self._basis_weights_ok(weights if weights is not None else [mweight_d]):
            return False
        lib = _lib.lib
        nslots = int(lib.dmm_ml_gram_cache_slots(slab.plan))
        if nslots == 0:
            return False
        ntel = 2 * self.provider.telescope.npairs
        rmax = int(self.basis_rmax)
        if buf.basis is None or buf.basis[2].numel() < nslots or buf.basis[3] != (rmax, ntel):
            buf.basis = None
            U = torch.empty(nslots * rmax * ntel, dtype=torch.complex128, device=self.ctx.device)
            sg = torch.empty(nslots * rmax, dtype=torch.float64, device=self.ctx.device)
            rk = torch.empty(nslots, dtype=torch.int32, device=self.ctx.device)
            buf.basis = (U, sg, rk, (rmax, ntel))
            buf.basis_content = None
        U, sg, rk, _ = buf.basis
        if buf.basis_content != buf.content:
            _lib.check(lib.dmm_ctx_set_ml_basis(self.ctx.handle, ptr(U), ptr(sg), ptr(rk), nslots, rmax, 1))
            try:
                ones = torch.ones_like(mweight_d)
                _lib.check(lib.dmm_ml_run(slab.plan, ptr(slab.pool), ptr(mvis_d), ptr(ones), float(params.get("acond", 1e-4)),
                                          float(params.get("rcond", 1e-3)), ptr(ws), ptr(alm)))
                torch.cuda.current_stream(self.ctx.device).synchronize()  # (`ones` dies here)
            finally:
                _lib.check(lib.dmm_ctx_set_ml_basis(self.ctx.handle, None, None, None, 0, 0, 0))
            buf.basis_content = buf.content
            self.basis_builds = getattr(self, "basis_builds", 0) + 1
        _lib.check(lib.dmm_ctx_set_ml_basis(self.ctx.handle, ptr(U), ptr(sg), ptr(rk), nslots, rmax, 0))
        return True

    def _gram_cache_on(self, slab, tag=("ml",)):
        """Hand the library the resident beam Gram products of the slab's buffer (``gram_cache``): the arrays live with
        the buffer and are started over whenever it holds other tiles than the ones they were computed from -- or the
        products of another maker / prior (``tag``: ML keeps B B^H, Wiener B S B^H)."""
        buf = getattr(slab, "buf", None)
        if not self.gram_cache or buf is None or buf.content is None:
            return False
        lib = _lib.lib
        nslots = int(lib.dmm_ml_gram_cache_slots(slab.plan))
        nbytes = int(lib.dmm_ml_gram_cache_bytes(slab.plan))
        if nslots == 0:
            return False
        reset = 0
        if buf.gram is None or buf.gram.numel() < nbytes or buf.gram_valid.numel() < nslots:
            buf.gram = buf.gram_valid = None
            buf.gram = torch.empty(nbytes, dtype=torch.uint8, device=self.ctx.device)
            buf.gram_valid = torch.zeros(nslots, dtype=torch.int32, device=self.ctx.device)
            buf.gram_content = None
        if buf.gram_content != (tag, buf.content):
            reset = 1
            buf.gram_content = (tag, buf.content)
        _lib.check(lib.dmm_ctx_set_ml_gram_cache(self.ctx.handle, ptr(buf.gram), ptr(buf.gram_valid), nslots, reset))
        return True

    # ---- the four batched operations
    def solve(self, kind, mvis_d, mweight_d, freq_ind, mmax, on_freqs_done=None, **params):
        """``alm [nfreq, npol, mmax+1, lmax+1]`` complex128 on the device.

        ``on_freqs_done(alm, f0, f1)`` is called after each slab's launch with the range of
        (data) frequencies whose every m has now been issued, so a caller can start the next
        stage for them on another stream while the following slab is filled and solved.
        """
        tel = self.provider.telescope
        n_m_data, _, nfreq, npairs = mvis_d.shape
        if npairs != tel.npairs:
            raise ValueError(f"m-modes have {npairs} baselines, the beam transfers {tel.npairs}")
        n_m = mmax + 1
        alm = torch.empty((nfreq, tel.num_pol_sky, n_m, tel.lmax + 1), dtype=torch.complex128, device=self.ctx.device)
        self.last_b_bytes = 0
        self._ws_offer = {}
        self._basis_wkey = None  # (the weight-range verdict of the resident bases is per call)
        self._ws = None
        lib = _lib.lib
        issued, f_done = 0, 0
        for slab in self.slabs(freq_ind, mmax, nfreq, n_m):
            self.last_b_bytes += slab.b_bytes
            if kind == "dirty":
                if self.launch_events is not None:  # HIP events on the launch stream, read by the caller after a sync
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.check(lib.dmm_dirty_run(slab.plan, ptr(slab.pool), ptr(mvis_d), ptr(mweight_d), ptr(alm)))
                if self.launch_events is not None:
                    e1.record()
                    self.launch_events.append((e0, e1, slab.b_bytes, slab.ntile))
            elif kind == "wiener":
                self._offer_workspace(b"wiener_workspace_mib", 24 << 10)  # 0.0649 -> 0.0620 ms per cfg-3 solve against 6 GiB
                ws = self._workspace(int(lib.dmm_wiener_workspace_bytes(slab.plan)))
                cached = self._gram_cache_on(slab, ("wiener", float(params["prior_amp"]), float(params["prior_tilt"])))
                try:
                    _lib.check(
                        lib.dmm_wiener_run(
                            slab.plan, ptr(slab.pool), ptr(mvis_d), ptr(mweight_d), float(params["prior_amp"]), float(params["prior_tilt"]), ptr(ws), ptr(alm)
                        )
                    )
                finally:
                    if cached:
                        _lib.check(lib.dmm_ctx_set_ml_gram_cache(self.ctx.handle, None, None, 0, 0))
            elif kind == "ml":
                self._offer_workspace(b"ml_workspace_mib", 64 << 10)
                ws = self._workspace(int(lib.dmm_ml_workspace_bytes(slab.plan)))
                based = self._basis_on(slab, mvis_d, mweight_d, params, ws, alm)
                cached = self._gram_cache_on(slab)
                try:
                    _lib.check(
                        lib.dmm_ml_run(
                            slab.plan, ptr(slab.pool), ptr(mvis_d), ptr(mweight_d), float(params.get("acond", 1e-4)), float(params.get("rcond", 1e-3)), ptr(ws), ptr(alm)
                        )
                    )
                finally:
                    if cached:
                        _lib.check(lib.dmm_ctx_set_ml_gram_cache(self.ctx.handle, None, None, 0, 0))
                    if based:
                        _lib.check(lib.dmm_ctx_set_ml_basis(self.ctx.handle, None, None, None, 0, 0, 0))
            else:
                raise ValueError(kind)
            issued += slab.ntile  # slabs are consecutive ranges of the f-major, m-minor tile list
            if on_freqs_done is not None and issued // n_m > f_done:
                on_freqs_done(alm, f_done, issued // n_m)
                f_done = issued // n_m
        self._ws = None
        return alm

    def solve_many(self, kind, mvis_l, mweight_l, freq_ind, mmax, on_freqs_done=None, **params):
        """D sidereal days against ONE pass over B: a list of ``alm`` as :meth:`solve` returns them, one per day.

        Every slab of B is made resident once (one PCIe crossing for providers whose tiles live on the host) and
        serves all D days before the next slab replaces it.  ``"dirty"``: ``dmm_dirty_run_multi`` -- up to eight days
        share every tile READ too (the kernel keeps eight accumulators per column), each day bit-identical to its
        own :meth:`solve`.  ``"wiener"`` / ``"ml"``: the Gram matrices depend on the day's noise weights, so the days
        run one after the other inside the slab.  ``on_freqs_done(d, alm_d, f0, f1)`` as in :meth:`solve`, per day.
        """
        tel = self.provider.telescope
        D = len(mvis_l)
        if D == 0:
            return []
        if len(mweight_l) != D:
            raise ValueError("solve_many: one weight array per day")
        shape = tuple(mvis_l[0].shape)
        for v, w in zip(mvis_l, mweight_l):
            if tuple(v.shape) != shape or tuple(w.shape) != shape:
                raise ValueError("solve_many: every day must have the same [m, msign, freq, stack] shape")
        n_m_data, _, nfreq, npairs = shape
        if npairs != tel.npairs:
            raise ValueError(f"m-modes have {npairs} baselines, the beam transfers {tel.npairs}")
        n_m = mmax + 1
        # (ONE allocation for the group: the caching allocator then recycles one block per group instead of juggling D of them
        # against the side stream's events -- every fresh hipMalloc of a few GB stalls the device for tens of milliseconds)
        alm_all = torch.empty((D, nfreq, tel.num_pol_sky, n_m, tel.lmax + 1), dtype=torch.complex128, device=self.ctx.device)
        alms = [alm_all[d] for d in range(D)]
        self.last_b_bytes = 0
        self._ws_offer = {}
        self._basis_wkey = None
        self._ws = None
        lib = _lib.lib
        PA = C.c_void_p * D
        pv, pw, pa = PA(*[ptr(x) for x in mvis_l]), PA(*[ptr(x) for x in mweight_l]), PA(*[ptr(x) for x in alms])
        issued, f_done = 0, 0
        for slab in self.slabs(freq_ind, mmax, nfreq, n_m):
            self.last_b_bytes += slab.b_bytes
            if kind == "dirty":
                if self.launch_events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.check(lib.dmm_dirty_run_multi(slab.plan, ptr(slab.pool), pv, pw, pa, D))
                if self.launch_events is not None:
                    e1.record()
                    self.launch_events.append((e0, e1, slab.b_bytes, slab.ntile))
            elif kind == "wiener":
                self._offer_workspace(b"wiener_workspace_mib", 24 << 10)
                ws = self._workspace(int(lib.dmm_wiener_workspace_bytes(slab.plan)))
                cached = self._gram_cache_on(slab, ("wiener", float(params["prior_amp"]), float(params["prior_tilt"])))
                try:
                    for d in range(D):
                        _lib.check(lib.dmm_wiener_run(slab.plan, ptr(slab.pool), ptr(mvis_l[d]), ptr(mweight_l[d]), float(params["prior_amp"]),
                                                      float(params["prior_tilt"]), ptr(ws), ptr(alms[d])))
                finally:
                    if cached:
                        _lib.check(lib.dmm_ctx_set_ml_gram_cache(self.ctx.handle, None, None, 0, 0))
            elif kind == "ml":
                self._offer_workspace(b"ml_workspace_mib", 64 << 10)
                ws = self._workspace(int(lib.dmm_ml_workspace_bytes(slab.plan)))
                based = self._basis_on(slab, mvis_l[0], mweight_l[0], params, ws, alms[0], weights=mweight_l)
                cached = self._gram_cache_on(slab)
                try:
                    for d in range(D):
                        _lib.check(lib.dmm_ml_run(slab.plan, ptr(slab.pool), ptr(mvis_l[d]), ptr(mweight_l[d]), float(params.get("acond", 1e-4)),
                                                  float(params.get("rcond", 1e-3)), ptr(ws), ptr(alms[d])))
                finally:
                    if cached:
                        _lib.check(lib.dmm_ctx_set_ml_gram_cache(self.ctx.handle, None, None, 0, 0))
                    if based:
                        _lib.check(lib.dmm_ctx_set_ml_basis(self.ctx.handle, None, None, None, 0, 0, 0))
            else:
                raise ValueError(kind)
            issued += slab.ntile
            if on_freqs_done is not None and issued // n_m > f_done:
                for d in range(D):
                    on_freqs_done(d, alms[d], f_done, issued // n_m)
                f_done = issued // n_m
        self._ws = None
        return alms

    def _workspace(self, nbytes):
        """The pass's workspace: one allocation, reused by every slab it is large enough for (the library drains its
        own streams before a run returns, and the caller's stream orders the slabs)."""
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.ctx.device)
        return self._ws

    def _offer_workspace(self, option, cap_mib):
        """Let the dense solvers size their sub-batches for the HBM that is actually free (B block, m-modes and a_lm
        are allocated by now): half of it, at most ``cap_mib``.  The ML eigen pass pays a fixed cost per Householder
        column; with 64 GiB (1400 cfg-3 matrices per half-batch instead of 357) that cost is shared four times wider.

        Decided ONCE per pass (``solve`` clears ``_ws_offer``): the driver's free-memory figure does not count blocks
        torch holds in its cache, so asking again per slab gave a slightly different size each time -- and a new
        allocation of tens of GiB per slab.  What torch has reserved but not handed out is ours as well."""
        if option in self._ws_offer:
            return
        free, _ = torch.cuda.mem_get_info(self.ctx.device)
        st = torch.cuda.memory_stats(self.ctx.device)
        free += max(int(st.get("reserved_bytes.all.current", 0)) - int(st.get("allocated_bytes.all.current", 0)), 0)
        mib = min(int(cap_mib), int(free * 0.5) >> 20)
        self._ws_offer[option] = mib
        # (0 = the library's own default: an offer that fell below 1 GiB must not leave an earlier, larger one standing)
        _lib.check(_lib.lib.dmm_ctx_set_option(self.ctx.handle, option, mib if mib >= 1024 else 0))

    def project(self, alm_d, freq_ind, mmax):
        """``vis [mmax+1, 2, nfreq, npairs] = B_m[f] a_m[f]`` (``stream.py:109-112``)."""
        tel = self.provider.telescope
        nfreq, npol, n_m, nl = alm_d.shape
        assert npol == tel.num_pol_sky and nl == tel.lmax + 1 and n_m == mmax + 1
        vis = torch.empty((n_m, 2, nfreq, tel.npairs), dtype=torch.complex128, device=self.ctx.device)
        self.last_b_bytes = 0
        for slab in self.slabs(freq_ind, mmax, nfreq, n_m):
            self.last_b_bytes += slab.b_bytes
            if self.launch_events is not None:  # (HIP events on the launch stream, as around the Dirty launches)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            _lib.check(_lib.lib.dmm_project_run(slab.plan, ptr(slab.pool), ptr(alm_d), ptr(vis)))
            if self.launch_events is not None:
                e1.record()
                self.launch_events.append((e0, e1, slab.b_bytes, slab.ntile))
        return vis


def project_single_m(provider, mi, vec):
    """Host-facing ``project_vector_sky_to_telescope(mi, vec[nfreq, npol, lmax+1]) -> [nfreq, ntel]``."""
    ctx = Context.get()
    tel = provider.telescope
    vec = np.asarray(vec, dtype=np.complex128)
    nfreq = vec.shape[0]
    ms = np.full(nfreq, mi, dtype=np.int32)
    fs = np.arange(nfreq, dtype=np.int32)
    # a one-m batch: alm buffer [nfreq, npol, n_m = mi+1, lmax+1] with only row mi populated
    alm = torch.zeros((nfreq, tel.num_pol_sky, mi + 1, tel.lmax + 1), dtype=torch.complex128, device=ctx.device)
    alm[:, :, mi, :] = ctx.to_device(vec)
    slab = Slab(ctx, provider, ms, fs, fs, _lib.DMM_C128, _lib.DMM_B_PACKED, nfreq, mi + 1)
    vis = torch.zeros((mi + 1, 2, nfreq, tel.npairs), dtype=torch.complex128, device=ctx.device)
    _lib.check(_lib.lib.dmm_project_run(slab.plan, ptr(slab.pool), ptr(alm), ptr(vis)))
    out = vis[mi].permute(1, 0, 2).reshape(nfreq, 2 * tel.npairs).cpu().numpy()
    slab.close()
    return out

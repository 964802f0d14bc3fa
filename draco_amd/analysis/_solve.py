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


# One multi-slab pool per device, kept between passes (a day's map-making is several passes: simulate, dirty, Wiener
# ...): pools are a large fraction of the HBM, and handing one back to the caching allocator only to ask for a
# slightly different size a moment later ends in a second pool-sized allocation.  Contents never outlive a slab.
_POOLS: dict[int, torch.Tensor] = {}


def _take_pool(ctx, nelem, b_dtype):
    """A ``[nelem]`` tensor of the B storage type on the context's device, carved from the kept pool when it fits."""
    need = int(nelem) * _ELEM[b_dtype]
    kept = _POOLS.get(ctx.device_index)
    if kept is None or kept.numel() < need:
        _POOLS.pop(ctx.device_index, None)
        del kept
        torch.cuda.empty_cache()  # give the old block back before asking for a larger one
        kept = torch.empty(need, dtype=torch.uint8, device=ctx.device)
        _POOLS[ctx.device_index] = kept
    return kept[:need].view(_TORCH[b_dtype])


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
    """Give the kept pools back (to the caching allocator, and on to the device)."""
    _POOLS.clear()
    torch.cuda.empty_cache()


class Slab:
    """A resident pool of B tiles + the plan that solves them."""

    def __init__(self, ctx, provider, ms, fs_data, fs_bt, b_dtype, b_layout, nfreq_data, n_m, pool=None):
        tel = provider.telescope
        self.ctx = ctx
        self.ntile = len(ms)
        sizes = _tile_sizes(provider, ms, b_layout)
        offs = np.zeros(self.ntile, dtype=np.int64)
        if self.ntile:
            np.cumsum(sizes[:-1], out=offs[1:])
        self.nelem = int(sizes.sum())
        self.tiles = _lib.tile_array(ms, fs_data, offs)
        fill_tiles = _lib.tile_array(ms, fs_bt, offs)
        if pool is None or pool.numel() < self.nelem or pool.dtype != _TORCH[b_dtype]:
            pool = torch.empty(max(self.nelem, 1), dtype=_TORCH[b_dtype], device=ctx.device)
        self.pool = pool
        provider.fill_pool(ctx, pool, fill_tiles, b_dtype, b_layout)
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
    """Runs one of {dirty, wiener, ml, project} over all (m <= mmax, f) with slabbed B."""

    def __init__(self, provider, ctx=None, b_dtype=_lib.DMM_C128, b_layout=_lib.DMM_B_PACKED, pool_bytes=None, cache=True):
        self.provider = provider
        self.ctx = ctx or Context.get()
        self.b_dtype = b_dtype
        self.b_layout = b_layout
        self.pool_bytes = pool_bytes
        self.cache = cache
        self._cached_key = None
        self._cached_slabs = None
        self.last_b_bytes = 0

    def _budget(self):
        if self.pool_bytes is not None:
            return int(self.pool_bytes)
        free, _total = torch.cuda.mem_get_info(self.ctx.device)
        kept = _POOLS.get(self.ctx.device_index)  # the previous pass's pool is ours to take again
        return int((free + (kept.numel() if kept is not None else 0)) * 0.6)

    def _slab_ranges(self, ms):
        """Split the tile list into consecutive ranges whose pool fits the budget; also the largest range's size
        in elements (the one pool every slab of the pass is filled into)."""
        budget = self._budget() // _ELEM[self.b_dtype]
        sizes = _tile_sizes(self.provider, ms, self.b_layout)
        if len(sizes) and int(sizes.max()) > budget:
            raise MemoryError(f"one B tile ({int(sizes.max())} elements) exceeds the pool budget ({budget})")
        cum = np.concatenate([[0], np.cumsum(sizes)])
        ranges, start, largest = [], 0, 0
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

    def slabs(self, freq_ind, mmax, nfreq_data, n_m):
        """Yield :class:`Slab` objects covering f-major, m-minor order (cached if it is one slab)."""
        nf = len(freq_ind)
        ms = np.tile(np.arange(mmax + 1, dtype=np.int32), nf)
        fs_data = np.repeat(np.arange(nf, dtype=np.int32), mmax + 1)
        fs_bt = np.repeat(np.asarray(freq_ind, dtype=np.int32), mmax + 1)
        key = (tuple(int(f) for f in freq_ind), int(mmax), int(nfreq_data), int(n_m), self.b_dtype, self.b_layout)
        if self.cache and self._cached_key == key:
            yield from self._cached_slabs
            return
        ranges = self._slab_ranges(ms)
        if self.cache and len(ranges) == 1:
            a, b = ranges[0]
            s = Slab(self.ctx, self.provider, ms[a:b], fs_data[a:b], fs_bt[a:b], self.b_dtype, self.b_layout, nfreq_data, n_m)
            self._cached_key, self._cached_slabs = key, [s]
            yield s
            return
        # one pool for the whole pass, sized for its largest slab (slabs differ by a few tiles: growing the pool
        # for a later one would need a second pool-sized allocation while the first is still alive)
        pool = _take_pool(self.ctx, max(self._pool_elems, 1), self.b_dtype)
        for a, b in ranges:
            s = Slab(self.ctx, self.provider, ms[a:b], fs_data[a:b], fs_bt[a:b], self.b_dtype, self.b_layout, nfreq_data, n_m, pool)
            pool = s.pool
            try:
                yield s
            finally:
                self.ctx.sync()  # the pool is about to be overwritten by the next slab
                s.close()

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
        lib = _lib.lib
        issued, f_done = 0, 0
        for slab in self.slabs(freq_ind, mmax, nfreq, n_m):
            self.last_b_bytes += slab.b_bytes
            if kind == "dirty":
                _lib.check(lib.dmm_dirty_run(slab.plan, ptr(slab.pool), ptr(mvis_d), ptr(mweight_d), ptr(alm)))
            elif kind == "wiener":
                ws = torch.empty(max(int(lib.dmm_wiener_workspace_bytes(slab.plan)), 16), dtype=torch.uint8, device=self.ctx.device)
                _lib.check(
                    lib.dmm_wiener_run(
                        slab.plan, ptr(slab.pool), ptr(mvis_d), ptr(mweight_d), float(params["prior_amp"]), float(params["prior_tilt"]), ptr(ws), ptr(alm)
                    )
                )
            elif kind == "ml":
                ws = torch.empty(max(int(lib.dmm_ml_workspace_bytes(slab.plan)), 16), dtype=torch.uint8, device=self.ctx.device)
                _lib.check(
                    lib.dmm_ml_run(
                        slab.plan, ptr(slab.pool), ptr(mvis_d), ptr(mweight_d), float(params.get("acond", 1e-4)), float(params.get("rcond", 1e-3)), ptr(ws), ptr(alm)
                    )
                )
            else:
                raise ValueError(kind)
            issued += slab.ntile  # slabs are consecutive ranges of the f-major, m-minor tile list
            if on_freqs_done is not None and issued // n_m > f_done:
                on_freqs_done(alm, f_done, issued // n_m)
                f_done = issued // n_m
        return alm

    def project(self, alm_d, freq_ind, mmax):
        """``vis [mmax+1, 2, nfreq, npairs] = B_m[f] a_m[f]`` (``stream.py:109-112``)."""
        tel = self.provider.telescope
        nfreq, npol, n_m, nl = alm_d.shape
        assert npol == tel.num_pol_sky and nl == tel.lmax + 1 and n_m == mmax + 1
        vis = torch.empty((n_m, 2, nfreq, tel.npairs), dtype=torch.complex128, device=self.ctx.device)
        self.last_b_bytes = 0
        for slab in self.slabs(freq_ind, mmax, nfreq, n_m):
            self.last_b_bytes += slab.b_bytes
            _lib.check(_lib.lib.dmm_project_run(slab.plan, ptr(slab.pool), ptr(alm_d), ptr(vis)))
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

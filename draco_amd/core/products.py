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

    def __init__(self, frequencies, lmax, mmax=None, num_pol_sky=4, ncyl=1, nfeed_cyl=8, npol_feed=2, npairs=None):
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
                if ckey < key:
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


class BeamTransferProvider:
    """Base provider: the reference-visible protocol + the bulk pool fill."""

    def __init__(self, telescope):
        self.telescope = telescope

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

    # ---- bulk hand-over
    def tile_elems(self, m, layout):
        tel = self.telescope
        w = tel.lmax + 1 if layout == _lib.DMM_B_FULL else tel.lmax + 1 - m
        return self.ntel * tel.num_pol_sky * w

    def fill_pool(self, ctx, pool, tiles, dtype, layout):
        """Write the tiles described by ``tiles`` (ctypes ``dmm_tile`` array) into ``pool``.

        Generic implementation: one ``beam_m`` call per tile, pack on the host, copy up.
        Providers that can do better (procedural, already-resident) override this.
        """
        import torch

        tel = self.telescope
        npdt = np.complex128 if dtype == _lib.DMM_C128 else np.complex64
        flat = pool.view(-1)
        for t in tiles:
            b = np.asarray(self.beam_m(t.m, fi=t.f)).reshape(self.ntel, tel.num_pol_sky, tel.lmax + 1)
            if layout == _lib.DMM_B_PACKED:
                b = b[..., t.m :]
            h = np.ascontiguousarray(b, dtype=npdt).reshape(-1)
            flat[t.b_off : t.b_off + h.size].copy_(torch.from_numpy(h), non_blocking=False)


class SyntheticProvider(BeamTransferProvider):
    """Seeded procedural B tiles, ``B ~ U``-complex with variance ``1/ntel``, zeros for ``l < m``."""

    def __init__(self, telescope, seed=3000):
        super().__init__(telescope)
        self.seed = int(seed)

    def beam_m(self, m, fi=None):
        tel = self.telescope
        if fi is None:
            return np.stack([self.beam_m(m, fi=f) for f in range(tel.nfreq)])
        return synth_beam_tile(self.seed, m, fi, tel.npairs, tel.num_pol_sky, tel.lmax)

    def fill_pool(self, ctx, pool, tiles, dtype, layout):
        from ..device import ptr

        tel = self.telescope
        _lib.check(
            _lib.lib.dmm_synth_beam_fill(
                ctx.handle, tiles, len(tiles), tel.npairs, tel.num_pol_sky, tel.lmax, dtype, layout, self.seed, ptr(pool)
            )
        )


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


class ForeignProvider(BeamTransferProvider):
    """Wrap any object with ``telescope`` + ``beam_m`` (e.g. driftscan's ``BeamTransfer``)."""

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


class ProductManager:
    """Stand-in for ``drift.core.manager.ProductManager``: holds ``beamtransfer`` / ``telescope``."""

    def __init__(self, beamtransfer):
        self.beamtransfer = beamtransfer
        self.telescope = beamtransfer.telescope

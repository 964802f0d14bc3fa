"""Deconvolving ring-map makers on hybrid beam-formed m-modes, on the GPU.

Drop-in for ``DeconvolveHybridMBase`` / ``TikhonovRingMapMaker`` / ``WienerRingMapMaker``
(``draco/analysis/ringmapmaker.py:538-930, 1075-1194``): same config attributes, the same
``setup(manager=None)`` / ``process(hybrid_vis_m, hybrid_beam_m) -> RingMap`` signatures and
validation errors.  This is the production CHIME map-maker; it consumes the
``HybridVisMModes`` that :class:`draco_amd.analysis.transform.MModeTransform` produces.

The per-frequency NumPy loop of the reference (``ringmapmaker.py:744-823``) runs as four
kernels over all local frequencies at once (``csrc/ringmap.hip``).  Arithmetic is float64
throughout.  The ``DeconvolveAnalyticalBeam`` variants (``ringmapmaker.py:968-1072, 1189-1190``) generate
their beam m-modes on the device (``dmm_analytic_beam_mmodes``) and then run the same kernels.
"""

from __future__ import annotations

import numpy as np
import scipy.constants

from .. import _lib
from ..core import containers, io
from ..core.task import ContainerTask
from ..device import Context, ptr
from ..util import tools
from .transform import _dev_dataset

_WINDOW_COEF = {
    "uniform": (1, 0, 0, 0),
    "hann": (0.5, -0.5, 0, 0),
    "hanning": (0.5, -0.5, 0, 0),
    "hamming": (0.53836, -0.46164, 0, 0),
    "blackman": (0.42, -0.5, 0.08, 0),
    "nuttall": (0.355768, -0.487396, 0.144232, -0.012604),
    "blackman_nuttall": (0.3635819, -0.4891775, 0.1365995, -0.0106411),
    "blackman_harris": (0.35875, -0.48829, 0.14128, -0.01168),
}


def window_generalised(x, window="nuttall"):
    """Cosine-sum window at arbitrary locations, zero outside [0, 1] (``util/tools.py:547-601``)."""
    a = np.asarray(_WINDOW_COEF[window], dtype=np.float64)
    w = sum(a[i] * np.cos(2 * np.pi * i * x) for i in range(4))
    return np.where((x >= 0) & (x <= 1), w, 0)


class DeconvolveHybridMBase(ContainerTask):
    """Base class for deconvolving ring-map makers (``ringmapmaker.py:538-930``); subclasses define the
    EW weighting and the regularisation."""

    exclude_cyl = ()
    exclude_intracyl = False
    skip_deconvolution = False
    reference_declination = None
    save_dirty_beam = False
    window_type = "none"
    window_size = 1.0
    window_scaled = False
    _config_names = ("exclude_cyl", "exclude_intracyl", "skip_deconvolution", "reference_declination", "save_dirty_beam",
                     "window_type", "window_size", "window_scaled")
    telescope = None

    def setup(self, manager=None):
        if manager is not None:
            self.telescope = io.get_telescope(manager)
        elif self.window_type != "none":
            raise RuntimeError("Must provide manager object if applying window.")
        else:
            self.telescope = None
        excl = list(self.exclude_cyl or [])
        if self.exclude_intracyl:
            excl.append(0)
        self.exclude_cyl = sorted(set(excl))

    # ---- hooks of the reference (host side, tiny)
    def _ew_table(self, n_ew):
        """``(weight_mode, table[n_ew])`` for the kernel, see ``dmm_ringmap_deconvolve``."""
        raise NotImplementedError(f"{self.__class__} must define a _get_weight method.")

    def _get_regularisation(self, freq, m):
        raise NotImplementedError(f"{self.__class__} must define a _get_regularisation method.")

    def _get_window(self, hybrid_vis_m, freq):
        """Window over (freq, m, el) shaping the EW synthesized beam (``ringmapmaker.py:842-930``), on the device.

        The per-(freq, el) limits are a handful of numbers (host, float64, exactly the reference's expressions); the
        ``[nfreq, nm, nel]`` table itself is filled by ``dmm_ringmap_window``.
        """
        import ctypes as C

        ctx = Context.get()
        nm = len(hybrid_vis_m.index_map["m"])
        el = np.asarray(hybrid_vis_m.index_map["el"], dtype=np.float64)
        ew = np.array([x for i, x in enumerate(hybrid_vis_m.index_map["ew"]) if i not in self.exclude_cyl], dtype=np.float64)
        nlocal = len(freq)
        dec = np.arcsin(el[np.newaxis, :]) + np.radians(self.telescope.latitude)
        lmbda = scipy.constants.c / (np.asarray(freq, dtype=np.float64)[:, np.newaxis] * 1e6)
        ews = np.sort(np.abs(ew))
        max_ew = ews[-1] + 0.5 * (ews[-1] - ews[-2])
        min_ew = 0.5 * ews[ews > 0.0][0] if np.min(ews) > 0.0 else -max_ew
        center, width = 0.5 * (min_ew + max_ew), self.window_size * (max_ew - min_ew)
        ew_to_m = 2.0 * np.pi * np.abs(np.cos(dec)) / lmbda
        min_m, max_m = ew_to_m * (center - 0.5 * width), ew_to_m * (center + 0.5 * width)
        if self.window_scaled:
            min_m = np.repeat(np.max(min_m, axis=0, keepdims=True), nlocal, axis=0)
            max_m = np.repeat(np.min(max_m, axis=0, keepdims=True), nlocal, axis=0)
        lo = ctx.to_device(np.ascontiguousarray(min_m), np.float64)
        hi = ctx.to_device(np.ascontiguousarray(max_m), np.float64)
        window = ctx.empty((nlocal, nm, len(el)), np.float32)
        coef = (C.c_double * 4)(*[float(a) for a in _WINDOW_COEF[self.window_type]])
        _lib.check(_lib.lib.dmm_ringmap_window(ctx.handle, int(nlocal), int(nm), int(len(el)), ptr(lo), ptr(hi), coef, ptr(window)))
        return window

    def process(self, hybrid_vis_m, hybrid_beam_m):
        """Generate a deconvolved ringmap using an input beam model (``ringmapmaker.py:627-823``)."""
        if not np.array_equal(hybrid_vis_m.freq, hybrid_beam_m.freq):
            raise ValueError("Frequencies do not match for beam and visibilities.")
        for ax, name in (("el", "Elevations"), ("ew", "EW baselines"), ("pol", "Polarisations")):
            if not np.array_equal(hybrid_vis_m.index_map[ax], hybrid_beam_m.index_map[ax]):
                raise ValueError(f"{name} do not match for beam and visibilities.")
        if hybrid_vis_m.mmax > hybrid_beam_m.mmax:
            raise ValueError("Beam model must have higher m-max than the visibilities")

        hybrid_vis_m.redistribute("freq")
        hybrid_beam_m.redistribute("freq")
        ctx = Context.get()
        freq = np.asarray(hybrid_vis_m.freq, dtype=np.float64)
        m = np.asarray(hybrid_vis_m.index_map["m"])
        mmax = hybrid_vis_m.mmax
        nra = 2 * mmax + int(hybrid_vis_m.oddra)

        hv = _dev_dataset(hybrid_vis_m.vis, ctx, np.complex64)
        hw = _dev_dataset(hybrid_vis_m.weight, ctx, np.float32)
        bv = _dev_dataset(hybrid_beam_m.vis, ctx, np.complex64)
        nm, _, npol, nfreq, n_ew, nel = hv.shape

        rm = containers.RingMap(beam=1, ra=nra, axes_from=hybrid_vis_m, attrs_from=hybrid_vis_m, comm=hybrid_vis_m.comm, allocate=False)
        rm.add_dataset("dirty_beam_power")
        if self.save_dirty_beam:
            rm.add_dataset("dirty_beam")
        rm.attrs["exclude_cyl"] = list(self.exclude_cyl)
        if hasattr(self, "weight_ew"):
            rm.attrs["weight_ew"] = self.weight_ew

        window = None
        if self.window_type != "none":
            window = self._get_window(hybrid_vis_m, freq)
        iref = 0
        if self.skip_deconvolution:
            el = np.asarray(rm.index_map["el"], dtype=np.float64)
            if self.reference_declination is None:
                iref = int(np.argmin(np.abs(el)))
            else:
                dec = np.degrees(np.arcsin(el)) + self.telescope.latitude
                iref = int(np.argmin(np.abs(dec - self.reference_declination)))
        mode, table = self._ew_table(n_ew)
        eps = np.empty((nfreq, nm), dtype=np.float64)
        for fi, f in enumerate(freq):
            eps[fi] = 1.0 if self.skip_deconvolution else np.broadcast_to(np.asarray(self._get_regularisation(f, m), dtype=np.float64).reshape(-1), (nm,))
        table_d = ctx.to_device(np.asarray(table, dtype=np.float64), np.float64)
        eps_d = ctx.to_device(eps, np.float64)

        rmap = ctx.empty((1, npol, nfreq, nra, nel), np.float64)
        rwgt = ctx.empty((npol, nfreq, nra, nel), np.float64)
        rdbp = ctx.empty((1, npol, nfreq, nel), np.float64)
        rdb = ctx.empty((1, npol, nfreq, nra, nel), np.float64) if self.save_dirty_beam else None
        _lib.check(
            _lib.lib.dmm_ringmap_deconvolve(
                ctx.handle, int(nm), int(bv.shape[0]), int(npol), int(nfreq), int(n_ew), int(nel), int(nra), int(mode),
                int(bool(self.skip_deconvolution)), int(iref), ptr(hv), ptr(hw), ptr(bv), ptr(table_d), ptr(eps_d), ptr(window),
                ptr(rmap), ptr(rwgt), ptr(rdbp), ptr(rdb),
            )
        )
        rm.attach("map", rmap)
        rm.attach("weight", rwgt)
        rm.attach("dirty_beam_power", rdbp)
        if self.save_dirty_beam:
            rm.attach("dirty_beam", rdb)
        return rm


class TikhonovRingMapMaker(DeconvolveHybridMBase):
    """Ring maps with a Tikhonov regularisation (``ringmapmaker.py:1075-1121``).

    Attributes
    ----------
    weight_ew : {"natural", "uniform", "inverse_variance"}
        How to weight the EW baselines.
    inv_SN : float
        Regularisation parameter.
    """

    weight_ew = "natural"
    inv_SN = 1e-6
    _config_names = ("weight_ew", "inv_SN")

    def _ew_table(self, n_ew):
        keep = np.ones(n_ew)
        for cyl in self.exclude_cyl:
            keep[cyl] = 0.0
        if self.weight_ew == "inverse_variance":
            return 1, keep
        if self.weight_ew not in ("natural", "uniform"):
            raise ValueError(f"unknown weight_ew {self.weight_ew!r}")
        w = np.ones(n_ew) if self.weight_ew == "uniform" else (n_ew - np.arange(n_ew)).astype(np.float64)
        w = w * keep
        return 0, w * tools.invert_no_zero(w.sum())

    def _get_regularisation(self, *args):
        return self.inv_SN


class WienerRingMapMaker(DeconvolveHybridMBase):
    r"""Ring maps with a Wiener regularisation: noise-to-signal from a power-law sky model
    (``ringmapmaker.py:1124-1183``), inverse-variance EW weights.

    Attributes: ``gal_amp``, ``gal_alpha``, ``gal_beta``, ``psrc_amp``, ``psrc_alpha``.
    """

    gal_amp = 1.41
    gal_alpha = -1.75
    gal_beta = -0.75
    psrc_amp = 0.045
    psrc_alpha = -1.0
    _config_names = ("gal_amp", "gal_alpha", "gal_beta", "psrc_amp", "psrc_alpha")
    pivot_freq = 600.0
    weight_ew = "inverse_variance"

    def _get_regularisation(self, freq, m, *args):
        m = np.asarray(m, dtype=np.float64)
        gal = self.gal_amp * (freq / self.pivot_freq) ** self.gal_alpha * np.where(m > 0.0, m, 1.0) ** self.gal_beta
        psrc = self.psrc_amp * (freq / self.pivot_freq) ** self.psrc_alpha
        return tools.invert_no_zero(gal**2 + psrc**2)

    def _ew_table(self, n_ew):
        keep = np.ones(n_ew)
        for cyl in self.exclude_cyl:
            keep[cyl] = 0.0
        return 2, keep


class DeconvolveAnalyticalBeam(DeconvolveHybridMBase):
    """Deconvolve the analytic (Gaussian-on-the-circle) transit beam model (``ringmapmaker.py:968-1072``);
    non-functional on its own, like the reference's."""

    # EW voltage-beam widths: sigma = coef / freq[MHz] / cos(dec), per feed polarisation (:1009-1017)
    _beam_coef = {"X": 14.87857614, "Y": 9.95746878}

    def setup(self, telescope):
        self.telescope = io.get_telescope(telescope)
        excl = list(self.exclude_cyl or [])
        if self.exclude_intracyl:
            excl.append(0)
        self.exclude_cyl = sorted(set(excl))

    def process(self, hybrid_vis_m):
        return super().process(hybrid_vis_m, self._get_beam_mmodes(hybrid_vis_m))

    def _get_beam_mmodes(self, hybrid_vis_m):
        """Beam m-modes on the axes of ``hybrid_vis_m`` (``ringmapmaker.py:1004-1072``), device resident."""
        ctx = Context.get()
        mmax = hybrid_vis_m.mmax
        nra = 2 * mmax + int(hybrid_vis_m.oddra)
        dec = np.arcsin(np.asarray(hybrid_vis_m.index_map["el"], dtype=np.float64)) + np.radians(self.telescope.latitude)
        pol = [str(x) for x in hybrid_vis_m.index_map["pol"]]
        try:
            ca = np.array([self._beam_coef[x[0]] for x in pol], dtype=np.float64)
            cb = np.array([self._beam_coef[x[1]] for x in pol], dtype=np.float64)
        except (KeyError, IndexError):
            raise KeyError(f"polarisations must be pairs of 'X'/'Y' feeds, got {pol}") from None
        freq = np.asarray(hybrid_vis_m.freq, dtype=np.float64)
        ew = np.asarray(hybrid_vis_m.index_map["ew"], dtype=np.float64)
        out = ctx.empty((mmax + 1, 2, len(pol), len(freq), len(ew), len(dec)), np.complex64)
        tabs = [ctx.to_device(x, np.float64) for x in (freq, ew, dec, ca, cb)]  # held until the launch is queued
        _lib.check(
            _lib.lib.dmm_analytic_beam_mmodes(
                ctx.handle, len(pol), len(freq), len(ew), len(dec), int(nra), int(mmax), *(ptr(t) for t in tabs), ptr(out)
            )
        )
        beam = containers.HybridVisMModes(mmax=mmax, oddra=hybrid_vis_m.oddra, axes_from=hybrid_vis_m, attrs_from=hybrid_vis_m,
                                          comm=hybrid_vis_m.comm, allocate=False)
        beam.attach("vis", out)
        return beam


class TikhonovRingMapMakerAnalytical(DeconvolveAnalyticalBeam, TikhonovRingMapMaker):
    """Tikhonov deconvolution of the analytical beam model (``ringmapmaker.py:1189``)."""


class WienerRingMapMakerAnalytical(DeconvolveAnalyticalBeam, WienerRingMapMaker):
    """Wiener deconvolution of the analytical beam model (``ringmapmaker.py:1190``)."""


# Aliases to support old names (ringmapmaker.py:1193-1194)
TikhonovRingMapMakerExternal = TikhonovRingMapMaker
WienerRingMapMakerExternal = WienerRingMapMaker

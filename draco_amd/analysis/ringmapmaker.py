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


# ------------------------------------------------------------------------------------------------------------------
# The ring-map chain MakeVisGrid -> BeamformNS -> BeamformEW (``ringmapmaker.py:38-534``): BeamformNS's output is what
# MModeTransform turns into the HybridVisMModes the deconvolving makers above consume; BeamformEW makes the plain ring map.
def window_generalised_full(x, window="nuttall"):
    """``tools.window_generalised`` with every name it knows (``util/tools.py:547-601``)."""
    x = np.asarray(x, dtype=np.float64)
    if window == "triangular":
        w = 1.0 - 2.0 * np.abs(x - 0.5)
    elif window.startswith("tukey"):
        alpha = 0.5 * float(window.split("-")[1])
        w = np.ones_like(x)
        lo = x < alpha
        w[lo] = 0.5 * (1.0 + np.cos(np.pi * (x[lo] - alpha) / alpha))
        hi = x >= 1.0 - alpha
        w[hi] = 0.5 * (1.0 + np.cos(np.pi * (x[hi] - (1.0 - alpha)) / alpha))
    else:
        return window_generalised(x, window)
    return np.where((x >= 0) & (x <= 1), w, 0)


def find_basis(baselines):
    """Unit vectors of the baseline grid (``ringmapmaker.py:1715-1742``): the shortest baseline gives one axis."""
    baselines = np.asarray(baselines, dtype=np.float64)
    length2 = np.sum(baselines**2, axis=1)
    length2[length2 == 0] = 1e30
    first = baselines[np.argmin(length2)]
    second = np.array([first[1], -first[0]])
    xh, yh = (first, second) if abs(first[0]) > abs(second[0]) else (second, first)
    return xh / np.dot(xh, xh) ** 0.5 * np.sign(xh[0]), yh / np.dot(yh, yh) ** 0.5 * np.sign(yh[1])


def find_grid_indices(baselines):
    """Grid index of every baseline and the grid spacings (``ringmapmaker.py:1745-1771``)."""
    baselines = np.asarray(baselines, dtype=np.float64)
    xh, yh = find_basis(baselines)
    out = []
    for proj in (baselines @ xh, baselines @ yh):
        mag = np.abs(proj)
        step = mag[mag > 1e-4].min()
        out.append((np.rint(proj / step).astype(np.int64), step))
    return out[0][0], out[1][0], out[0][1], out[1][1]


class MakeVisGrid(ContainerTask):
    """Arrange the visibilities onto a 2D grid, the half plane ``x >= 0`` of the EW separation (``ringmapmaker.py:38-176``).

    Attributes
    ----------
    centered : bool
        Zero NS separation at the centre of the y axis (ascending order) instead of at position zero (FFT order).
    save_redundancy : bool
        Compute and store the redundancy of each visibility.
    """

    centered = False
    save_redundancy = True
    _config_names = ("centered", "save_redundancy")
    telescope = None

    def setup(self, tel):
        self.telescope = io.get_telescope(tel)

    def process(self, sstream):
        tel = self.telescope
        ps_stream = np.stack([np.asarray(sstream.prodstack["input_a"], dtype=np.int16), np.asarray(sstream.prodstack["input_b"], dtype=np.int16)], axis=1)
        ps_tel = np.stack([np.asarray(tel.prodstack["input_a"], dtype=np.int16), np.asarray(tel.prodstack["input_b"], dtype=np.int16)], axis=1)
        if not np.array_equal(ps_stream, ps_tel):
            raise ValueError("Products in sstream do not match those in the beam transfers.")
        # the pairs of the stack entries as the data hold them (conjugated representatives swapped): for a driftscan
        # telescope these ARE `uniquepairs` (:81)
        polprod = np.asarray(tel.polarisation)[ps_tel.astype(np.int64)]
        pol, pind = np.unique(np.char.add(polprod[:, 0], polprod[:, 1]), return_inverse=True)
        if len(pol) != 4:
            raise RuntimeError(f"Expected to find four polarisations. Got {pol}")
        pconjmap = np.unique([b + a for a, b in pol], return_inverse=True)[1]
        xind, yind, min_xsep, min_ysep = find_grid_indices(tel.baselines)
        nx = int(np.abs(xind).max()) + 1
        max_yind = int(np.abs(yind).max())
        ny = 2 * max_yind + 1
        vis_pos_x = np.arange(nx) * min_xsep
        if self.centered:
            vis_pos_y, ns_offset = np.arange(-max_yind, max_yind + 1) * min_ysep, max_yind
        else:
            vis_pos_y, ns_offset = np.fft.fftfreq(ny, d=(1.0 / (ny * min_ysep))), 0
        ra = np.asarray(sstream.index_map["ra"])
        grid = containers.VisGridStream(pol=pol, ew=vis_pos_x, ns=vis_pos_y, ra=ra, axes_from=sstream, attrs_from=sstream, comm=sstream.comm, allocate=False)
        # the reference's scatter loop (:166-176), inverted: which stack (conjugated?) lands in each grid cell; later
        # baselines overwrite earlier ones exactly as the loop's assignments do
        ncell_pol = nx * ny
        src = np.full(4 * ncell_pol, -1, dtype=np.int32)
        conj = np.zeros(4 * ncell_pol, dtype=np.uint8)
        for vi, (p_, x_, y_) in enumerate(zip(pind, xind, yind)):
            # (negative grid indices wrap round like the reference's NumPy indexing does: x = -1 is the last EW slot)
            cell = (p_ * nx + x_ % nx) * ny + (ns_offset + y_) % ny
            src[cell], conj[cell] = vi, 0
            if x_ == 0:
                cellc = (pconjmap[p_] * nx + x_) * ny + (ns_offset - y_) % ny
                src[cellc], conj[cellc] = vi, 1
        sstream.redistribute("freq")
        ctx = Context.get()
        ssv = _dev_dataset(sstream.vis, ctx, np.complex64)
        ssw = _dev_dataset(sstream.weight, ctx, np.float32)
        nfreq, nstack, nra = ssv.shape
        red_d = None
        gr = None
        if self.save_redundancy:
            flags = np.asarray(sstream.input_flags[:], dtype=np.float32)
            prod = sstream.index_map["prod"]
            stack_index = np.asarray(sstream.reverse_map["stack"]["stack"], dtype=np.int32)
            red_d = ctx.empty((nstack, nra), np.float32)
            f_d = ctx.to_device(flags, np.float32)
            pa = ctx.to_device(np.asarray(prod["input_a"], dtype=np.int32))
            pb = ctx.to_device(np.asarray(prod["input_b"], dtype=np.int32))
            st = ctx.to_device(stack_index)
            _lib.check(_lib.lib.dmm_calc_redundancy(ctx.handle, ptr(f_d), int(flags.shape[0]), int(nra), ptr(pa), ptr(pb), ptr(st), int(len(stack_index)),
                                                    int(nstack), int(not np.any(flags)), ptr(red_d)))
            grid.add_dataset("redundancy")
            gr = ctx.empty((4, nx, ny, nra), np.int32)
        gv = ctx.empty((4, nfreq, nx, ny, nra), np.complex64)
        gw = ctx.empty((4, nfreq, nx, ny, nra), np.float32)
        src_d, conj_d = ctx.to_device(src), ctx.to_device(conj)
        _lib.check(_lib.lib.dmm_vis_grid(ctx.handle, ptr(ssv), ptr(ssw), ptr(red_d), int(nfreq), int(nstack), int(nra), 4, int(ncell_pol), ptr(src_d), ptr(conj_d),
                                         ptr(gv), ptr(gw), ptr(gr)))
        ctx.sync()  # (the index tables go out of scope)
        grid.attach("vis", gv)
        grid.attach("vis_weight", gw)
        if gr is not None:
            grid.attach("redundancy", gr)
        return grid


class BeamformNS(ContainerTask):
    """Form a series of beams on the meridian from the gridded visibilities (``ringmapmaker.py:179-346``).

    Attributes
    ----------
    npix : int
        Number of map pixels in the declination dimension.
    span : float
        Span of the map in sin(za): 1.0 is horizon to horizon.
    weight : str
        'natural' (by redundancy), 'inverse_variance', 'uniform', or any window of ``window_generalised``.
    scaled : bool
        Scale the window to match the lowest frequency.
    include_auto : bool
        Include auto-correlations.
    save_dirty_beam : bool
        Also compute the dirty beam.
    precision : int
        Accepted for compatibility (32 / 64): the GPU always applies the beamforming matrix in float64.
    """

    npix = 512
    span = 1.0
    weight = "natural"
    scaled = False
    include_auto = False
    save_dirty_beam = False
    precision = 64
    _config_names = ("npix", "span", "weight", "scaled", "include_auto", "save_dirty_beam", "precision")

    def process(self, gstream):
        if int(self.precision) not in (32, 64):
            raise ValueError(f"precision must be 32 or 64, got {self.precision}")
        gstream.redistribute("freq")
        if self.weight == "natural" and "redundancy" not in gstream.datasets:
            raise RuntimeError("Must set save_redundancy = True for task MakeVisGrid in order to use a natural weight scheme.")
        ctx = Context.get()
        gsv = _dev_dataset(gstream.vis, ctx, np.complex64)
        gsw = _dev_dataset(gstream.weight, ctx, np.float32)
        npol, nfreq, nx, ny, nra = gsv.shape
        el = self.span * np.linspace(-1.0, 1.0, int(self.npix))
        hv = containers.HybridVisStream(el=el, axes_from=gstream, attrs_from=gstream, comm=gstream.comm, allocate=False)
        nspos = np.asarray(gstream.index_map["ns"], dtype=np.float64)
        freq = np.asarray(gstream.freq, dtype=np.float64)
        # the largest NS baseline present, masking accounted for (:276-283; one rank: the all-reduce is the identity)
        present = (gsw > 0).any(dim=4).any(dim=2).any(dim=1).any(dim=0).cpu().numpy()
        nsmax = float(np.abs(nspos[present]).max()) if present.sum() > 0 else 0.0
        hv.attrs["beamform_ns_weight"] = self.weight
        hv.attrs["beamform_ns_scaled"] = self.scaled
        hv.attrs["beamform_ns_include_auto"] = self.include_auto
        hv.attrs["beamform_ns_freqmin"] = freq.min()
        hv.attrs["beamform_ns_nsmax"] = nsmax
        iwv = freq * 1e6 / scipy.constants.c
        mode, table, red_d = 0, None, None
        if self.weight == "natural":
            mode = 1
            red_d = gstream.redundancy.device(ctx)
        elif self.weight != "inverse_variance":
            mode = 2
            rows = []
            for fi in range(nfreq):
                vmax = nsmax * (iwv.min() if self.scaled else iwv[fi])
                with np.errstate(divide="ignore", invalid="ignore"):
                    x = 0.5 * (nspos * iwv[fi] / vmax + 1)
                rows.append(window_generalised_full(x, window=self.weight))
            table = ctx.to_device(np.asarray(rows, dtype=np.float64))
        hvv = ctx.empty((npol, nfreq, nx, len(el), nra), np.complex64)
        hvw = ctx.empty((npol, nfreq, nx, nra), np.float32)
        hvb = ctx.empty((npol, nfreq, nx, len(el), nra), np.float32) if self.save_dirty_beam else None
        ns_d, el_d = ctx.to_device(nspos), ctx.to_device(el)
        iwv_h = np.ascontiguousarray(iwv, dtype=np.float64)
        import ctypes as C

        _lib.check(_lib.lib.dmm_beamform_ns(ctx.handle, int(npol), int(nfreq), int(nx), int(ny), int(nra), int(len(el)), mode, int(bool(self.include_auto)),
                                            ptr(gsv), ptr(gsw), ptr(red_d), ptr(table), ptr(ns_d), ptr(el_d), C.c_void_p(iwv_h.ctypes.data), ptr(hvv), ptr(hvw), ptr(hvb)))
        ctx.sync()
        hv.attach("vis", hvv)
        hv.attach("vis_weight", hvw)
        if hvb is not None:
            hv.add_dataset("dirty_beam")
            hv.attach("dirty_beam", hvb)
        return hv


class BeamformEW(ContainerTask):
    """Final beam forming in the EW direction (``ringmapmaker.py:349-530``).

    Attributes
    ----------
    exclude_intracyl : bool
        Exclude intracylinder baselines.
    single_beam : bool
        Only the central beam.
    weight_ew : str
        'natural' (by the redundancy of the EW baselines) or 'uniform'.
    flag_ew : array of bool, optional
        Which EW baselines to include.

    Where the input carries a dirty beam it is beamformed like the data (the reference's own line for it,
    ``ringmapmaker.py:489``, fails to broadcast: ``tests/golden/ringmap_chain.npz::ew_db_error``).
    """

    exclude_intracyl = False
    single_beam = False
    weight_ew = "natural"
    flag_ew = None
    _config_names = ("exclude_intracyl", "single_beam", "weight_ew", "flag_ew")

    @staticmethod
    def _get_pol(pols):
        """Output polarisations and the rotation from the XY / YX basis into reXY / imXY (``ringmapmaker.py:499-530``)."""
        pols = [str(x) for x in pols]
        dpol = []
        if ("XY" in pols) or ("YX" in pols):
            if ("XY" in pols) ^ ("YX" in pols):
                raise ValueError(f"If cross-pols exist, both XY and YX must be present. Got {pols}.")
            dpol = ["reXY", "imXY"]
        if "XX" in pols:
            dpol = ["XX", *dpol]
        if "YY" in pols:
            dpol.append("YY")
        rot = np.eye(len(dpol), dtype=np.complex64)
        if "reXY" in dpol:
            i = dpol.index("reXY")
            rot[i, i : i + 2] = [0.5, 0.5]
            rot[i + 1, i : i + 2] = [-0.5j, 0.5j]
        return np.array(dpol, dtype="U4"), rot

    def process(self, hstream):
        if self.weight_ew not in ("natural", "uniform"):
            raise ValueError(f"weight_ew must be 'natural' or 'uniform', got {self.weight_ew!r}")
        hstream.redistribute("freq")
        n_ew = len(hstream.index_map["ew"])
        nbeam = 1 if self.single_beam else 2 * n_ew - 1
        w = np.ones(n_ew) if self.weight_ew == "uniform" else (n_ew - np.arange(n_ew)).astype(np.float64)
        if self.exclude_intracyl:
            w[0] = 0.0
        if self.flag_ew is not None and np.size(self.flag_ew) == n_ew:
            w = w * np.asarray(self.flag_ew).astype(bool).astype(w.dtype)
        if self.single_beam:
            w[1:] *= 2
        w = w / w.sum()
        pol, rot = self._get_pol(hstream.index_map["pol"])
        rm = containers.RingMap(beam=nbeam, pol=pol, axes_from=hstream, attrs_from=hstream, comm=hstream.comm, allocate=False)
        rm.add_dataset("rms")
        ctx = Context.get()
        hvv = _dev_dataset(hstream.vis, ctx, np.complex64)
        hvw = _dev_dataset(hstream.weight, ctx, np.float32)
        npol_in, nfreq, _, nel, nra = hvv.shape
        npol_out = len(pol)
        has_db = "dirty_beam" in hstream.datasets
        rmm = ctx.empty((nbeam, npol_out, nfreq, nra, nel), np.float64)
        rmw = ctx.empty((npol_out, nfreq, nra, nel), np.float64)
        rmr = ctx.empty((npol_out, nfreq, nra), np.float64)
        rmb = ctx.empty((nbeam, npol_out, nfreq, nra, nel), np.float64) if has_db else None
        hvb = hstream.dirty_beam.device(ctx) if has_db else None
        rot_d = ctx.to_device(np.ascontiguousarray(rot, dtype=np.complex128))
        w_d = ctx.to_device(np.ascontiguousarray(w, dtype=np.float64))
        _lib.check(_lib.lib.dmm_beamform_ew(ctx.handle, int(npol_in), int(npol_out), int(nfreq), int(n_ew), int(nel), int(nra), int(bool(self.single_beam)),
                                            ptr(hvv), ptr(hvw), ptr(hvb), ptr(rot_d), ptr(w_d), ptr(rmm), ptr(rmw), ptr(rmr), ptr(rmb)))
        ctx.sync()
        rm.attach("map", rmm)
        rm.attach("weight", rmw)
        rm.attach("rms", rmr)
        if has_db:
            rm.add_dataset("dirty_beam")
            rm.attach("dirty_beam", rmb)
        return rm


class RingMapMaker(ContainerTask):
    """Make a ring map from the data: ``MakeVisGrid -> BeamformNS -> BeamformEW`` as one task
    (``ringmapmaker.py:533-534``: ``group_tasks`` of the three; their config attributes are accepted here)."""

    _stages = (MakeVisGrid, BeamformNS, BeamformEW)
    _config_names = tuple(n for c in _stages for n in c._config_names)

    def __init__(self, **kw):
        self._tasks = [c(**{k: v for k, v in kw.items() if k in c._config_names}) for c in self._stages]
        unknown = set(kw) - set(self._config_names)
        if unknown:
            raise TypeError(f"unknown config attributes {sorted(unknown)}")

    def setup(self, tel):
        self._tasks[0].setup(tel)

    def process(self, sstream):
        out = sstream
        for t in self._tasks:
            out = t.process(out)
        return out

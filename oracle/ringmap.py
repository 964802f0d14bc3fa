"""Oracle: deconvolving ring-map makers.  TEST INFRASTRUCTURE ONLY.

Restates ``DeconvolveHybridMBase.process`` with the ``TikhonovRingMapMaker`` /
``WienerRingMapMaker`` weights and regularisation (reference
``draco/analysis/ringmapmaker.py:538-823, 842-930, 1075-1186``; window shapes from
``draco/util/tools.py:547-601``) on plain arrays:

    hv [nm, 2, npol, nfreq, new, nel] complex64   hybrid visibility m-modes
    hw [nm, 2, npol, nfreq, new]      float32     their inverse variance
    bv [>=nm, 2, npol, nfreq, new, nel] complex64 beam m-modes

-> map [1, npol, nfreq, nra, nel], weight [npol, nfreq, nra, nel], dirty_beam_power
[1, npol, nfreq, nel], dirty_beam [1, npol, nfreq, nra, nel], all float64.
Pinned by ``tests/golden/ringmap_deconvolve.npz`` (outputs of the reference classes).
"""

from __future__ import annotations

import numpy as np
import scipy.constants

from .transform import invert_no_zero

_WINDOWS = {
    "uniform": [1, 0, 0, 0],
    "hann": [0.5, -0.5, 0, 0],
    "hanning": [0.5, -0.5, 0, 0],
    "hamming": [0.53836, -0.46164, 0, 0],
    "blackman": [0.42, -0.5, 0.08, 0],
    "nuttall": [0.355768, -0.487396, 0.144232, -0.012604],
    "blackman_nuttall": [0.3635819, -0.4891775, 0.1365995, -0.0106411],
    "blackman_harris": [0.35875, -0.48829, 0.14128, -0.01168],
}


def window_generalised(x, window="nuttall"):
    """Cosine-sum windows at arbitrary locations, zero outside [0, 1] (``tools.py:583-601``)."""
    a = np.array(_WINDOWS[window])
    t = 2 * np.pi * np.arange(4)[:, np.newaxis] * x[np.newaxis, :]
    w = (a[:, np.newaxis] * np.cos(t)).sum(axis=0)
    return np.where((x >= 0) & (x <= 1), w, 0)


def get_window(freq, m, el, ew, latitude, window_type, window_size=1.0, window_scaled=False, exclude_cyl=()):
    """``_get_window`` (``ringmapmaker.py:842-930``) -> float32 ``[nfreq, nm, nel]``."""
    ew = np.array([x for i, x in enumerate(ew) if i not in exclude_cyl])
    nlocal = len(freq)
    dec = np.arcsin(el[np.newaxis, :]) + np.radians(latitude)
    lmbda = scipy.constants.c / (np.asarray(freq)[:, np.newaxis] * 1e6)
    ews = np.abs(ew)[np.argsort(np.abs(ew))]
    max_ew = ews[-1] + 0.5 * (ews[-1] - ews[-2])
    min_ew = 0.5 * ews[ews > 0.0][0] if np.min(ews) > 0.0 else -max_ew
    center = 0.5 * (min_ew + max_ew)
    width = window_size * (max_ew - min_ew)
    ew_to_m = 2.0 * np.pi * np.abs(np.cos(dec)) / lmbda
    min_m = ew_to_m * (center - 0.5 * width)
    max_m = ew_to_m * (center + 0.5 * width)
    if window_scaled:
        min_m = np.max(min_m, axis=0, keepdims=True)
        max_m = np.min(max_m, axis=0, keepdims=True)
    nf, nel = min_m.shape
    window = np.zeros((nf, m.size, nel), dtype=np.float32)
    for ff in range(nf):
        for ee in range(nel):
            lo, hi = min_m[ff, ee], max_m[ff, ee]
            in_range = np.flatnonzero((m >= lo) & (m <= hi))
            if in_range.size > 0:
                x = (m[in_range] - lo) / (hi - lo)
                window[ff, in_range, ee] = window_generalised(x, window=window_type)
    if window_scaled:
        window = np.repeat(window, nlocal, axis=0)
    return window


def ew_weight(kind, weight_ew, inv_var, exclude_cyl):
    """``_get_weight`` of the two makers (``ringmapmaker.py:1096-1118, 1178-1183``); may modify ``inv_var``."""
    if kind == "wiener":
        w = inv_var
        for cyl in exclude_cyl:
            w[..., cyl, :] = 0.0
        return w
    if weight_ew == "inverse_variance":
        w = inv_var
    else:
        n_ew = inv_var.shape[-2]
        w = np.ones(n_ew) if weight_ew == "uniform" else n_ew - np.arange(n_ew)
        expand = [None] * inv_var.ndim
        expand[-2] = slice(None)
        w = w[tuple(expand)]
    for cyl in exclude_cyl:
        w[..., cyl, :] = 0.0
    return w * invert_no_zero(np.sum(w, axis=-2, keepdims=True))


def regularisation(kind, freq, m, inv_SN=1e-6, gal_amp=1.41, gal_alpha=-1.75, gal_beta=-0.75, psrc_amp=0.045, psrc_alpha=-1.0, pivot=600.0):
    """``_get_regularisation`` (``ringmapmaker.py:1120-1121, 1161-1176``)."""
    if kind == "tikhonov":
        return inv_SN
    gal = gal_amp * (freq / pivot) ** gal_alpha * np.where(m > 0.0, m, 1.0) ** gal_beta
    psrc = psrc_amp * (freq / pivot) ** psrc_alpha
    spectrum = gal**2 + psrc**2
    return invert_no_zero(spectrum[:, np.newaxis, np.newaxis])


def deconvolve(kind, hv, hw, bv, freq, el, ew, oddra, exclude_cyl=(), skip_deconvolution=False, window=None, weight_ew="natural", iref=None, **reg):
    """The per-frequency loop of ``DeconvolveHybridMBase.process`` (``ringmapmaker.py:683-823``).

    ``window``: None (``window_type == "none"``) or float32 ``[nfreq, nm, nel]`` from :func:`get_window`.
    """
    nm, _, npol, nfreq, new, nel = hv.shape
    mmax = nm - 1
    m = np.arange(nm)
    nra = 2 * mmax + int(oddra)
    bv = bv[: mmax + 1]
    rmm = np.zeros((1, npol, nfreq, nra, nel))
    rmw = np.zeros((npol, nfreq, nra, nel))
    rmbp = np.zeros((1, npol, nfreq, nel))
    rmb = np.zeros((1, npol, nfreq, nra, nel))
    if window is not None:
        window = window[:, :, np.newaxis, :]
    else:
        window = np.ones(nfreq, dtype=np.float32)
    if skip_deconvolution and iref is None:
        iref = np.argmin(np.abs(el))
    for lfi, f in enumerate(freq):
        find = (slice(None),) * 3 + (lfi,)
        hvf, bvf = hv[find], bv[find]
        winf = window[lfi]
        inv_var = hw[find][..., np.newaxis].copy()
        weight = ew_weight(kind, weight_ew, inv_var, exclude_cyl) * (inv_var > 0.0)
        sum_weight = (weight * np.abs(bvf) ** 2).sum(axis=(1, -2))
        if not skip_deconvolution:
            C_inv = regularisation(kind, f, m, **reg) + sum_weight
        else:
            C_inv = 1.0
        map_m = winf * (bvf.conj() * weight * hvf).sum(axis=(1, -2)) * invert_no_zero(C_inv)
        dirty_beam_m = winf * sum_weight * invert_no_zero(C_inv)
        norm = invert_no_zero(dirty_beam_m.mean(axis=0))[:, np.newaxis, :]
        if skip_deconvolution:
            norm = norm[:, :, iref, np.newaxis]
        rmm[0, :, lfi] = np.fft.irfft(map_m.transpose(1, 2, 0), axis=-1, n=nra).transpose(0, 2, 1) * norm
        dirty_beam_ra = np.fft.irfft(dirty_beam_m.transpose(1, 2, 0), axis=-1, n=nra).transpose(0, 2, 1) * norm
        rmbp[0, :, lfi] = np.sum(dirty_beam_ra**2, axis=1) / nra
        rmb[0, :, lfi] = dirty_beam_ra
        var = invert_no_zero(inv_var)
        sigma = np.sqrt(np.sum((weight * np.abs(bvf)) ** 2 * var, axis=(1, -2)))
        sum_var_map_m = 0.5 * np.sum((sigma * winf * norm[np.newaxis, :, 0] * invert_no_zero((mmax + 1) * C_inv)) ** 2, axis=0)[:, np.newaxis, :]
        rmw[:, lfi] = invert_no_zero(sum_var_map_m)
    return rmm, rmw, rmbp, rmb


# EW voltage-beam widths (sigma, radians) per feed polarisation: sigma = coef / freq[MHz] / cos(dec)
# (ringmapmaker.py:1009-1017)
_BEAM_COEF = {"X": 14.87857614, "Y": 9.95746878}


def analytic_beam_mmodes(freq, ew, el, pol, latitude, mmax, oddra):
    """``DeconvolveAnalyticalBeam._get_beam_mmodes`` (ringmapmaker.py:1004-1072) on plain arrays.

    Returns the beam m-modes ``[mmax+1, 2, npol, nfreq, new, nel]`` complex64: for every (pol, freq, ew, el) the
    transit ``conj(exp(2 pi i u cos(dec) sin(phi)) * exp(-(2 tan(phi/2))^2 / (2 sigma^2)))`` over
    ``phi = 2 pi k / nra``, ``nra = 2 mmax + oddra``, transformed with ``_make_marray`` (complex128 FFT, stored
    complex64).  Pinned by ``tests/golden/ringmap_analytic.npz``.
    """
    from .transform import make_marray

    freq = np.asarray(freq, dtype=np.float64)
    ew = np.asarray(ew, dtype=np.float64)
    nra = 2 * int(mmax) + int(oddra)
    dec = np.arcsin(np.asarray(el, dtype=np.float64)) + np.radians(latitude)
    phi = np.radians(np.linspace(0.0, 360.0, nra, endpoint=False))
    out = np.zeros((mmax + 1, 2, len(pol), len(freq), len(ew), len(dec)), dtype=np.complex64)
    for fi, f in enumerate(freq):
        u = ew / (scipy.constants.c * 1e-6 / f)
        u_dec = u[:, None] * np.cos(dec)[None, :]
        sig = np.zeros((len(pol), len(dec)))
        for pi, (pa, pb) in enumerate(pol):
            sa = _BEAM_COEF[pa] / f / np.cos(dec)
            sb = _BEAM_COEF[pb] / f / np.cos(dec)
            sig[pi] = sa * sb / (sa**2 + sb**2) ** 0.5
        amp = np.exp(-((2 * np.tan(phi / 2)) ** 2) / (2 * sig[:, None, :, None] ** 2))
        b = np.exp(2.0j * np.pi * u_dec[None, :, :, None] * np.sin(phi)) * amp
        out[:, :, :, fi] = make_marray(b.conj(), mmax=mmax, dtype=np.complex64)
    return out


# ------------------------------------------------------------------ the ring-map chain: MakeVisGrid -> BeamformNS -> BeamformEW
def window_generalised_full(x, window="nuttall"):
    """``tools.window_generalised`` with every name it knows (``tools.py:547-601``): cosine sums, 'triangular', 'tukey-0.X'."""
    x = np.asarray(x, dtype=float)
    if window == "triangular":
        w = 1.0 - 2.0 * np.abs(x - 0.5)
    elif window.startswith("tukey"):
        alpha = 0.5 * float(window.split("-")[1])
        w = np.ones_like(x)
        b = x < alpha
        w[b] = 0.5 * (1.0 + np.cos(np.pi * (x[b] - alpha) / alpha))
        e = x >= 1.0 - alpha
        w[e] = 0.5 * (1.0 + np.cos(np.pi * (x[e] - (1.0 - alpha)) / alpha))
    else:
        return window_generalised(x, window)
    return np.where((x >= 0) & (x <= 1), w, 0)


def find_grid_indices(baselines):
    """``find_basis`` + ``find_grid_indices`` (``ringmapmaker.py:1715-1771``): grid index of every baseline, grid spacings."""
    bl = np.sum(baselines**2, axis=1)
    bl[bl == 0] = 1e30
    e1 = baselines[np.argmin(bl)]
    e2 = np.array([e1[1], -e1[0]])
    xh, yh = (e1, e2) if abs(e1[0]) > abs(e2[0]) else (e2, e1)
    xh = xh / np.dot(xh, xh) ** 0.5 * np.sign(xh[0])
    yh = yh / np.dot(yh, yh) ** 0.5 * np.sign(yh[1])

    def inds(s):
        sa = np.abs(s)
        d = sa[sa > 1e-4].min()
        return np.rint(s / d).astype(np.int64), d

    xind, dx = inds(baselines @ xh)
    yind, dy = inds(baselines @ yh)
    return xind, yind, dx, dy


def calculate_redundancy(input_flags, prod, stack_index, nstack):
    """``tools.calculate_redundancy`` (``tools.py:313-356``; its compiled inner loop restated): good-input pairs per stack."""
    flags = np.asarray(input_flags, dtype=np.float32)
    if not np.any(flags):
        flags = np.ones_like(flags)
    red = np.zeros((nstack, flags.shape[1]), dtype=np.float32)
    for (a, b), s in zip(prod, stack_index):
        if 0 <= s < nstack:
            red[s] += flags[a] * flags[b]
    return red


def make_vis_grid(vis, weight, uniquepairs, polarisation, baselines, input_flags, prod, stack_index, centered=False):
    """``MakeVisGrid.process`` (``ringmapmaker.py:68-176``): stacked visibilities onto the (pol, ew, ns) grid, the
    intra-cylinder row filled on both sides of ns = 0 (conjugate, swapped polarisation pair)."""
    polpair = np.char.add(polarisation[uniquepairs[:, 0]], polarisation[uniquepairs[:, 1]])
    pol, pind = np.unique(polpair, return_inverse=True)
    if len(pol) != 4:
        raise RuntimeError(f"Expected to find four polarisations. Got {pol}")
    pconjmap = np.unique([pj + pi for pi, pj in pol], return_inverse=True)[1]
    xind, yind, dx, dy = find_grid_indices(np.asarray(baselines, dtype=float))
    nx = np.abs(xind).max() + 1
    max_y = np.abs(yind).max()
    ny = 2 * max_y + 1
    ew = np.arange(nx) * dx
    if centered:
        ns, off = np.arange(-max_y, max_y + 1) * dy, max_y
    else:
        ns, off = np.fft.fftfreq(ny, d=1.0 / (ny * dy)), 0
    nfreq, nstack, nra = vis.shape
    red = calculate_redundancy(input_flags, prod, stack_index, nstack)
    gv = np.zeros((4, nfreq, nx, ny, nra), np.complex64)
    gw = np.zeros((4, nfreq, nx, ny, nra), np.float32)
    gr = np.zeros((4, nx, ny, nra), np.int32)
    for vi, (p, x, y) in enumerate(zip(pind, xind, yind)):
        gv[p, :, x, off + y, :] = vis[:, vi]
        gw[p, :, x, off + y, :] = weight[:, vi]
        gr[p, x, off + y, :] = red[vi]
        if x == 0:
            pc = pconjmap[p]
            gv[pc, :, x, off - y, :] = vis[:, vi].conj()
            gw[pc, :, x, off - y, :] = weight[:, vi]
            gr[pc, x, off - y, :] = red[vi]
    return gv, gw, gr, pol, ew, ns


def beamform_ns(gv, gw, gr, nspos, freq, npix=512, span=1.0, weight="natural", scaled=False, include_auto=False, precision=64):
    """``BeamformNS.process`` (``ringmapmaker.py:230-346``): weights over ns, normalised; the DFT in the NS direction to
    `npix` elevations; noise weights.  Returns (vis c64, weight f32, dirty_beam f32, el, nsmax)."""
    cdt = np.complex128 if precision == 64 else np.complex64
    rdt = np.float64 if precision == 64 else np.float32
    el = span * np.linspace(-1.0, 1.0, npix)
    npol, nfreq, nx, ny, nra = gv.shape
    present = np.any(gw > 0, axis=(0, 1, 2, 4))
    nsmax = np.abs(nspos[present]).max() if present.sum() > 0 else 0.0
    phase = (2.0 * np.pi * nspos[np.newaxis] * el[:, np.newaxis]).astype(cdt)
    hv = np.zeros((npol, nfreq, nx, npix, nra), np.complex64)
    hw = np.zeros((npol, nfreq, nx, nra), np.float32)
    hb = np.zeros((npol, nfreq, nx, npix, nra), np.float32)
    for fi in range(nfreq):
        iwv = freq[fi] * 1e6 / scipy.constants.c
        vpos = nspos * iwv
        vmax = nsmax * (freq.min() * 1e6 / scipy.constants.c if scaled else iwv)
        if weight == "inverse_variance":
            w = gw[:, fi].copy()
        elif weight == "natural":
            w = gr.astype(np.float32)
        else:
            x = 0.5 * (vpos / vmax + 1)
            nsw = window_generalised_full(x, window=weight).astype(rdt)
            w = (gw[:, fi] > 0) * nsw[np.newaxis, np.newaxis, :, np.newaxis]
        w = w * (gw[:, fi] > 0)
        if not include_auto:
            w[..., 0, 0, :] = 0.0
        norm = np.sum(w, axis=-2)
        w = w * invert_no_zero(norm)[..., np.newaxis, :]
        F = np.exp(-1.0j * phase * iwv)
        hv[:, fi] = np.matmul(F, gv[:, fi] * w)
        hb[:, fi] = np.matmul(F, w * np.ones_like(gv[:, fi])).real
        t = np.sum(invert_no_zero(gw[:, fi]) * w**2, axis=-2)
        hw[:, fi] = invert_no_zero(t)
    return hv, hw, hb, el, nsmax


def ew_pol_rotation(pols):
    """``BeamformEW._get_pol`` (``ringmapmaker.py:499-530``)."""
    pols = list(pols)
    if ("XY" in pols) or ("YX" in pols):
        if ("XY" in pols) ^ ("YX" in pols):
            raise ValueError(f"If cross-pols exist, both XY and YX must be present. Got {pols}.")
        dpol = ["reXY", "imXY"]
    else:
        dpol = []
    if "XX" in pols:
        dpol = ["XX", *dpol]
    if "YY" in pols:
        dpol.append("YY")
    P = np.eye(len(dpol), dtype=np.complex64)
    if "reXY" in dpol:
        i = dpol.index("reXY")
        P[i, i : i + 2] = [0.5, 0.5]
        P[i + 1, i : i + 2] = [-0.5j, 0.5j]
    return np.array(dpol, dtype="U4"), P


def beamform_ew(hv, hw, pols, exclude_intracyl=False, single_beam=False, weight_ew="natural", flag_ew=None, dirty_beam=None):
    """``BeamformEW.process`` (``ringmapmaker.py:372-497``): polarisation rotation, EW weights, inverse real FFT over the
    EW baselines into `2 n_ew - 1` beams (or the central one), variance propagation.  The dirty beam is transformed the
    same way (the reference's own line for it, :489, fails to broadcast)."""
    n_ew = hv.shape[2]
    nbeam = 1 if single_beam else 2 * n_ew - 1
    w = np.ones(n_ew) if weight_ew == "uniform" else (n_ew - np.arange(n_ew)).astype(float)
    if exclude_intracyl:
        w[0] = 0.0
    if flag_ew is not None and np.size(flag_ew) == n_ew:
        w = w * np.asarray(flag_ew).astype(bool).astype(w.dtype)
    if single_beam:
        w[1:] *= 2
    w = w / w.sum()
    pol, P = ew_pol_rotation(pols)
    P2 = np.abs(P) ** 2
    npolo, nfreq = len(pol), hv.shape[1]
    nel, nra = hv.shape[3], hv.shape[4]
    rmm = np.zeros((nbeam, npolo, nfreq, nra, nel))
    rmw = np.zeros((npolo, nfreq, nra, nel))
    rmr = np.zeros((npolo, nfreq, nra))
    rmb = np.zeros_like(rmm) if dirty_beam is not None else None

    def beams(v):
        v = v * w[:, np.newaxis, np.newaxis]
        if single_beam:
            return np.sum(v.real, axis=1)[:, np.newaxis]
        return np.fft.irfft(v, nbeam, axis=1) * nbeam

    for fi in range(nfreq):
        rmm[:, :, fi] = beams(np.tensordot(P, hv[:, fi], axes=(1, 0))).transpose(1, 0, 3, 2)
        var = np.tensordot(P2, invert_no_zero(hw[:, fi]), axes=(1, 0))
        rm_var = 0.5 * np.sum((w[:, np.newaxis] ** 2) * var, axis=1)
        rmw[:, fi] = invert_no_zero(rm_var[..., np.newaxis])
        rmr[:, fi] = rm_var**0.5
        if dirty_beam is not None:
            rmb[:, :, fi] = beams(np.tensordot(P, dirty_beam[:, fi], axes=(1, 0))).transpose(1, 0, 3, 2)
    return rmm, rmw, rmr, pol, rmb

"""Projections of m-modes into the SVD and KL bases of the beam-transfer products, on the GPU.

Drop-in for ``draco/analysis/fgfilter.py``: :class:`SVDModeProject` (``:53-146``) and :class:`KLModeProject`
(``:149-239``) with the ``mode`` attribute (``forward`` / ``backward`` / ``filter``, ``:10-50``), ``threshold`` and
``klname``.  Per m the reference calls one driftscan method [3P] that multiplies the m's data by a basis matrix per
frequency (SVD) or per m (KL); here the basis comes over in bulk through the provider protocol
(``core/products.py``: :class:`SVDBasisMixin`, :class:`KLTransform`) and every product of a container runs in one
launch of the batched GEMV kernel (``csrc/gemv.hip``), the basis streamed through HBM in slabs of m.
The weight carried to the output is the reference's crude one: the median of the m's input weights (``:94``) --
``dmm_row_median``.

What the basis matrices contain is driftscan's arithmetic: parity of THAT is unpinned (driftscan is absent).  The
task logic (packing of the modes of all frequencies, ``nmode``, the zero padding, the weight rule, the axis order of
the backward transform) is pinned by the reference's own classes run from source (``tests/golden/fgfilter.npz``).
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..core import containers, io
from ..core.task import ContainerTask
from ..device import Context, ptr
from .transform import _dev_dataset

_SLAB_BYTES = 8 << 30  # basis bytes resident per launch


def _run_gemv(ctx, mats, x_d, y_d, x_off, y_off):
    """``y[y_off[t] : +nrow_t] = mats[t] @ x[x_off[t] : +ncol_t]`` for a list of host matrices, in slabs of
    ``_SLAB_BYTES`` of basis (uploaded pinned, one launch per slab)."""
    t0 = 0
    n = len(mats)
    while t0 < n:
        t1, nbytes = t0, 0
        while t1 < n and (t1 == t0 or nbytes + mats[t1].size * 16 <= _SLAB_BYTES):
            nbytes += mats[t1].size * 16
            t1 += 1
        sizes = np.array([a.size for a in mats[t0:t1]], dtype=np.int64)
        a_off = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        host = torch.empty(int(sizes.sum()), dtype=torch.complex128, pin_memory=True)
        hv = host.numpy()
        for a, o in zip(mats[t0:t1], a_off):
            hv[o : o + a.size] = np.asarray(a, dtype=np.complex128).reshape(-1)
        dev = host.to(ctx.device, non_blocking=True)
        desc = _lib.gemv_desc_array(a_off, x_off[t0:t1], y_off[t0:t1], [a.shape[0] for a in mats[t0:t1]], [a.shape[1] for a in mats[t0:t1]])
        _lib.check(_lib.lib.dmm_gemv_batch(ctx.handle, ptr(dev), _lib.DMM_C128, desc, t1 - t0, ptr(x_d), ptr(y_d)))
        ctx.sync()  # `host` / `dev` are released when the names die
        t0 = t1


def _row_median(ctx, w_d):
    """``np.median`` over everything but the leading axis of a device float64 array -> device ``[n]``."""
    n = w_d.shape[0]
    out = ctx.empty((n,), np.float64)
    flat = w_d.reshape(n, -1).contiguous()
    _lib.check(_lib.lib.dmm_row_median(ctx.handle, ptr(flat), n, flat.shape[1], ptr(out)))
    return out


def _svd_forward(bt, mvis_d):
    """device ``[n_m, 2, nfreq, npairs]`` -> (device ``[n_m, ndofmax]`` packed modes, ``nmode [n_m]``)."""
    ctx = Context.get()
    tel = bt.telescope
    n_m, _, nfreq, npairs = mvis_d.shape
    ntel = 2 * npairs
    x = mvis_d.permute(0, 2, 1, 3).contiguous()  # [m, f, (sign, pair)]: the reference's `tm` (:86), data movement only
    y = ctx.zeros((n_m, bt.ndofmax), np.complex128)
    mats, x_off, y_off = [], [], []
    nmode = np.zeros(n_m, dtype=np.int32)
    for m in range(n_m):
        lens = np.asarray(bt.svd_len(m), dtype=np.int64)
        bounds = np.concatenate([[0], np.cumsum(lens)])
        if bounds[-1] > bt.ndofmax:
            raise ValueError(f"m={m}: {bounds[-1]} SVD modes exceed ndofmax={bt.ndofmax}")
        nmode[m] = bounds[-1]
        for f in range(nfreq):
            if lens[f] == 0:
                continue
            a = np.asarray(bt.beam_ut(m, f))
            if a.shape != (lens[f], ntel):
                raise ValueError(f"beam_ut({m}, {f}) has shape {a.shape}, expected {(int(lens[f]), ntel)}")
            mats.append(a)
            x_off.append((m * nfreq + f) * ntel)
            y_off.append(m * bt.ndofmax + bounds[f])
    _run_gemv(ctx, mats, x, y, np.array(x_off, dtype=np.int64), np.array(y_off, dtype=np.int64))
    return y, nmode


def _svd_backward(bt, svis_d):
    """device ``[n_m, ndofmax]`` packed modes -> device ``[n_m, 2, nfreq, npairs]``."""
    ctx = Context.get()
    tel = bt.telescope
    n_m = svis_d.shape[0]
    nfreq, npairs = tel.nfreq, tel.npairs
    ntel = 2 * npairs
    y = ctx.zeros((n_m, nfreq, ntel), np.complex128)
    mats, x_off, y_off = [], [], []
    for m in range(n_m):
        lens = np.asarray(bt.svd_len(m), dtype=np.int64)
        bounds = np.concatenate([[0], np.cumsum(lens)])
        for f in range(nfreq):
            if lens[f] == 0:
                continue
            a = np.asarray(bt.beam_ut_inv(m, f))
            if a.shape != (ntel, lens[f]):
                raise ValueError(f"beam_ut_inv({m}, {f}) has shape {a.shape}, expected {(ntel, int(lens[f]))}")
            mats.append(a)
            x_off.append(m * svis_d.shape[1] + bounds[f])
            y_off.append((m * nfreq + f) * ntel)
    _run_gemv(ctx, mats, svis_d.contiguous(), y, np.array(x_off, dtype=np.int64), np.array(y_off, dtype=np.int64))
    return y.reshape(n_m, nfreq, 2, npairs).permute(0, 2, 1, 3).contiguous()  # tm.transpose((1, 0, 2)) per m (:135)


def project_one_m(bt, direction, mi, vec):
    """The two reference-visible driftscan calls for ONE m (host in, host out), served by the same kernel."""
    ctx = Context.get()
    tel = bt.telescope
    if direction == "forward":  # vec [nfreq, ntel] -> packed modes of this m
        x = ctx.to_device(np.asarray(vec, dtype=np.complex128).reshape(1, tel.nfreq, 2, tel.npairs).transpose(0, 2, 1, 3), np.complex128)
        y, nmode = _svd_forward(_OneM(bt, mi), x)
        return y[0, : int(nmode[0])].cpu().numpy()
    full = np.zeros((1, bt.ndofmax), dtype=np.complex128)
    n = min(len(vec), bt.ndofmax)
    full[0, :n] = np.asarray(vec)[:n]
    out = _svd_backward(_OneM(bt, mi), ctx.to_device(full, np.complex128))  # [1, 2, nfreq, npairs]
    return out[0].permute(1, 0, 2).contiguous().cpu().numpy()  # [nfreq, 2, npairs] like driftscan returns it


class _OneM:
    """View of a basis provider in which row 0 is m = ``mi``."""

    def __init__(self, bt, mi):
        self._bt, self._mi = bt, mi
        self.telescope, self.ndofmax = bt.telescope, bt.ndofmax

    def svd_len(self, m):
        return self._bt.svd_len(self._mi)

    def beam_ut(self, m, f):
        return self._bt.beam_ut(self._mi, f)

    def beam_ut_inv(self, m, f):
        return self._bt.beam_ut_inv(self._mi, f)


def _kl_apply(kl, direction, ms, nin, vis_d, nmax, threshold):
    """Per m: forward ``evecs[kept] @ vis[m, :nin[m]]``, backward ``inv[:, kept] @ vis[m, :nin[m]]`` -> device
    ``[n_m, nmax]`` + lengths."""
    ctx = Context.get()
    n_m = len(ms)
    y = ctx.zeros((n_m, nmax), np.complex128)
    mats, x_off, y_off = [], [], []
    nout = np.zeros(n_m, dtype=np.int32)
    for i, m in enumerate(ms):
        ev, evecs, inv = kl.modes(int(m))
        keep = kl.kept(int(m), threshold)
        a = evecs[keep] if direction == "forward" else inv[:, keep]
        if a.shape[1] != int(nin[i]):
            raise ValueError(f"m={m}: the KL basis takes {a.shape[1]} modes, the data hold {int(nin[i])}")
        if a.shape[0] > nmax:
            raise ValueError(f"m={m}: {a.shape[0]} output modes exceed the container's {nmax}")
        nout[i] = a.shape[0]
        if a.size:
            mats.append(np.ascontiguousarray(a))
            x_off.append(i * vis_d.shape[1])
            y_off.append(i * nmax)
    _run_gemv(ctx, mats, vis_d.contiguous(), y, np.array(x_off, dtype=np.int64), np.array(y_off, dtype=np.int64))
    return y, nout


def kl_one_m(kl, direction, mi, vec, threshold):
    ctx = Context.get()
    x = ctx.to_device(np.asarray(vec, dtype=np.complex128).reshape(1, -1), np.complex128)
    ev, evecs, inv = kl.modes(mi)
    nmax = max(evecs.shape[0], inv.shape[0], 1)
    y, nout = _kl_apply(kl, direction, [mi], [x.shape[1]], x, nmax, threshold)
    return y[0, : int(nout[0])].cpu().numpy()


class _ProjectFilterBase(ContainerTask):
    """Project data to/from a different basis (``fgfilter.py:10-50``).

    Attributes
    ----------
    mode : {"forward", "backward", "filter"}
        Into the new basis, out of it, or forward then backward (filtering through the basis).
    """

    mode = "forward"
    _config_names = ("mode",)

    def process(self, inp):
        if self.mode == "forward":
            return self._forward(inp)
        if self.mode == "backward":
            return self._backward(inp)
        if self.mode == "filter":
            return self._backward(self._forward(inp))
        return None  # like the reference for any other value (its config type rejects them earlier)

    def _forward(self, inp):
        pass

    def _backward(self, inp):
        pass


def _broadcast_weight(ctx, med, nmode_axis):
    return med[:, None].expand(med.shape[0], nmode_axis).contiguous()


class SVDModeProject(_ProjectFilterBase):
    """SVD projection between the raw m-modes and the reduced degrees of freedom (``fgfilter.py:53-146``).

    Produces the packed SVD modes: per m the modes of each frequency concatenated.
    """

    def setup(self, bt):
        self.beamtransfer = io.get_beamtransfer(bt)

    def _forward(self, mmodes):
        bt = self.beamtransfer
        ctx = Context.get()
        svdmodes = containers.SVDModes(mode=bt.ndofmax, axes_from=mmodes, attrs_from=mmodes, allocate=False)
        mvis = _dev_dataset(mmodes.vis, ctx, np.complex128)
        mw = _dev_dataset(mmodes.weight, ctx, np.float64)
        y, nmode = _svd_forward(bt, mvis)
        svdmodes.attach("vis", y)
        svdmodes.datasets["nmode"] = containers.Dataset(host=nmode)
        svdmodes.attach("vis_weight", _broadcast_weight(ctx, _row_median(ctx, mw), bt.ndofmax))  # :94
        return svdmodes

    def _backward(self, svdmodes):
        bt = self.beamtransfer
        tel = bt.telescope
        ctx = Context.get()
        feed_index = getattr(tel, "input_index", None)
        if feed_index is None:
            feed_index = tel.nfeed  # :104-107
        freqmap = np.zeros(len(tel.frequencies), dtype=[("centre", np.float64), ("width", np.float64)])
        freqmap["centre"][:] = tel.frequencies
        freqmap["width"][:] = np.abs(np.diff(tel.frequencies)[0]) if len(tel.frequencies) > 1 else 0.0  # :113 (IndexError there for one channel)
        prod = np.zeros(len(tel.uniquepairs), dtype=[("input_a", int), ("input_b", int)])
        prod["input_a"], prod["input_b"] = np.asarray(tel.uniquepairs)[:, 0], np.asarray(tel.uniquepairs)[:, 1]
        mmodes = containers.MModes(freq=freqmap, prod=prod, stack=len(prod), input=feed_index, attrs_from=svdmodes, axes_from=svdmodes, allocate=False)
        svis = _dev_dataset(svdmodes.vis, ctx, np.complex128)
        sw = _dev_dataset(svdmodes.weight, ctx, np.float64)
        out = _svd_backward(bt, svis)
        mmodes.attach("vis", out)
        med = _row_median(ctx, sw)
        mmodes.attach("vis_weight", med[:, None, None, None].expand(out.shape).contiguous())  # :141
        svdmodes.nmode[:] = svis.shape[1]  # the reference overwrites its INPUT's nmode with the row length (:133)
        return mmodes


class KLModeProject(_ProjectFilterBase):
    """Project between the SVD and KL basis (``fgfilter.py:149-239``).

    Attributes
    ----------
    threshold : float, optional
        KL mode threshold.
    klname : str
        Name of filter to use.
    """

    threshold = None
    klname = None
    _config_names = ("threshold", "klname")

    def setup(self, manager):
        self.product_manager = manager

    def _kl(self):
        pm = self.product_manager
        if self.klname not in pm.kltransforms:
            # (the reference's forward branch misspells the attribute in this message, fgfilter.py:180, and dies with
            # AttributeError before it can raise; the backward branch raises this RuntimeError, :213-217)
            raise RuntimeError(f"Requested KL basis {self.klname} not available (options are {list(pm.kltransforms.items())!r})")
        return pm.kltransforms[self.klname]

    def _apply(self, direction, inp, out_cls):
        bt = io.get_beamtransfer(self.product_manager.beamtransfer)
        kl = self._kl()
        ctx = Context.get()
        out = out_cls(mode=bt.ndofmax, axes_from=inp, attrs_from=inp, allocate=False)
        vis = _dev_dataset(inp.vis, ctx, np.complex128)
        w = _dev_dataset(inp.weight, ctx, np.float64)
        nin = np.asarray(inp.nmode[:])
        ms = np.asarray(inp.index_map["m"])
        y, nout = _kl_apply(kl, direction, ms, nin, vis, bt.ndofmax, None if self.threshold is None else float(self.threshold))
        out.attach("vis", y)
        out.datasets["nmode"] = containers.Dataset(host=nout)
        out.attach("vis_weight", _broadcast_weight(ctx, _row_median(ctx, w), bt.ndofmax))  # :200, :236
        return out

    def _forward(self, svdmodes):
        return self._apply("forward", svdmodes, containers.KLModes)

    def _backward(self, klmodes):
        return self._apply("backward", klmodes, containers.SVDModes)

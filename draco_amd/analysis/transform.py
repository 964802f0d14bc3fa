"""m-mode transform tasks on the GPU.

Drop-in for ``draco/analysis/transform.py``:

* :class:`MModeTransform`         ``transform.py:535-641``
* :class:`MModeInverseTransform`  ``transform.py:708-792``
* :func:`_make_marray`            ``transform.py:644-705``
* :func:`_make_ssarray`           ``transform.py:814-817`` (+ ``_unpack_marray`` :820-851)

Same class names, config attributes (``remove_integration_window``, ``use_fftw``,
``nra``, ``apply_integration_window``), ``setup``/``process`` signatures and exceptions.
The arithmetic runs in ``libdraco_amd.so`` (``csrc/mfft.hip``): a batched in-LDS FFT fused
with the +/-m pack and the transposed store.  ``use_fftw`` is accepted and ignored
(there is one FFT, the HIP one).
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..core import containers, io
from ..core.task import ContainerTask
from ..device import Context, ptr
from ..util import tools


def _dev_dataset(ds, ctx, dtype):
    """Device tensor for a dataset-like (our Dataset, ndarray, or foreign ``ds[:]``)."""
    if isinstance(ds, containers.Dataset):
        t = ds.device(ctx)
        want = {np.complex64: torch.complex64, np.complex128: torch.complex128, np.float32: torch.float32, np.float64: torch.float64}[dtype]
        return t if t.dtype == want else t.to(want)
    if isinstance(ds, torch.Tensor):
        return ctx.to_device(ds, dtype)
    return ctx.to_device(np.asarray(ds[:]), dtype)


def mmode_forward(ctx, vis_d, weight_d, mmax, remove_integration_window=False, vis_dtype=np.complex128):
    """Device-level transform: ``vis [..., nra]`` c64 (+ weight f32) -> ``(mvis, mweight)``.

    ``mvis [mmax+1, 2, ...]`` complex128 (complex64 on request) and ``mweight`` float64, as
    ``transform.py:594-639``.  ``weight`` may have fewer leading axes than ``vis`` (hybrid streams).
    """
    lead = tuple(vis_d.shape[:-1])
    nra = int(vis_d.shape[-1])
    nrow = int(np.prod(lead)) if lead else 1
    mscale = wscale = None
    if remove_integration_window:
        m = np.arange(mmax + 1)
        w = np.sinc(m / nra)  # transform.py:631-633
        mscale = ctx.to_device(tools.invert_no_zero(w), np.float64)
        wscale = ctx.to_device(w**2, np.float64)
    mvis = ctx.empty((mmax + 1, 2, *lead), vis_dtype)
    out_dt = _lib.DMM_C128 if np.dtype(vis_dtype) == np.complex128 else _lib.DMM_C64
    _lib.check(_lib.lib.dmm_mfft_pack(ctx.handle, ptr(vis_d), nrow, nra, ptr(mvis), mmax, out_dt, ptr(mscale)))
    mweight = None
    if weight_d is not None:
        wlead = tuple(weight_d.shape[:-1])
        wrow = int(np.prod(wlead)) if wlead else 1
        mweight = ctx.empty((mmax + 1, 2, *wlead), np.float64)
        _lib.check(_lib.lib.dmm_mmode_weight(ctx.handle, ptr(weight_d), wrow, nra, ptr(mweight), mmax, ptr(wscale)))
    return mvis, mweight


class MModeTransform(ContainerTask):
    """Transform a sidereal stream to m-modes (``transform.py:535-641``).

    The maximum m is ``telescope.mmax`` if a manager was given to :meth:`setup`, else
    ``nra // 2``.

    Attributes
    ----------
    remove_integration_window : bool
        Deconvolve the rectangular RA integration window (vis and weights).
    use_fftw : bool
        Accepted for compatibility; the transform always runs the HIP FFT.
    """

    remove_integration_window = False
    use_fftw = True
    _config_names = ("remove_integration_window", "use_fftw")

    telescope = None

    def setup(self, manager=None):
        """Set the telescope instance if a manager object is given (``transform.py:557-571``)."""
        if manager is not None:
            self.telescope = io.get_telescope(manager)
        else:
            self.telescope = None

    def process(self, sstream):
        """Perform the m-mode transform: ``SiderealStream -> MModes`` (``transform.py:573-641``)."""
        contmap = {
            containers.SiderealStream: containers.MModes,
            containers.HybridVisStream: containers.HybridVisMModes,
        }
        out_cont = contmap[sstream.__class__]  # KeyError for unsupported containers, like :590

        sstream.redistribute("freq")
        ctx = Context.get()
        svis = _dev_dataset(sstream.vis, ctx, np.complex64)
        sweight = _dev_dataset(sstream.weight, ctx, np.float32)
        nra = int(svis.shape[-1])

        if self.telescope is not None:
            mmax = int(self.telescope.mmax)
        else:
            mmax = nra // 2

        ma = out_cont(mmax=mmax, oddra=bool(nra % 2), axes_from=sstream, attrs_from=sstream, comm=sstream.comm, allocate=False)
        ma.redistribute("freq")
        hybrid = out_cont is containers.HybridVisMModes  # keeps complex64 / float32 (containers.py:1559-1574)
        mvis, mweight = mmode_forward(ctx, svis, sweight, mmax, self.remove_integration_window, np.complex64 if hybrid else np.complex128)
        ma.attach("vis", mvis)
        ma.attach("vis_weight", mweight.to(torch.float32) if hybrid else mweight)
        return ma


def _make_marray(ts, mmodes=None, mmax=None, dtype=None, use_fftw=True):
    """GPU version of the reference helper (``transform.py:644-705``), same contract.

    ``ts [..., N]`` complex -> ``mmodes [mmax+1, 2, ...]``; writes into ``mmodes`` if given
    (every slot of it is defined: filled modes, zeros elsewhere).
    """
    ts = np.asarray(ts)
    if dtype is None:
        dtype = np.complex64
    if mmodes is None and mmax is None:
        raise ValueError("One of `mmodes` or `mmax` must be set.")
    if mmodes is not None and mmax is not None:
        raise ValueError("If mmodes is set, mmax must be None.")
    if mmodes is not None and mmodes.shape[2:] != ts.shape[:-1]:
        raise ValueError(f"ts and mmodes have incompatible shapes: {mmodes.shape[2:]} != {ts.shape[:-1]}")
    if mmax is None:
        mmax = mmodes.shape[0] - 1
    ctx = Context.get()
    ts_d = ctx.to_device(ts, np.complex64)
    mvis, _ = mmode_forward(ctx, ts_d, None, int(mmax))
    out = mvis.cpu().numpy()
    if mmodes is None:
        return out.astype(dtype, copy=False)
    mmodes[...] = out
    return mmodes


def _unpack_limits(ctx, mvis_d, n=None):
    """``mmax_plus`` / ``mmax_minus`` / ``ntimes`` of ``_unpack_marray`` (``transform.py:824-836``)."""
    import ctypes as C

    n_m = int(mvis_d.shape[0])
    nrow = int(np.prod(mvis_d.shape[2:])) if mvis_d.dim() > 2 else 1
    mmax_plus = n_m - 1
    z = C.c_int(0)
    _lib.check(_lib.lib.dmm_mrow_is_zero(ctx.handle, ptr(mvis_d), n_m, nrow, mmax_plus, 1, C.byref(z)))
    mmax_minus = mmax_plus - 1 if z.value else mmax_plus
    if n is None:
        ntimes = mmax_plus + mmax_minus + 1
    else:
        ntimes = int(n)
        mmax_plus = min(ntimes // 2, mmax_plus)
        mmax_minus = min((ntimes - 1) // 2, mmax_minus)
    return mmax_plus, max(mmax_minus, 0), ntimes


def mmode_inverse(ctx, mvis_d, n=None, mscale=None, limits=None):
    """Device-level inverse: ``mvis [n_m, 2, ...]`` c128 -> ``vis [..., ntimes]`` complex64."""
    lead = tuple(mvis_d.shape[2:])
    nrow = int(np.prod(lead)) if lead else 1
    mp, mm, ntimes = limits if limits is not None else _unpack_limits(ctx, mvis_d, n)
    out = ctx.empty((*lead, ntimes), np.complex64)
    _lib.check(_lib.lib.dmm_mifft_unpack(ctx.handle, ptr(mvis_d), int(mvis_d.shape[0]), nrow, ntimes, mp, mm, ptr(mscale), ptr(out)))
    return out


def _make_ssarray(mmodes, n=None):
    """GPU version of ``transform.py:814-817``; returns complex64 (the SiderealStream dtype).

    The reference returns complex128 and casts on assignment into ``SiderealStream.vis``
    (``transform.py:787``); the kernel computes in float64 and rounds once at the store.
    """
    ctx = Context.get()
    mv = ctx.to_device(np.asarray(mmodes), np.complex128)
    return mmode_inverse(ctx, mv, n).cpu().numpy()


class MModeInverseTransform(ContainerTask):
    """Transform m-modes back to a sidereal stream (``transform.py:708-792``).

    Attributes
    ----------
    nra : int
        Number of RA bins in the output (default: natural ``2*mmax + oddra``).
    apply_integration_window : bool
        Apply the rectangular integration window to visibilities and weights.  Unlike the
        reference (warning at ``transform.py:713-714``) the input container is NOT modified.
    """

    nra = None
    apply_integration_window = False
    _config_names = ("nra", "apply_integration_window")

    def process(self, mmodes):
        """``MModes -> SiderealStream`` (``transform.py:733-792``)."""
        mmodes.redistribute("freq")
        ctx = Context.get()
        nra_cont = 2 * mmodes.mmax + (1 if mmodes.oddra else 0)
        nra = self.nra if self.nra is not None else nra_cont

        mvis = _dev_dataset(mmodes.vis, ctx, np.complex128)
        mweight = _dev_dataset(mmodes.weight, ctx, np.float64)
        mscale = None
        wfac = 1.0
        if self.apply_integration_window:
            m = np.arange(mmodes.mmax + 1)
            w = np.sinc(m / nra)
            mscale = ctx.to_device(w, np.float64)
            wfac = float(tools.invert_no_zero(w)[0] ** 2)  # weight[0, 0] *= inv_w[0]**2 (:769)

        vis = mmode_inverse(ctx, mvis, int(nra), mscale)
        nra = int(vis.shape[-1])
        sstream = containers.SiderealStream(ra=nra, axes_from=mmodes, attrs_from=mmodes, comm=mmodes.comm, allocate=False)
        sstream.redistribute("freq")
        sstream.attach("vis", vis)
        # no time information survives for the weights: the time average per (freq, baseline), :790
        w0 = (mweight[0, 0] * (wfac / nra)).to(torch.float32)
        sstream.attach("vis_weight", w0.unsqueeze(-1).expand(*w0.shape, nra).contiguous())
        return sstream

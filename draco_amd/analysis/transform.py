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


class SiderealMModeResample(ContainerTask):
    """Resample a sidereal stream by FFT: a forward then an inverse m-mode transform
    (``transform.py:795-811``, the reference's ``group_tasks(MModeTransform, MModeInverseTransform)``).

    Attributes
    ----------
    nra : int
        The number of RA bins for the output stream.
    remove_integration_window, apply_integration_window : bool
        Remove the integration window from the incoming data, and/or apply it to the output stream.
    use_fftw : bool
        Accepted and ignored, as in :class:`MModeTransform`.
    """

    nra = None
    remove_integration_window = False
    apply_integration_window = False
    use_fftw = True
    _config_names = ("nra", "remove_integration_window", "apply_integration_window", "use_fftw")

    _manager = None

    def setup(self, manager=None):
        self._manager = manager

    def process(self, sstream):
        fwd = MModeTransform(remove_integration_window=self.remove_integration_window, use_fftw=self.use_fftw)
        fwd.setup(self._manager)
        inv = MModeInverseTransform(nra=self.nra, apply_integration_window=self.apply_integration_window)
        return inv.process(fwd.process(sstream))  # the m-modes never leave the device


def _cmap(i, j, n):
    if i > j:
        i, j = j, i
    return (n * (n + 1) // 2) - ((n - i) * (n - i + 1) // 2) + (j - i)


def _find_inputs(input_index, inputs, require_match=False):
    """``tools.find_inputs`` (``util/tools.py:130-169``): match on ``correlator_input`` or ``chan_id``."""
    names = input_index.dtype.names or ()
    field = "correlator_input" if "correlator_input" in names else ("chan_id" if "chan_id" in names else None)
    if field is None:
        raise ValueError("`input_index` must have either a `chan_id` or `correlator_input` field.")
    if field not in (inputs.dtype.names or ()):
        raise ValueError(f"`inputs` array does not have a `{field!s}` field.")
    return tools.find_keys(input_index[field], inputs[field], require_match=require_match)


class TelescopeStreamMixIn:
    """Pre-computes the telescope's prod / stack / reverse-stack index maps (``transform.py:91-139``)."""

    def setup(self, tel):
        self.telescope = tel = io.get_telescope(tel)
        n = tel.nfeed
        self.bt_stack = np.array(
            [(_cmap(a, b, n), 0) if a <= b else (_cmap(b, a, n), 1) for a, b in tel.uniquepairs],
            dtype=[("prod", "<u4"), ("conjugate", "u1")],
        )
        triu = np.triu_indices(n)
        self.bt_prod = np.zeros(len(triu[0]), dtype=[("input_a", "<u2"), ("input_b", "<u2")])
        self.bt_prod["input_a"], self.bt_prod["input_b"] = triu
        fm = np.asarray(tel.feedmask)[triu]
        self.bt_rev = np.empty(fm.size, dtype=[("stack", "<u4"), ("conjugate", "u1")])
        self.bt_rev["stack"] = np.where(fm, np.asarray(tel.feedmap)[triu], tel.npairs)
        self.bt_rev["conjugate"] = np.where(fm, np.asarray(tel.feedconj)[triu], 0)


def _redefine_stack_index_map(tel, tel_index, prod, stack, reverse_stack):
    """Give every stack entry a representative product the telescope can use (``util/tools.py:359-414``).

    A product is *usable* when both of its inputs exist in the telescope (``tel_index[i]`` is not None) and the pair
    is not masked there.  Entries whose current representative is usable keep it; the others take the first usable
    member of their stack (lowest product index) together with that member's conjugation flag.  Returns the new map
    and, per entry, whether a usable representative exists at all.
    """
    present = np.array([t is not None for t in tel_index], dtype=bool)
    slot = np.array([t if t is not None else 0 for t in tel_index], dtype=np.int64)
    pa, pb = np.asarray(prod["input_a"], dtype=np.int64), np.asarray(prod["input_b"], dtype=np.int64)
    usable = present[pa] & present[pb] & np.asarray(tel.feedmask)[slot[pa], slot[pb]]  # per product of the file

    nstack = stack.size
    member_of = np.asarray(reverse_stack["stack"], dtype=np.int64)
    cand = np.flatnonzero(usable & (member_of >= 0) & (member_of < nstack))
    first = np.full(nstack, len(pa), dtype=np.int64)
    np.minimum.at(first, member_of[cand], cand)  # lowest usable member of every stack

    current = np.asarray(stack["prod"], dtype=np.int64)
    keep = usable[current]
    found = keep | (first < len(pa))
    swap = ~keep & found
    out = stack.copy()
    out["prod"][swap] = first[swap]
    out["conjugate"][swap] = reverse_stack["conjugate"][first[swap]]
    return out, found


class CollateProducts(TelescopeStreamMixIn, ContainerTask):
    """Extract, order and stack the correlation products for map-making (``transform.py:142-330``).

    The input may hold more inputs and frequencies than the telescope; the converse raises
    ``ValueError`` like the reference.  For inputs that are ALREADY redundancy-stacked the
    representative product of every stack entry is re-derived so that it only involves inputs the
    telescope has and does not mask (``transform.py:206-221``, ``util/tools.py:359-414``).

    Attributes
    ----------
    weight : {"natural", "uniform", "inverse_variance"}
        How to weight the redundant baselines when stacking.
    """

    weight = "natural"
    _config_names = ("weight",)

    def process(self, ss):
        tel = self.telescope
        ss_input = np.asarray(ss.index_map["input"])
        input_ind = _find_inputs(tel.input_index, ss_input, require_match=False)
        rev_input_ind = _find_inputs(ss_input, tel.input_index, require_match=True)
        freq_ind = tools.find_keys(list(ss.index_map["freq"]["centre"]), list(tel.frequencies), require_match=True)
        bt_freq = ss.index_map["freq"][freq_ind]
        file_prod = np.asarray(ss.index_map["prod"])
        stacked = bool(getattr(ss, "is_stacked", False))
        if stacked:
            stack_new, stack_flag = _redefine_stack_index_map(tel, input_ind, file_prod, np.asarray(ss.index_map["stack"]), np.asarray(ss.reverse_map["stack"]))
            if not np.all(stack_flag):
                self.log.warning(f"There are {np.sum(~stack_flag):0.0f} stacked baselines that are masked in the telescope instance.")
            ss_prod = file_prod[stack_new["prod"]]
            ss_conj = stack_new["conjugate"].astype(bool)
        else:
            ss_prod = file_prod
            ss_conj = np.zeros(len(ss_prod), dtype=bool)
        if self.weight not in ("natural", "uniform", "inverse_variance"):
            raise ValueError(f"unknown weight {self.weight!r}")

        sp = type(ss)(freq=bt_freq, ra=np.asarray(ss.index_map["ra"]), input=tel.input_index, prod=self.bt_prod, stack=self.bt_stack,
                      reverse_map_stack=self.bt_rev, attrs_from=ss, comm=ss.comm, allocate=False)
        ctx = Context.get()
        ssv = _dev_dataset(ss.vis, ctx, np.complex64)
        ssw = _dev_dataset(ss.weight, ctx, np.float32)
        nf_in, nprod_in, nt = ssv.shape

        # invert the reference's product loop (transform.py:277-320) into output-major lists
        lists = [[] for _ in range(tel.npairs)]
        for ss_pi, (ii, ij) in enumerate(zip(ss_prod["input_a"], ss_prod["input_b"])):
            bi, bj = input_ind[ii], input_ind[ij]
            if bi is None or bj is None:
                continue
            sp_pi = int(tel.feedmap[bi, bj])
            if sp_pi < 0:
                continue
            lists[sp_pi].append((ss_pi, bool(tel.feedconj[bi, bj]) != bool(ss_conj[ss_pi])))  # conjugate unless feedconj == conj (:303)
        ptr_ = np.zeros(tel.npairs + 1, dtype=np.int32)
        ptr_[1:] = np.cumsum([len(x) for x in lists])
        src = np.array([p for x in lists for p, _ in x], dtype=np.int32)
        cj = np.array([c for x in lists for _, c in x], dtype=np.uint8)

        flags = None
        if "input_flags" in ss.datasets:
            flags = np.asarray(ss.input_flags[:], dtype=np.float32)
        red_d = None
        if self.weight != "inverse_variance":
            fl = flags if flags is not None and np.any(flags) else np.ones((len(ss_input), nt), np.float32)
            if stacked:  # products that went into each stack entry with both inputs good (util/tools.py:313-356)
                red = np.zeros((nprod_in, nt), np.float32)
                np.add.at(red, np.asarray(ss.reverse_map["stack"]["stack"]).astype(np.int64),
                          fl[file_prod["input_a"].astype(int)] * fl[file_prod["input_b"].astype(int)])
            else:
                red = fl[ss_prod["input_a"].astype(int)] * fl[ss_prod["input_b"].astype(int)]  # one product per "stack" entry
            if self.weight == "uniform":
                red = (red > 0).astype(np.float32)
            red_d = ctx.to_device(np.ascontiguousarray(red, dtype=np.float32))
        out_v = ctx.empty((len(freq_ind), tel.npairs, nt), np.complex64)
        out_w = ctx.empty((len(freq_ind), tel.npairs, nt), np.float32)
        d = lambda a: ctx.to_device(a) if a.size else ctx.to_device(np.zeros(1, a.dtype))  # noqa: E731
        # keep the index tensors referenced until the launch is enqueued (a temporary would be recycled at once)
        find_d, ptr_d, src_d, cj_d = d(np.asarray(freq_ind, dtype=np.int32)), d(ptr_), d(src), d(cj)
        _lib.check(
            _lib.lib.dmm_collate_products(
                ctx.handle, ptr(ssv), ptr(ssw), int(nf_in), int(nprod_in), int(nt), len(freq_ind), ptr(find_d),
                int(tel.npairs), ptr(ptr_d), ptr(src_d), ptr(cj_d), ptr(red_d), ptr(out_v), ptr(out_w),
            )
        )
        ctx.sync()  # the small index tensors go out of scope below
        sp.attach("vis", out_v)
        sp.attach("vis_weight", out_w)
        if flags is not None:
            sp.add_dataset("input_flags", allocate=True)
            sp.input_flags[:] = flags[rev_input_ind, :]
        return sp

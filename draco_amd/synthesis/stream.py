"""Simulating sidereal stream data from a sky map, on the GPU.

Drop-in for ``SimulateSidereal`` (``draco/synthesis/stream.py:22-178``): same config
attribute (``stacked``), ``setup(bt)`` / ``process(map_) -> SiderealStream`` signatures,
same ``ValueError`` on a frequency mismatch (``stream.py:81-82``).

Pipeline (all device resident): forward SHT ``map -> a_lm`` (``csrc/sht.hip`` replaces
``hputil.sphtrans_sky``, ``stream.py:85``), batched ``v_m[f] = B_m[f] a_m[f]``
(``csrc/solve_dirty.hip::k_project`` replaces the ``project_vector_sky_to_telescope``
loop, ``stream.py:109-112``), +/-m unpack fused with the inverse FFT of length
``2*mmax+1`` (``csrc/mfft.hip::k_mifft_unpack`` replaces ``stream.py:124-140``).  The two
frequency<->m transposes of the reference (``stream.py:96,119``) do not exist here.
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..analysis import _solve
from ..analysis.transform import _dev_dataset, mmode_inverse
from ..core import containers, io
from ..core.task import ContainerTask
from ..device import Context, ptr


class SimulateSidereal(ContainerTask):
    """Create a simulated sidereal dataset from an input map (``stream.py:22-178``).

    Attributes
    ----------
    stacked : bool
        Treat non-full-triangle beam transfers as collated products and set the
        ``index_map/stack`` and ``reverse_map/stack`` entries (default True).
    map2alm_iter : int
        Jacobi refinement iterations of the forward SHT (healpy's ``iter``, default 3).
        Not a reference attribute (cora's wrapper does not expose it).
    """

    stacked = True
    map2alm_iter = 3
    b_dtype = "complex128"
    pool_bytes = None
    _config_names = ("stacked", "map2alm_iter", "b_dtype", "pool_bytes")

    _engine = None

    def setup(self, bt):
        """Setup the simulation (``stream.py:37-46``)."""
        self.beamtransfer = io.get_beamtransfer(bt)
        self.telescope = io.get_telescope(bt)
        self._engine = None

    def _get_engine(self):
        if self._engine is None:
            dt = {"complex128": _lib.DMM_C128, "complex64": _lib.DMM_C64}[str(self.b_dtype)]
            self._engine = _solve.SolveEngine(self.beamtransfer, Context.get(), dt, _lib.DMM_B_PACKED, self.pool_bytes)
        return self._engine

    def _sky_alm(self, ctx, row_map, nside):
        """sky -> a_lm trimmed to m <= mmax, device ``[nfreq, npol, mmax+1, lmax+1]`` (``stream.py:85-90``).

        The one stage that stands in for third-party code (``hputil.sphtrans_sky`` -> healpy); kept apart so that
        the rest of ``process`` can be checked against the reference run with a recorded a_lm
        (``tests/golden/stream_simulate.npz``).
        """
        tel = self.telescope
        nfreq, npol = row_map.shape[:2]
        alm = ctx.empty((nfreq, npol, tel.mmax + 1, tel.lmax + 1), np.complex128)
        _lib.check(_lib.lib.dmm_map2alm(ctx.handle, ptr(row_map), nfreq, npol, tel.lmax, tel.mmax, nside, int(self.map2alm_iter), ptr(alm)))
        return alm

    def process(self, map_):
        """Simulate a SiderealStream from ``map_`` (``stream.py:48-178``)."""
        tel = self.telescope
        lmax, mmax, nfreq, npol = tel.lmax, tel.mmax, tel.nfreq, tel.num_pol_sky
        ntime = 2 * mmax + 1  # stream.py:76

        freqmap = map_.index_map["freq"][:]
        if (np.asarray(tel.frequencies) != freqmap["centre"]).any():
            raise ValueError("Frequencies in map do not match those in Beam Transfers.")

        eng = self._get_engine()
        ctx = eng.ctx
        row_map = _dev_dataset(map_.map, ctx, np.float64)  # [nfreq, npol_map, npix]
        if row_map.shape[1] < npol:
            raise ValueError(f"map has {row_map.shape[1]} polarisations, the telescope needs {npol}")
        row_map = row_map[:, :npol].contiguous()
        nside = int(round((row_map.shape[-1] // 12) ** 0.5))

        alm = self._sky_alm(ctx, row_map, nside)

        # a_lm -> m-mode visibilities for every (m, f) (stream.py:102-112)
        vis_m = eng.project(alm, list(range(nfreq)), mmax)  # [mmax+1, 2, nfreq, npairs]

        # unwrap +/-m (conjugate only, stream.py:128-133) and transform to time (stream.py:138)
        vis_stream = mmode_inverse(ctx, vis_m, limits=(mmax, mmax, ntime))  # [nfreq, npairs, ntime] c64

        feed_index = getattr(tel, "input_index", None)
        if feed_index is None:
            feed_index = tel.nfeed  # stream.py:143-146
        kwargs = {}
        if tel.npairs != (tel.nfeed + 1) * tel.nfeed // 2 and self.stacked and hasattr(tel, "index_map_prod") and not getattr(tel, "_free", False):
            kwargs["prod"] = tel.index_map_prod
            kwargs["stack"] = tel.index_map_stack
            kwargs["reverse_map_stack"] = tel.reverse_map_stack
        else:
            prod_map = np.zeros(tel.uniquepairs.shape[0], dtype=[("input_a", int), ("input_b", int)])
            prod_map["input_a"] = tel.uniquepairs[:, 0]
            prod_map["input_b"] = tel.uniquepairs[:, 1]
            kwargs["prod"] = prod_map

        sstream = containers.SiderealStream(freq=freqmap, ra=ntime, input=feed_index, comm=getattr(map_, "comm", None), allocate=False, **kwargs)
        sstream.attach("vis", vis_stream)
        sstream.attach("vis_weight", torch.ones(vis_stream.shape, dtype=torch.float32, device=ctx.device))  # stream.py:176
        return sstream


class ExpandProducts(ContainerTask):
    """Un-wrap collated products to the full triangle (``synthesis/stream.py:181-246``).

    The inverse data-format step of :class:`~draco_amd.analysis.transform.CollateProducts`: every
    product ``(i, j >= i)`` of the telescope's inputs gets the (conjugated, if ``feedconj``)
    visibility of its unique baseline and unit weight; products of masked pairs
    (``feedmap < 0``) stay zero with zero weight.
    """

    def setup(self, telescope):
        self.telescope = io.get_telescope(telescope)

    def process(self, sstream):
        sstream.redistribute("freq")
        tel = self.telescope
        inputs = sstream.index_map.get("input")
        ninput = len(inputs) if inputs is not None else tel.nfeed
        ii, jj = np.triu_indices(ninput)
        prod = np.zeros(len(ii), dtype=[("input_a", int), ("input_b", int)])
        prod["input_a"], prod["input_b"] = ii, jj
        nprod = len(prod)
        fwd_stack = np.zeros(nprod, dtype=[("prod", "<u4"), ("conjugate", "u1")])
        fwd_stack["prod"] = np.arange(nprod)
        rev_stack = np.zeros(nprod, dtype=[("stack", "<u4"), ("conjugate", "u1")])
        rev_stack["stack"] = np.arange(nprod)
        out = containers.SiderealStream(
            freq=sstream.index_map["freq"], ra=np.asarray(sstream.index_map["ra"]), input=inputs if inputs is not None else ninput,
            prod=prod, stack=fwd_stack, reverse_map_stack=rev_stack, attrs_from=sstream, comm=sstream.comm, allocate=False,
        )
        ctx = Context.get()
        vis = _dev_dataset(sstream.vis, ctx, np.complex64)
        nfreq, nstack, nra = vis.shape
        src = ctx.to_device(np.asarray(tel.feedmap)[ii, jj].astype(np.int32))
        cj = ctx.to_device(np.asarray(tel.feedconj)[ii, jj].astype(np.uint8))
        out_v = ctx.empty((nfreq, nprod, nra), np.complex64)
        out_w = ctx.empty((nfreq, nprod, nra), np.float32)
        _lib.check(_lib.lib.dmm_expand_products(ctx.handle, ptr(vis), int(nfreq), int(nstack), int(nra), int(nprod), ptr(src), ptr(cj), ptr(out_v), ptr(out_w)))
        ctx.sync()  # src / cj go out of scope below
        out.attach("vis", out_v)
        out.attach("vis_weight", out_w)
        return out

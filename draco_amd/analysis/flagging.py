"""Masking of m-mode data ahead of map-making, on the GPU.

Drop-in for ``MaskMModeData`` (``draco/analysis/flagging.py:113-173``): same config attributes
(``auto_correlations``, ``m_zero``, ``positive_m``, ``negative_m``, ``mask_low_m``) and the same
in-place semantics (the input container is returned with its weights zeroed).  It sits between
``MModeTransform`` and the map-makers in the reference's real-data pipeline
(``test/pipe_config.yaml:100-131``); the map-makers treat zero weights as "row absent".
"""

from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..core.task import ContainerTask
from ..device import Context, ptr
from .transform import _dev_dataset


def _prodstack(cont):
    """Representative (input_a, input_b) of every stack entry (``containers.py:211-229``)."""
    prod = cont.index_map.get("prod")
    stack = cont.index_map.get("stack")
    if prod is None:
        return None
    if stack is not None and stack.dtype.names is not None and "prod" in stack.dtype.names:
        return prod[stack["prod"]]
    return prod


class MaskMModeData(ContainerTask):
    """Mask out m-mode data ahead of map making (``flagging.py:113-173``).

    Attributes
    ----------
    auto_correlations : bool
        Exclude auto correlations if set (default False: autos ARE masked, as in the reference).
    m_zero : bool
        Ignore the m=0 mode (default False: m=0 is masked).
    positive_m, negative_m : bool
        Include positive / negative m-modes (default True).
    mask_low_m : int, optional
        If set, mask out m's lower than this threshold.
    """

    auto_correlations = False
    m_zero = False
    positive_m = True
    negative_m = True
    mask_low_m = None
    _config_names = ("auto_correlations", "m_zero", "positive_m", "negative_m", "mask_low_m")

    def process(self, mmodes):
        mmodes.redistribute("freq")
        ctx = Context.get()
        mw = _dev_dataset(mmodes.weight, ctx, np.float64)
        n_m, _, nfreq, nstack = mw.shape
        is_auto = None
        if not self.auto_correlations:
            ps = _prodstack(mmodes)
            if ps is None or len(ps) != nstack:
                raise ValueError("MaskMModeData needs the prod/stack index maps to find the auto-correlations")
            is_auto = ctx.to_device((ps["input_a"] == ps["input_b"]).astype(np.uint8))
        _lib.check(
            _lib.lib.dmm_mask_mmode_weight(
                ctx.handle, ptr(mw), int(n_m), int(nfreq), int(nstack), ptr(is_auto), int(bool(self.m_zero)),
                int(bool(self.positive_m)), int(bool(self.negative_m)), int(self.mask_low_m or 0),
            )
        )
        mmodes.weight.set_device(mw)
        return mmodes

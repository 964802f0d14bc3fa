"""Oracle: m-mode masking.  TEST INFRASTRUCTURE ONLY.

Restates ``MaskMModeData.process`` (reference ``draco/analysis/flagging.py:153-173``) on a plain
weight array ``[n_m, 2, nfreq, nstack]``.  Pinned by ``tests/golden/flagging_mask_mmode.npz``.
"""

from __future__ import annotations

import numpy as np


def mask_mmode_weight(mw, prodstack, auto_correlations=False, m_zero=False, positive_m=True, negative_m=True, mask_low_m=None):
    mw = np.array(mw, copy=True)
    if not auto_correlations:
        for pi, (fi, fj) in enumerate(prodstack):
            if fi == fj:
                mw[..., pi] = 0.0
    if not m_zero:
        mw[0] = 0.0
    if not positive_m:
        mw[1:, 0] = 0.0
    if not negative_m:
        mw[1:, 1] = 0.0
    if mask_low_m:
        mw[:mask_low_m] = 0.0
    return mw

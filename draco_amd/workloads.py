"""The benchmark workloads of BASELINE.json / SURVEY.md 8(d): telescope shape, band, RA samples, band limit, map
resolution.  Product-side copy (``bench.py``, ``tools/``); the CPU checker keeps its own in ``oracle/synth.py`` and
a test asserts that the two agree.

``ncyl`` cylinders x ``nfeed_cyl`` feed positions x 2 polarisations on a regular grid; ``lmax = mmax``.
"""

from __future__ import annotations

import numpy as np

CONFIGS = {
    1: dict(ncyl=1, nfeed_cyl=8, nfreq=4, nra=127, lmax=63, nside=32),
    2: dict(ncyl=2, nfeed_cyl=16, nfreq=64, nra=512, lmax=256, nside=128),
    3: dict(ncyl=2, nfeed_cyl=32, nfreq=256, nra=1024, lmax=512, nside=256),
    4: dict(ncyl=2, nfeed_cyl=64, nfreq=512, nra=2048, lmax=1024, nside=512),
    5: dict(ncyl=2, nfeed_cyl=64, nfreq=1024, nra=2047, lmax=1023, nside=512),
}


def frequencies(nfreq):
    """Band of the synthetic telescope: ``nfreq`` channels over 400-800 MHz (MHz, lower edges spaced evenly)."""
    return np.linspace(400.0, 800.0, nfreq, endpoint=False)

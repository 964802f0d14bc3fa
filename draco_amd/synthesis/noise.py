"""Instrumental-noise tasks for simulated streams (host side).

Drop-in for ``GaussianNoise`` (``draco/synthesis/noise.py:178-284``) and ``SampleNoise``
(``:287-374``): same config attributes and ``setup``/``process`` signatures, in-place
modification of the input container like the reference.  These are input generation for
BASELINE config 5: random numbers come from a host ``np.random.Generator``
(the reference's ``RandomTask.rng`` [3P]); nothing here is on the timed hot path.
"""

from __future__ import annotations

import numpy as np

from ..core import containers, io
from ..core.task import ContainerTask
from ..util import random

STELLAR_S = 1.0 / 1.002737909350795  # length of a sidereal second in SI seconds (caput.astro.constants [3P])


def _cmap(i, j, n):
    """Upper-triangle product index of the feed pair (i <= j) (``tools.py:21-39``)."""
    if i > j:
        i, j = j, i
    return (n * (n + 1) // 2) - ((n - i) * (n - i + 1) // 2) + (j - i)


def _prodstack(data):
    """Representative input pair per stacked product, conjugation applied (``containers.py:211-229``)."""
    prod = data.index_map["prod"]
    stack = data.index_map.get("stack")
    if stack is None or stack.dtype.names is None or "prod" not in stack.dtype.names:
        return prod
    t = prod[stack["prod"]].copy()
    conj = stack["conjugate"].astype(bool)
    t["input_a"] = np.where(conj, prod[stack["prod"]]["input_b"], prod[stack["prod"]]["input_a"])
    t["input_b"] = np.where(conj, prod[stack["prod"]]["input_a"], prod[stack["prod"]]["input_b"])
    return t


class _RandomTask(ContainerTask):
    seed = None
    _config_names = ("seed",)
    _rng = None

    @property
    def rng(self):
        if self._rng is None:
            self._rng = np.random.default_rng(self.seed)
        return self._rng

    @rng.setter
    def rng(self, value):
        self._rng = value


class GaussianNoise(_RandomTask):
    """Add Gaussian distributed noise to a visibility dataset (``noise.py:178-284``).

    Attributes
    ----------
    recv_temp : float
        The temperature of the noise to add.
    ndays : float
        Multiplies the number of samples in each measurement.
    set_weights : bool
        Set the weights to ``1/sigma**2``.
    add_noise : bool
        Add the noise (False: only set the weights).
    """

    recv_temp = 50.0
    ndays = 733.0
    set_weights = True
    add_noise = True
    _config_names = ("recv_temp", "ndays", "set_weights", "add_noise")
    telescope = None

    def setup(self, manager=None):
        self.telescope = io.get_telescope(manager) if manager is not None else None

    def process(self, data):
        data.redistribute("freq")
        visdata = data.vis[:]
        if isinstance(data, containers.SiderealStream):
            dt = 240 * (data.ra[1] - data.ra[0]) * STELLAR_S
            ntime = len(data.ra)
        else:
            dt = data.time[1] - data.time[0]
            ntime = len(data.time)
        df = data.index_map["freq"]["width"][0] * 1e6  # assumes uniform channels, like the reference
        nfreq = visdata.shape[0]
        prodstack = _prodstack(data)
        nprod = len(prodstack)
        ninput = len(data.index_map["input"])

        if self.telescope is not None and nprod == getattr(self.telescope, "nbase", -1):
            redundancy = np.asarray(self.telescope.redundancy)
        elif nprod == ninput * (ninput + 1) / 2:
            redundancy = np.ones(nprod)
        else:
            raise ValueError("Unexpected number of products")

        nsamp = int(self.ndays * dt * df) * redundancy
        std = self.recv_temp / np.sqrt(nsamp)

        if self.add_noise:
            noise = random.complex_normal(size=(nfreq, nprod, ntime), scale=std[np.newaxis, :, np.newaxis], rng=self.rng)
            auto = prodstack["input_a"] == prodstack["input_b"]
            # autos are real with twice the variance (noise.py:270-277)
            visdata[:, auto] = visdata[:, auto] + (np.sqrt(2) * noise[:, auto].real).astype(visdata.real.dtype)
            visdata[:, ~auto] = visdata[:, ~auto] + noise[:, ~auto].astype(visdata.dtype)
            data.vis[:] = visdata
        if self.set_weights:
            data.weight[:] = (1.0 / std[:, np.newaxis] ** 2)[np.newaxis]
        return data


class SampleNoise(_RandomTask):
    """Draw complex-Wishart distributed samples around the expected visibilities (``noise.py:287-374``).

    The input must be the full triangle of products.

    Attributes
    ----------
    sample_frac : float
        Multiplies the number of samples in each measurement.
    set_weights : bool
        Set the weights to the appropriate values.
    """

    sample_frac = 1.0
    set_weights = True
    _config_names = ("sample_frac", "set_weights")

    def process(self, data_exp):
        data_exp.redistribute("freq")
        nfeed = len(data_exp.index_map["input"])
        vis_data = data_exp.vis[:]
        weight = data_exp.weight[:]
        if vis_data.shape[1] != nfeed * (nfeed + 1) // 2:
            raise ValueError("SampleNoise needs the full triangle of products")
        if isinstance(data_exp, containers.SiderealStream):
            dt = 240 * (data_exp.ra[1] - data_exp.ra[0]) * STELLAR_S
        else:
            dt = data_exp.time[1] - data_exp.time[0]
        iu = np.triu_indices(nfeed)
        diag = np.array([_cmap(i, i, nfeed) for i in range(nfeed)])
        pa, pb = iu
        for fi in range(vis_data.shape[0]):
            df = data_exp.index_map["freq"]["width"][fi] * 1e6
            nsamp = int(self.sample_frac * dt * df)
            for ti in range(vis_data.shape[2]):
                mat = np.zeros((nfeed, nfeed), dtype=vis_data.dtype)
                mat[iu] = vis_data[fi, :, ti]  # upper triangle in product order, noise.py:350-357
                mat = mat + np.triu(mat, 1).T.conj()
                samp = random.complex_wishart(mat, nsamp, rng=self.rng) / nsamp
                vis_data[fi, :, ti] = samp[iu]
            if self.set_weights:
                autos = vis_data[fi][diag].real  # [nfeed, ntime]
                fac = nsamp**0.5 / autos
                weight[fi] = weight[fi] * (fac[pa] * fac[pb])  # apply_gain on the weights, noise.py:366-372
        data_exp.vis[:] = vis_data
        data_exp.weight[:] = weight
        return data_exp

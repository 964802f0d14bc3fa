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


class ReceiverTemperature(ContainerTask):
    """Add a basic receiver temperature term into the data (``noise.py:21-45``).

    An uncorrelated, frequency and time independent offset: it only reaches the auto-correlations.  The random
    fluctuations that go with it come from :class:`SampleNoise` afterwards.

    Attributes
    ----------
    recv_temp : float
        The receiver temperature in Kelvin.
    """

    recv_temp = 0.0
    _config_names = ("recv_temp",)

    def process(self, data):
        pairs = _prodstack(data)
        autos = np.flatnonzero(pairs["input_a"] == pairs["input_b"])
        vis = data.vis[:]
        vis[:, autos] += self.recv_temp
        data.vis[:] = vis
        return data


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

    def _samples_per_product(self, data, nprod):
        """Independent samples behind each (stacked) product of ``data``: ``int(ndays * dt * df) * redundancy``.

        The radiometer count uses the first channel's width for the whole band, as the reference does
        (``noise.py:243-244``).  A stream that matches the telescope's unique baselines is taken to be redundancy
        stacked; the full triangle of its inputs counts every product once; anything else is refused (``:250-256``).
        """
        cadence = _sample_interval(data)
        bandwidth = data.index_map["freq"]["width"][0] * 1e6
        ninput = len(data.index_map["input"])
        tel = self.telescope
        if tel is not None and nprod == getattr(tel, "nbase", -1):
            copies = np.asarray(tel.redundancy, dtype=np.float64)
        elif 2 * nprod == ninput * (ninput + 1):
            copies = np.ones(nprod)
        else:
            raise ValueError("Unexpected number of products")
        return int(self.ndays * cadence * bandwidth) * copies

    def process(self, data):
        """Add the noise to ``data`` in place and/or set its weights to ``1 / sigma^2`` (``noise.py:219-284``)."""
        data.redistribute("freq")
        pairs = _prodstack(data)
        sigma = self.recv_temp / np.sqrt(self._samples_per_product(data, len(pairs)))  # [nprod]

        if self.add_noise:
            vis = data.vis[:]
            nfreq, nprod, ntime = vis.shape
            # one draw for the whole stream, standard deviation per product (the reference's single call, :261-265)
            draw = random.complex_normal(size=(nfreq, nprod, ntime), scale=sigma[None, :, None], rng=self.rng)
            is_auto = np.asarray(pairs["input_a"]) == np.asarray(pairs["input_b"])
            # an auto-correlation is real: all of sigma^2 goes into its real part (sqrt(2) x the real half, :270-273)
            draw[:, is_auto] = np.sqrt(2) * draw[:, is_auto].real
            np.add(vis, draw, out=vis, casting="same_kind")  # summed in float64, rounded once to the stream's dtype
            data.vis[:] = vis
        if self.set_weights:
            data.weight[:] = np.broadcast_to((1.0 / sigma**2)[None, :, None], data.weight.shape)
        return data


def _sample_interval(data):
    """Seconds between samples: 240 s of sidereal time per degree of RA for a sidereal stream, else the time step."""
    if isinstance(data, containers.SiderealStream):
        return 240 * (data.ra[1] - data.ra[0]) * STELLAR_S
    return data.time[1] - data.time[0]


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
        dt = _sample_interval(data_exp)
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

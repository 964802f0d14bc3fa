"""Minimal container duck-types for the m-mode path.

These mirror the parts of the reference containers the path touches
(``draco/core/containers.py``): dataset names, axis order and dtypes, ``index_map``,
``attrs`` and the ``axes_from`` / ``attrs_from`` constructor protocol.  They are not a
re-implementation of caput's memh5 containers [3P]; HDF5 I/O, MPI distribution and
selections are out of scope.

* :class:`SiderealStream`  ``containers.py:489-593``  vis c64 ``[freq, stack, ra]``, weight f32
* :class:`MModes`          ``containers.py:1167-1193`` vis c128 ``[m, msign, freq, stack]``, weight f64
* :class:`Map`             ``containers.py:470-486`` (+ cora ``Map`` [3P]) map f64 ``[freq, pol, pixel]``

A dataset may live on the host (ndarray), on the GPU (torch tensor) or both; tasks of
this package hand device-resident containers to each other so that a
``MModeTransform -> DirtyMapMaker`` chain never crosses PCIe.  ``ds[:]`` always gives
NumPy (copying back once if needed).
"""

from __future__ import annotations

import numpy as np


class Dataset:
    """An array that is valid on the host, on a device, or both."""

    def __init__(self, host=None, dev=None, attrs=None, pending=None):
        assert host is not None or dev is not None
        self._host = host
        self._dev_t = dev
        self._pending = pending
        self.attrs = dict(attrs or {})

    # The device copy may still be in the making on another stream: `pending` is then what orders a reader behind
    # that work -- an object with `done()` (has the producer finished? never blocks) and `order()` (make the CALLER's
    # current stream wait for it: a stream wait, not a host wait).  EVERY access to the data orders the stream that is
    # current at that moment, until the producer has finished -- readers on different streams (the caller's, a side
    # context, the fill stream, the RCCL stream) are each ordered, whichever comes first.  A producer that returns
    # early (the map-maker's side-stream SHT) does not hold up work that never looks at its output (the next sidereal
    # day's transform and solves).
    @property
    def _dev(self):
        p = self._pending
        if p is not None:
            if p.done():
                self._pending = None
            else:
                p.order()
        return self._dev_t

    @_dev.setter
    def _dev(self, t):
        self._dev_t = t

    # -- shape/dtype without forcing a copy
    @property
    def shape(self):
        return tuple(self._host.shape) if self._host is not None else tuple(self._dev_t.shape)

    @property
    def dtype(self):
        if self._host is not None:
            return self._host.dtype
        import torch

        return np.dtype(
            {
                torch.complex64: np.complex64,
                torch.complex128: np.complex128,
                torch.float32: np.float32,
                torch.float64: np.float64,
                torch.int32: np.int32,
                torch.uint8: np.uint8,
            }[self._dev_t.dtype]
        )

    @property
    def on_device(self):
        return self._dev_t is not None

    def host(self) -> np.ndarray:
        if self._host is None:
            self._host = self._dev.detach().cpu().numpy()
        return self._host

    def device(self, ctx):
        """Device tensor on ``ctx`` (uploading the host copy the first time)."""
        if self._dev is None or self._dev.device != ctx.device:
            self._dev = ctx.to_device(self.host())
        return self._dev

    def set_device(self, t):
        """Make a device tensor the (only) valid copy."""
        self._dev = t
        self._host = None

    def __getitem__(self, key):
        return self.host()[key]

    def __setitem__(self, key, value):
        h = self.host()
        h[key] = value
        self._dev = None  # host copy is now the authority

    def __array__(self, dtype=None, copy=None):
        a = self.host()
        return a if dtype is None else a.astype(dtype)

    @property
    def local_array(self):  # MPIArray compatibility (single process)
        return self.host()


def _freq_map(freq):
    """Build the structured ``index_map['freq']`` (fields ``centre``, ``width``)."""
    if freq is None:
        return None
    f = np.asarray(freq)
    if f.dtype.names and "centre" in f.dtype.names:
        return f
    out = np.zeros(f.shape[0], dtype=[("centre", np.float64), ("width", np.float64)])
    out["centre"] = f
    out["width"] = np.abs(np.diff(f)).mean() if f.shape[0] > 1 else 1.0
    return out


class ContainerBase:
    """Axes + datasets + attrs; ``axes_from`` / ``attrs_from`` copy like caput does."""

    _axes: tuple = ()
    _dataset_spec: dict = {}

    def __init__(self, axes_from=None, attrs_from=None, comm=None, distributed=True, allocate=True, **axes):
        self.index_map = {}
        self.index_attrs = {}
        self.attrs = {}
        self.comm = comm
        self.reverse_map = {}
        axes = {k: v for k, v in axes.items() if v is not None}
        if axes_from is not None:
            for ax in self._axes:
                if ax in axes_from.index_map and ax not in axes:
                    self.index_map[ax] = axes_from.index_map[ax]
            if "stack" in getattr(axes_from, "reverse_map", {}):
                self.reverse_map["stack"] = axes_from.reverse_map["stack"]
        for ax, val in axes.items():
            if val is None:
                continue
            if ax == "freq":
                val = _freq_map(val)
            elif isinstance(val, (int, np.integer)):
                val = np.arange(int(val))
            self.index_map[ax] = np.asarray(val)
        for ax in self.index_map:
            self.index_attrs[ax] = {}
        if attrs_from is not None:
            self.attrs.update(attrs_from.attrs)
            for ax, a in getattr(attrs_from, "index_attrs", {}).items():  # (caput: axis attributes travel too)
                if ax in self.index_attrs:
                    self.index_attrs[ax].update(a)
        self.datasets = {}
        for name, spec in self._dataset_spec.items():
            missing = [a for a in spec["axes"] if a not in self.index_map]
            if missing:
                raise ValueError(f"{type(self).__name__}: axes {missing} are not defined")
            shape = tuple(len(self.index_map[a]) for a in spec["axes"])
            if allocate:  # allocate=False: the producing task attaches device tensors instead
                self.datasets[name] = Dataset(host=np.zeros(shape, dtype=spec["dtype"]), attrs={"axis": spec["axes"]})
                src = getattr(attrs_from, "datasets", {}).get(name) if attrs_from is not None else None
                if src is not None:  # dataset attributes too, but never the receiving dataset's own axis tuple
                    self.datasets[name].attrs.update({k: v for k, v in src.attrs.items() if k != "axis"})

    def copy(self, shared=()):
        """A copy with its own datasets, except those named in ``shared``, which are the SAME objects (data and
        attributes) as the original's -- caput's ``copy(shared=...)``.  Container and axis attributes are copied."""
        import copy as _copy

        out = type(self).__new__(type(self))
        out.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("datasets", "attrs", "index_attrs", "index_map", "reverse_map")})
        out.index_map = dict(self.index_map)
        out.reverse_map = dict(self.reverse_map)
        out.attrs = _copy.deepcopy(self.attrs)
        out.index_attrs = _copy.deepcopy(self.index_attrs)
        out.datasets = {}
        for name, ds in self.datasets.items():
            if name in shared:
                out.datasets[name] = ds
            else:
                out.datasets[name] = Dataset(host=np.array(ds.host(), copy=True), attrs=_copy.deepcopy(ds.attrs))
        return out

    def redistribute(self, axis):  # single process: nothing to move
        return None

    # ---- stage outputs on disk (the reference's `save` / `output_root` task parameters write every task's output
    # container to HDF5, and a pipeline resumes by loading it, doc/pipeline_params.yaml:12-13, examples/test.yaml:10-13
    # [caput, 3P]).  h5py is not in this image: the same round trip through one .npz per container.
    def save(self, path):
        """Write axes, attributes, reverse maps and datasets (copied back from the device if need be) to ``path``."""
        blob = {"__class__": np.array(type(self).__name__)}
        for k, v in self.index_map.items():
            blob["index_map/" + k] = np.asarray(v)
        for k, v in self.reverse_map.items():
            blob["reverse_map/" + k] = np.asarray(v)
        for k, v in self.attrs.items():
            blob["attrs/" + k] = np.asarray(v)
        for k, ds in self.datasets.items():
            blob["dataset/" + k] = np.asarray(ds.host())
            for ak, av in ds.attrs.items():
                if ak != "axis":
                    blob[f"dataset_attrs/{k}/{ak}"] = np.asarray(av)
        for ax, a in getattr(self, "index_attrs", {}).items():
            for ak, av in a.items():
                blob[f"index_attrs/{ax}/{ak}"] = np.asarray(av)
        with open(path, "wb") as fh:  # (np.savez would append ".npz" to a bare name)
            np.savez(fh, **blob)

    @classmethod
    def load(cls, path):
        """Rebuild a container written by :meth:`save` (host resident)."""
        with np.load(path, allow_pickle=False) as z:
            name = str(z["__class__"])
            if name != cls.__name__ and cls is not ContainerBase:
                raise TypeError(f"{path} holds a {name}, not a {cls.__name__}")
            klass = cls if cls is not ContainerBase else next(c for c in _all_containers() if c.__name__ == name)
            out = klass.__new__(klass)
            out.index_map, out.reverse_map, out.attrs, out.datasets, out.comm = {}, {}, {}, {}, None
            out.index_attrs = {}
            if hasattr(klass, "_optional_spec"):
                out._dataset_spec = dict(klass._dataset_spec)
            for key in z.files:
                kind, _, k = key.partition("/")
                if kind == "index_map":
                    out.index_map[k] = z[key]
                elif kind == "reverse_map":
                    out.reverse_map[k] = z[key]
                elif kind == "attrs":
                    v = z[key]
                    out.attrs[k] = v.item() if v.ndim == 0 else v
                elif kind == "dataset":
                    if k not in out._dataset_spec and k in getattr(klass, "_optional_spec", {}):
                        out._dataset_spec[k] = klass._optional_spec[k]
                    out.datasets[k] = Dataset(host=z[key])
            # second pass: attributes of the datasets and of the axes (the `axis` tuple of a dataset comes from its spec)
            item = lambda v: v.item() if v.ndim == 0 else v  # noqa: E731
            for k, ds in out.datasets.items():
                if k in out._dataset_spec:
                    ds.attrs["axis"] = out._dataset_spec[k]["axes"]
            for ax in out.index_map:
                out.index_attrs.setdefault(ax, {})
            for key in z.files:
                parts = key.split("/")
                if parts[0] == "dataset_attrs" and parts[1] in out.datasets:
                    out.datasets[parts[1]].attrs["/".join(parts[2:])] = item(z[key])
                elif parts[0] == "index_attrs":
                    out.index_attrs.setdefault(parts[1], {})["/".join(parts[2:])] = item(z[key])
        return out

    def dataset_shape(self, name):
        return tuple(len(self.index_map[a]) for a in self._dataset_spec[name]["axes"])

    def attach(self, name, dev_tensor, pending=None):
        """Attach a device tensor as dataset ``name`` (shape/dtype checked against the spec).  ``pending``: see
        :class:`Dataset` -- orders every reader behind the producer until the producer has finished."""
        spec = self._dataset_spec[name]
        if tuple(dev_tensor.shape) != self.dataset_shape(name):
            raise ValueError(f"{name}: shape {tuple(dev_tensor.shape)} != {self.dataset_shape(name)}")
        ds = Dataset(dev=dev_tensor, attrs={"axis": spec["axes"]}, pending=pending)
        if ds.dtype != np.dtype(spec["dtype"]):
            raise ValueError(f"{name}: dtype {ds.dtype} != {np.dtype(spec['dtype'])}")
        self.datasets[name] = ds
        return ds

    def __getitem__(self, name):
        return self.datasets[name]

    def __contains__(self, name):
        return name in self.datasets


class _FreqMixin:
    @property
    def freq(self):
        return self.index_map["freq"]["centre"]


class _VisMixin:
    @property
    def vis(self):
        return self.datasets["vis"]

    @property
    def weight(self):
        return self.datasets["vis_weight"]


class SiderealStream(ContainerBase, _FreqMixin, _VisMixin):
    """``vis [freq, stack, ra]`` complex64 + ``vis_weight`` float32 (``containers.py:489-593``)."""

    _axes = ("freq", "stack", "ra", "prod", "input")
    _dataset_spec = {
        "vis": {"axes": ["freq", "stack", "ra"], "dtype": np.complex64},
        "vis_weight": {"axes": ["freq", "stack", "ra"], "dtype": np.float32},
    }
    _optional_spec = {"input_flags": {"axes": ["input", "ra"], "dtype": np.float32}}  # containers.py:525-530

    def add_dataset(self, name, allocate=True):
        self._dataset_spec = dict(self._dataset_spec)
        self._dataset_spec[name] = self._optional_spec[name]
        if allocate:
            self.datasets[name] = Dataset(host=np.zeros(self.dataset_shape(name), dtype=self._optional_spec[name]["dtype"]))

    @property
    def input_flags(self):
        return self.datasets["input_flags"]

    @property
    def is_stacked(self):
        st = self.index_map.get("stack")
        return st is not None and st.dtype.names is not None and "prod" in st.dtype.names and len(st) != len(self.index_map.get("prod", st))

    def __init__(self, ra=None, stack=None, prod=None, input=None, reverse_map_stack=None, **kwargs):
        if isinstance(ra, (int, np.integer)):
            ra = np.linspace(0.0, 360.0, int(ra), endpoint=False)  # containers.py:407-410
        if stack is None and prod is not None and kwargs.get("axes_from") is None:
            stack = len(prod)  # VisContainer default: one stack entry per product
        super().__init__(ra=ra, stack=stack, prod=prod, input=input, **kwargs)
        if reverse_map_stack is not None:
            self.reverse_map["stack"] = reverse_map_stack

    @property
    def ra(self):
        return self.index_map["ra"]

    @property
    def prodstack(self):
        """The representative input pair of every stack entry, conjugation applied (``containers.py:211-229``)."""
        prod = self.index_map["prod"]
        stack = self.index_map.get("stack")
        if stack is None or stack.dtype.names is None or "prod" not in stack.dtype.names:
            return prod
        t = prod[stack["prod"]].copy()
        conj = stack["conjugate"].astype(bool)
        t["input_a"] = np.where(conj, prod[stack["prod"]]["input_b"], prod[stack["prod"]]["input_a"])
        t["input_b"] = np.where(conj, prod[stack["prod"]]["input_a"], prod[stack["prod"]]["input_b"])
        return t


class MContainer(ContainerBase):
    """m-mode containers: axes ``m``, ``msign`` and the ``oddra`` attribute (``containers.py:422-467``)."""

    def __init__(self, mmax=None, oddra=None, **kwargs):
        if mmax is not None:
            kwargs["m"] = int(mmax) + 1
        kwargs["msign"] = np.array(["+", "-"])
        super().__init__(**kwargs)
        if oddra is not None:
            self.attrs["oddra"] = bool(oddra)
        elif "oddra" not in self.attrs:
            self.attrs["oddra"] = False

    @property
    def mmax(self) -> int:
        return int(self.index_map["m"][-1])

    @property
    def oddra(self) -> bool:
        return bool(self.attrs["oddra"])


class MModes(MContainer, _FreqMixin, _VisMixin):
    """``vis [m, msign, freq, stack]`` complex128 + ``vis_weight`` float64 (``containers.py:1167-1193``)."""

    _axes = ("m", "msign", "freq", "stack", "prod", "input")
    _dataset_spec = {
        "vis": {"axes": ["m", "msign", "freq", "stack"], "dtype": np.complex128},
        "vis_weight": {"axes": ["m", "msign", "freq", "stack"], "dtype": np.float64},
    }


class HybridVisStream(ContainerBase, _FreqMixin, _VisMixin):
    """NS-beamformed visibilities: ``vis [pol, freq, ew, el, ra]`` c64, ``vis_weight [pol, freq, ew, ra]`` f32
    (``containers.py:1389-1428``; the optional datasets of the reference are not carried)."""

    _axes = ("pol", "freq", "ew", "el", "ra")
    _dataset_spec = {
        "vis": {"axes": ["pol", "freq", "ew", "el", "ra"], "dtype": np.complex64},
        "vis_weight": {"axes": ["pol", "freq", "ew", "ra"], "dtype": np.float32},
    }

    _optional_spec = {"dirty_beam": {"axes": ["pol", "freq", "ew", "el", "ra"], "dtype": np.float32}}  # containers.py:1409-1417

    def __init__(self, ra=None, **kwargs):
        if isinstance(ra, (int, np.integer)):
            ra = np.linspace(0.0, 360.0, int(ra), endpoint=False)
        self._dataset_spec = dict(type(self)._dataset_spec)
        super().__init__(ra=ra, **kwargs)

    def add_dataset(self, name, allocate=False):
        self._dataset_spec[name] = self._optional_spec[name]
        if allocate:
            self.datasets[name] = Dataset(host=np.zeros(self.dataset_shape(name), dtype=self._optional_spec[name]["dtype"]))

    @property
    def dirty_beam(self):
        return self.datasets["dirty_beam"]


class VisGridStream(ContainerBase, _FreqMixin, _VisMixin):
    """Visibilities on the (pol, ew, ns) baseline grid: ``vis [pol, freq, ew, ns, ra]`` c64, ``vis_weight`` f32, optional
    ``redundancy [pol, ew, ns, ra]`` int32 (``containers.py:1245-1299``)."""

    _axes = ("pol", "freq", "ew", "ns", "ra")
    _dataset_spec = {
        "vis": {"axes": ["pol", "freq", "ew", "ns", "ra"], "dtype": np.complex64},
        "vis_weight": {"axes": ["pol", "freq", "ew", "ns", "ra"], "dtype": np.float32},
    }
    _optional_spec = {"redundancy": {"axes": ["pol", "ew", "ns", "ra"], "dtype": np.int32}}

    def __init__(self, ra=None, **kwargs):
        if isinstance(ra, (int, np.integer)):
            ra = np.linspace(0.0, 360.0, int(ra), endpoint=False)
        self._dataset_spec = dict(type(self)._dataset_spec)
        super().__init__(ra=ra, **kwargs)

    def add_dataset(self, name, allocate=False):
        self._dataset_spec[name] = self._optional_spec[name]
        if allocate:
            self.datasets[name] = Dataset(host=np.zeros(self.dataset_shape(name), dtype=np.int32))

    @property
    def redundancy(self):
        if "redundancy" in self.datasets:
            return self.datasets["redundancy"]
        raise KeyError("Dataset 'redundancy' not initialised.")


class HybridVisMModes(MContainer, _FreqMixin, _VisMixin):
    """m-mode transformed hybrid visibilities: ``vis [m, msign, pol, freq, ew, el]`` **complex64**,
    ``vis_weight [m, msign, pol, freq, ew]`` **float32** (``containers.py:1550-1574``)."""

    _axes = ("m", "msign", "pol", "freq", "ew", "el")
    _dataset_spec = {
        "vis": {"axes": ["m", "msign", "pol", "freq", "ew", "el"], "dtype": np.complex64},
        "vis_weight": {"axes": ["m", "msign", "pol", "freq", "ew"], "dtype": np.float32},
    }


class RingMap(ContainerBase, _FreqMixin):
    """Multi-frequency ring maps: ``map [beam, pol, freq, ra, el]``, ``weight [pol, freq, ra, el]`` float64
    (``containers.py:1577-1653``); optional ``dirty_beam``, ``dirty_beam_power`` via :meth:`add_dataset`."""

    _axes = ("beam", "pol", "freq", "ra", "el", "ew")
    _dataset_spec = {
        "map": {"axes": ["beam", "pol", "freq", "ra", "el"], "dtype": np.float64},
        "weight": {"axes": ["pol", "freq", "ra", "el"], "dtype": np.float64},
    }
    _optional_spec = {
        "dirty_beam": {"axes": ["beam", "pol", "freq", "ra", "el"], "dtype": np.float64},
        "dirty_beam_power": {"axes": ["beam", "pol", "freq", "el"], "dtype": np.float64},
        "rms": {"axes": ["pol", "freq", "ra"], "dtype": np.float64},  # containers.py:1643-1650
    }

    def __init__(self, ra=None, **kwargs):
        if isinstance(ra, (int, np.integer)):
            ra = np.linspace(0.0, 360.0, int(ra), endpoint=False)
        self._dataset_spec = dict(type(self)._dataset_spec)
        super().__init__(ra=ra, **kwargs)

    def add_dataset(self, name, allocate=False):
        self._dataset_spec[name] = self._optional_spec[name]
        if allocate:
            self.datasets[name] = Dataset(host=np.zeros(self.dataset_shape(name), dtype=np.float64))

    @property
    def map(self):
        return self.datasets["map"]

    @property
    def weight(self):
        return self.datasets["weight"]

    @property
    def dirty_beam(self):
        return self.datasets["dirty_beam"]

    @property
    def dirty_beam_power(self):
        return self.datasets["dirty_beam_power"]

    @property
    def rms(self):
        return self.datasets["rms"]


class SVDSpectrum(ContainerBase):
    """``spectrum [m, singularvalue]`` float64: the per-m SVD spectrum of MModes (``containers.py:2589-2607``)."""

    _axes = ("m", "singularvalue")
    _dataset_spec = {"spectrum": {"axes": ["m", "singularvalue"], "dtype": np.float64}}

    @property
    def spectrum(self):
        return self.datasets["spectrum"]


class SVDModes(MContainer, _VisMixin):
    """SVD-projected m-modes: ``vis [m, mode]`` complex128, ``vis_weight [m, mode]`` float64, ``nmode [m]`` int32
    (``containers.py:1196-1234``): per m the modes of every frequency packed back to back, ``nmode[m]`` of them."""

    _axes = ("m", "msign", "mode")
    _dataset_spec = {
        "vis": {"axes": ["m", "mode"], "dtype": np.complex128},
        "vis_weight": {"axes": ["m", "mode"], "dtype": np.float64},
        "nmode": {"axes": ["m"], "dtype": np.int32},
    }

    def __init__(self, mode=None, **kwargs):
        super().__init__(mode=mode, **kwargs)

    @property
    def nmode(self):
        return self.datasets["nmode"]


class KLModes(SVDModes):
    """KL-projected m-modes, same layout (``containers.py:1237-1246``)."""


class Map(ContainerBase, _FreqMixin):
    """``map [freq, pol, pixel]`` float64, HEALPix RING (``containers.py:470-486``, cora ``Map`` [3P])."""

    _axes = ("freq", "pol", "pixel")
    _dataset_spec = {"map": {"axes": ["freq", "pol", "pixel"], "dtype": np.float64}}

    def __init__(self, nside=None, polarisation=True, **kwargs):
        if nside is not None:
            kwargs["pixel"] = 12 * int(nside) ** 2
        if "pol" not in kwargs:
            kwargs["pol"] = np.array(["I", "Q", "U", "V"] if polarisation else ["I"])
        super().__init__(**kwargs)

    @property
    def map(self):
        return self.datasets["map"]

    @property
    def nside(self):
        return int(round((len(self.index_map["pixel"]) // 12) ** 0.5))


def _all_containers():
    found, todo = [], [ContainerBase]
    while todo:
        c = todo.pop()
        for sub in c.__subclasses__():
            found.append(sub)
            todo.append(sub)
    return found

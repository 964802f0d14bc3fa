"""Container-local loader for the reference's pure NumPy/SciPy kernels.

TEST INFRASTRUCTURE ONLY.  Nothing under ``draco_amd/`` may import this file.

The reference (``/root/reference/draco``) cannot be imported as-is in the build
container: caput, driftscan (``drift``), cora, mpi4py, healpy ... are not
installed.  Those packages only supply the *framework*
around the arithmetic (task base classes, MPI arrays, HDF5 containers); the
arithmetic of the m-mode path itself is plain NumPy/SciPy inside the reference
source files.  This module serves permissive placeholder modules for the absent
third-party *imports* so that ``draco.analysis.transform`` and
``draco.analysis.mapmaker`` can be imported from where they lie and their
numerical functions executed unmodified to generate golden vectors
(``oracle/gen_golden.py``).  The only real semantics patched in are

* ``caput.algorithms.invert_no_zero``  -> NumPy (1/x where x != 0 else 0)
* ``caput.config.Property(default=..)`` -> returns the default
* ``scipy.linalg.solve(sym_pos=True)``  -> ``assume_a="pos"`` (the kwarg was
  removed from SciPy; reference ``mapmaker.py:272,277`` still passes it)

It refuses to do anything unless ``/root/reference`` exists, writes nothing
there (bytecode writing disabled) and never travels to the GPU box in any
executed form: only the ``.npz`` fixtures it helps to generate do.
"""

from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"

_STUB_ROOTS = {
    "caput",
    "drift",
    "cora",
    "mpi4py",
    "healpy",
    "skimage",
    "pywt",
    "skyfield",
    "h5py",
    "pyfftw",
    "ch_util",
    "chimedb",
}
_STUB_EXACT = {"draco.util._fast_tools", "draco.util.truncate"}


class _AnyMeta(type):
    """Metaclass whose classes answer every attribute and the ``|`` operator."""

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        sub = _AnyMeta(name, (), {})
        setattr(cls, name, sub)
        return sub

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls

    def __call__(cls, *a, **k):
        # used as decorator factory / type constructor at import time
        if len(a) == 1 and not k and isinstance(a[0], type | types.FunctionType):
            return a[0]
        return super().__call__()

    def __mro_entries__(cls, bases):  # pragma: no cover
        return (cls,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if full in sys.modules:
            return sys.modules[full]
        obj = _AnyMeta(name, (), {})
        setattr(self, name, obj)
        return obj


class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS or fullname in _STUB_EXACT:
            return importlib.machinery.ModuleSpec(fullname, _Loader(), is_package=True)
        return None


def _invert_no_zero(x, out=None):
    import numpy as np

    x = np.asarray(x)
    if x.ndim == 0:
        return type(x.item())(0) if x == 0 else 1.0 / x
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(x == 0, 0, 1.0 / np.where(x == 0, 1, x)).astype(
            x.dtype if x.dtype.kind in "fc" else np.float64
        )
    if out is not None:
        out[...] = r
        return out
    return r


def load_reference():
    """Return ``(transform, mapmaker)`` reference modules, imported under stubs."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "draco")):
        raise RuntimeError(
            f"{REFERENCE_ROOT} is not present: golden vectors can only be regenerated "
            "in the build container; use the committed fixtures in tests/golden/."
        )
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import importlib

    # real semantics the kernels need
    alg = importlib.import_module("caput.algorithms")
    alg.invert_no_zero = _invert_no_zero
    caput = importlib.import_module("caput")
    caput.algorithms = alg
    cfg = importlib.import_module("caput.config")

    def _prop(proptype=None, default=None, key=None):
        return default

    cfg.Property = _prop
    for nm in ("enum", "list_type", "utc_time", "float_in_range", "file_format", "logging_config"):
        setattr(cfg, nm, lambda *a, **k: None)
    caput.config = cfg

    tasklib_base = importlib.import_module("caput.pipeline.tasklib.base")

    class _Task:
        def __init__(self, *a, **k):
            pass

        class _Log:
            def debug(self, *a, **k):
                pass

            info = warning = error = debug

        log = _Log()

    tasklib_base.ContainerTask = _Task
    tasklib_base.MPILoggedTask = _Task
    tasklib_base.group_tasks = lambda *tasks: type("Grouped", (_Task,), {})
    tasklib = importlib.import_module("caput.pipeline.tasklib")
    tasklib.base = tasklib_base
    pipeline = importlib.import_module("caput.pipeline")
    pipeline.tasklib = tasklib
    caput.pipeline = pipeline

    # scipy.linalg.solve(sym_pos=True) was removed upstream
    import scipy.linalg as la

    if not getattr(la.solve, "_dmm_wrapped", False):
        _orig = la.solve

        def solve(a, b, *args, sym_pos=False, **kw):
            if sym_pos:
                kw.setdefault("assume_a", "pos")
            return _orig(a, b, *args, **kw)

        solve._dmm_wrapped = True
        la.solve = solve

    from draco.analysis import mapmaker, transform  # noqa: E402

    # tools.invert_no_zero was bound at import time through the stub; re-point it
    transform.tools.invert_no_zero = _invert_no_zero
    return transform, mapmaker

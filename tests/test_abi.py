"""The C-ABI library loads and exports every symbol include/draco_amd.h declares (no GPU needed)."""

import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "draco_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = ctypes.CDLL(os.environ.get("DRACO_AMD_LIBRARY") or os.path.join(ROOT, "draco_amd", "libdraco_amd.so"))
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in draco_amd.h but not exported"


def test_binding_matches_header():
    from draco_amd import _lib

    assert sorted(_lib.EXPORTED) == _declared()
    assert _lib.lib.dmm_version() == 100


def test_errors_without_gpu_are_loud():
    """No silent fallback: without a GPU the context refuses, with a message."""
    import torch

    from draco_amd import _lib

    if torch.cuda.is_available():
        return
    h = ctypes.c_void_p()
    rc = _lib.lib.dmm_ctx_create(0, ctypes.byref(h))
    assert rc != 0 and _lib.lib.dmm_last_error()
    import pytest

    from draco_amd.device import Context

    with pytest.raises(RuntimeError):
        Context(0)


def test_argument_errors():
    from draco_amd import _lib

    import pytest

    with pytest.raises(ValueError):
        _lib.check(_lib.lib.dmm_ctx_sync(None))
    with pytest.raises(ValueError):
        _lib.check(_lib.lib.dmm_mfft_pack(None, None, 1, 8, None, 4, 1, None))
    h = ctypes.c_void_p(1)  # a non-NULL ctx is not dereferenced before the argument checks
    with pytest.raises(ValueError, match="NULL argument"):
        _lib.check(_lib.lib.dmm_mfft_pack(h, None, 1, 8, None, 4, 1, None))


def test_beam_screen_coefficients_match_the_oracle_twin():
    """`dmm_beam_screen_coeffs` is a pure host function: the plane waves of the beam screens of a seed must be the
    ones the oracle's twin draws (oracle/synth.py::screen_coeffs) -- the hash, the integer wave numbers, the scaling."""
    import ctypes as C

    import numpy as np

    from draco_amd import _lib
    from oracle import synth as osyn

    for seed in (0, 77, 3005, 2**63 + 12345):
        ka = np.zeros(16, np.int32)
        kb = np.zeros(16, np.int32)
        cr = np.zeros(16)
        ci = np.zeros(16)
        vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        _lib.check(_lib.lib.dmm_beam_screen_coeffs(seed, vp(ka), vp(kb), vp(cr), vp(ci)))
        rka, rkb, rcr, rci = osyn.screen_coeffs(seed)
        assert np.array_equal(ka, rka.reshape(-1)) and np.array_equal(kb, rkb.reshape(-1))
        assert np.array_equal(cr, rcr.reshape(-1)) and np.array_equal(ci, rci.reshape(-1))
        assert np.all(np.abs(ka) <= 3) and np.all(np.abs(cr) <= 0.25)

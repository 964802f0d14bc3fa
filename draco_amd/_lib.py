"""ctypes binding of ``libdraco_amd.so`` (the C ABI in ``include/draco_amd.h``).

The binding is deliberately thin: argument marshalling and status -> exception
translation only.  If the shared library is missing this module raises at import: the
product path never falls back to a CPU implementation.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import os

# One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so (SONAME
# libamdhip64.so.7) and libdraco_amd.so is linked against the same SONAME, so importing
# torch FIRST makes the dynamic loader resolve our dependency to the runtime torch already
# loaded.  In the other order two runtimes coexist and the second one sees no device.
import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRACO_AMD_LIBRARY: load another build of the same library (the host-only sanitizer build, `make -C draco_amd/csrc
# asan`, for the CPU-side ABI tests); unset in normal use.
LIB_PATH = os.environ.get("DRACO_AMD_LIBRARY") or os.path.join(_HERE, "libdraco_amd.so")

DMM_C64, DMM_C128 = 0, 1
DMM_B_FULL, DMM_B_PACKED = 0, 1
DMM_E_ARG, DMM_E_UNSUPPORTED, DMM_E_NOMEM, DMM_E_STATE, DMM_E_COMM = -1, -2, -3, -4, -5
DMM_MAX_NRA = 8192


class DmmError(RuntimeError):
    """A call into libdraco_amd.so failed (HIP error or unsupported request)."""

    def __init__(self, code, msg):
        super().__init__(f"libdraco_amd status {code}: {msg}")
        self.code = code


class dmm_tile(C.Structure):
    _fields_ = [("b_off", C.c_int64), ("m", C.c_int32), ("f", C.c_int32)]


class dmm_gemv_desc(C.Structure):
    _fields_ = [("a_off", C.c_int64), ("x_off", C.c_int64), ("y_off", C.c_int64), ("nrow", C.c_int32), ("ncol", C.c_int32)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C draco_amd/csrc`.  draco_amd has no CPU fallback."
    )

lib = C.CDLL(LIB_PATH)

_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_SIGS = {
    "dmm_version": (C.c_int, []),
    "dmm_last_error": (C.c_char_p, []),
    "dmm_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "dmm_ctx_destroy": (_i, [_vp]),
    "dmm_ctx_set_stream": (_i, [_vp, _vp]),
    "dmm_ctx_sync": (_i, [_vp]),
    "dmm_ctx_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "dmm_ctx_get_counter": (_i, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "dmm_ctx_set_ml_diag": (_i, [_vp, _vp]),
    "dmm_ctx_set_ml_gram_cache": (_i, [_vp, _vp, _vp, _i64, _i]),
    "dmm_ctx_set_ml_basis": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i]),
    "dmm_ml_gram_cache_slots": (_i64, [_vp]),
    "dmm_ml_gram_cache_bytes": (_i64, [_vp]),
    "dmm_mmode_svd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp]),
    "dmm_timer_start": (_i, [_vp]),
    "dmm_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "dmm_mfft_pack": (_i, [_vp, _vp, _i64, _i, _vp, _i, _i, _vp]),
    "dmm_mmode_weight": (_i, [_vp, _vp, _i64, _i, _vp, _i, _vp]),
    "dmm_mifft_unpack": (_i, [_vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp]),
    "dmm_mrow_is_zero": (_i, [_vp, _vp, _i, _i64, _i, _i, C.POINTER(_i)]),
    "dmm_mask_mmode_weight": (_i, [_vp, _vp, _i, _i64, _i, _vp, _i, _i, _i, _i]),
    "dmm_expand_products": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dmm_collate_products": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dmm_solve_plan_create": (_i, [_vp, C.POINTER(dmm_tile), _i64, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "dmm_plan_destroy": (_i, [_vp]),
    "dmm_plan_b_bytes": (_i64, [_vp]),
    "dmm_dirty_run": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "dmm_dirty_run_multi": (_i, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i]),
    "dmm_wiener_workspace_bytes": (_i64, [_vp]),
    "dmm_wiener_run": (_i, [_vp, _vp, _vp, _vp, _d, _d, _vp, _vp]),
    "dmm_ml_workspace_bytes": (_i64, [_vp]),
    "dmm_ml_run": (_i, [_vp, _vp, _vp, _vp, _d, _d, _vp, _vp]),
    "dmm_project_run": (_i, [_vp, _vp, _vp, _vp]),
    "dmm_alm2map": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dmm_map2alm": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "dmm_ringmap_deconvolve": (_i, [_vp] + [_i] * 10 + [_vp] * 10),
    "dmm_analytic_beam_mmodes": (_i, [_vp] + [_i] * 6 + [_vp] * 6),
    "dmm_mmode_fill0": (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    "dmm_ringmap_window": (_i, [_vp, _i, _i, _i, _vp, _vp, C.POINTER(C.c_double), _vp]),
    "dmm_synth_beam_fill": (_i, [_vp, C.POINTER(dmm_tile), _i64, _i, _i, _i, _i, _i, C.c_uint64, _vp]),
    "dmm_calc_redundancy": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "dmm_vis_grid": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dmm_beamform_ns": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dmm_beamform_ew": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dmm_comm_unique_id": (_i, [_vp]),
    "dmm_comm_init": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "dmm_comm_destroy": (_i, [_vp]),
    "dmm_allgather_map": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "dmm_beam_screen_coeffs": (_i, [C.c_uint64, _vp, _vp, _vp, _vp]),
    "dmm_beam_screen_maps": (_i, [_vp, _i, _i, _d, _d, C.c_uint64, _d, _d, _d, _d, _vp, _vp, _vp, _vp, _i, _vp]),
    "dmm_beam_screen_pack": (_i, [_vp, _vp, _i, _i, C.POINTER(dmm_tile), _i64, _i, _i, _i, _i, _i, _i, _vp]),
    "dmm_gemv_batch": (_i, [_vp, _vp, _i, C.POINTER(dmm_gemv_desc), _i64, _vp, _vp]),
    "dmm_row_median": (_i, [_vp, _vp, _i64, _i64, _vp]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_SIGS)


def check(rc: int) -> None:
    """Translate a status code into an exception (0 = ok)."""
    if rc == 0:
        return
    msg = (lib.dmm_last_error() or b"").decode("utf-8", "replace")
    if rc == DMM_E_ARG:
        raise ValueError(f"libdraco_amd: {msg}")
    if rc == DMM_E_NOMEM:
        raise MemoryError(f"libdraco_amd: {msg}")
    raise DmmError(rc, msg)


_TILE_DTYPE = np.dtype([("b_off", np.int64), ("m", np.int32), ("f", np.int32)])
assert _TILE_DTYPE.itemsize == C.sizeof(dmm_tile)


_GEMV_DTYPE = np.dtype([("a_off", np.int64), ("x_off", np.int64), ("y_off", np.int64), ("nrow", np.int32), ("ncol", np.int32)])
assert _GEMV_DTYPE.itemsize == C.sizeof(dmm_gemv_desc)


def gemv_desc_array(a_off, x_off, y_off, nrow, ncol):
    """Pack parallel sequences into a ctypes array of ``dmm_gemv_desc``."""
    n = len(a_off)
    rec = np.empty(n, dtype=_GEMV_DTYPE)
    rec["a_off"], rec["x_off"], rec["y_off"], rec["nrow"], rec["ncol"] = a_off, x_off, y_off, nrow, ncol
    arr = (dmm_gemv_desc * max(n, 1))()
    if n:
        C.memmove(arr, rec.ctypes.data, rec.nbytes)
    return arr


def tile_array(ms, fs, offs):
    """Pack parallel sequences into a ctypes array of ``dmm_tile`` (a day has 10^5 tiles: no Python loop)."""
    n = len(ms)
    rec = np.empty(n, dtype=_TILE_DTYPE)
    rec["b_off"], rec["m"], rec["f"] = offs, ms, fs
    arr = (dmm_tile * n)()
    C.memmove(arr, rec.ctypes.data, rec.nbytes)
    return arr

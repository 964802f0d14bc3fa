"""Argument validation of the C ABI, on the CPU (no GPU call is reached).

Every entry point checks its arguments before it touches the context or the device, so a made-up non-NULL handle
is enough to drive the host-side checks: the tile-table validation loops of ``dmm_solve_plan_create`` /
``dmm_synth_beam_fill``, the size / enum / range tests of the transforms.  ``make -C draco_amd/csrc asan-test`` runs
this file (and ``test_abi.py``) against the host-sanitizer build of the library (ASan + UBSan on the host half).
"""

import ctypes as C

import pytest

from draco_amd import _lib

FAKE = C.c_void_p(0x1000)  # never dereferenced: every call below must fail its checks first
BUF = C.c_void_p(0x2000)


def _arg_error(rc, match):
    assert rc == _lib.DMM_E_ARG, rc
    msg = _lib.lib.dmm_last_error().decode()
    assert match in msg, msg
    with pytest.raises(ValueError, match=match):
        _lib.check(rc)


def test_plan_create_validates_the_tile_table():
    lib = _lib.lib
    out = C.c_void_p()
    good = _lib.tile_array([0, 1, 2], [0, 0, 1], [0, 1000, 2000])
    _arg_error(lib.dmm_solve_plan_create(None, good, 3, 4, 4, 5, 2, 3, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), "NULL argument")
    _arg_error(lib.dmm_solve_plan_create(FAKE, None, 3, 4, 4, 5, 2, 3, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), "NULL argument")
    _arg_error(lib.dmm_solve_plan_create(FAKE, good, 3, 0, 4, 5, 2, 3, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), "bad sizes")
    _arg_error(lib.dmm_solve_plan_create(FAKE, good, -1, 4, 4, 5, 2, 3, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), "bad sizes")
    _arg_error(lib.dmm_solve_plan_create(FAKE, good, 3, 4, 4, 5, 2, 3, 7, _lib.DMM_B_PACKED, C.byref(out)), "bad b_dtype")
    _arg_error(lib.dmm_solve_plan_create(FAKE, good, 3, 4, 4, 5, 2, 3, _lib.DMM_C64, 9, C.byref(out)), "bad b_layout")
    _arg_error(lib.dmm_solve_plan_create(FAKE, good, 3, 4000, 4, 5, 2, 3, _lib.DMM_C64, _lib.DMM_B_PACKED, C.byref(out)), "too large for the LDS")
    # a tile whose m is outside the data (n_m = 2), beyond lmax, a frequency outside the data, a negative offset
    _arg_error(lib.dmm_solve_plan_create(FAKE, good, 3, 4, 4, 5, 2, 2, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), "m=2 out of range")
    _arg_error(lib.dmm_solve_plan_create(FAKE, good, 3, 4, 4, 1, 2, 3, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), "m=2 out of range")
    _arg_error(lib.dmm_solve_plan_create(FAKE, good, 3, 4, 4, 5, 1, 3, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), "f=1 out of range")
    bad = _lib.tile_array([0, 1], [0, 0], [0, -16])
    _arg_error(lib.dmm_solve_plan_create(FAKE, bad, 2, 4, 4, 5, 2, 3, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), "negative b_off")
    assert out.value is None  # nothing was handed out
    # a long table: the loop walks every entry (the bad one is last)
    n = 50_000
    ms = [i % 6 for i in range(n)]
    fs = [i % 2 for i in range(n)]
    offs = [64 * i for i in range(n)]
    ms[-1] = 6
    _arg_error(lib.dmm_solve_plan_create(FAKE, _lib.tile_array(ms, fs, offs), n, 4, 4, 5, 2, 7, _lib.DMM_C128, _lib.DMM_B_PACKED, C.byref(out)), f"tile {n - 1}")
    assert lib.dmm_plan_b_bytes(None) == 0 and lib.dmm_plan_destroy(None) == 0


def test_synth_fill_and_run_calls():
    lib = _lib.lib
    good = _lib.tile_array([0, 3], [0, 1], [0, 512])
    _arg_error(lib.dmm_synth_beam_fill(FAKE, good, 2, 4, 4, 5, 5, _lib.DMM_B_FULL, 1, BUF), "bad b_dtype")
    _arg_error(lib.dmm_synth_beam_fill(FAKE, good, 2, 4, 4, 5, _lib.DMM_C64, 5, 1, BUF), "bad b_layout")
    _arg_error(lib.dmm_synth_beam_fill(FAKE, good, 2, 0, 4, 5, _lib.DMM_C64, _lib.DMM_B_FULL, 1, BUF), "bad sizes")
    _arg_error(lib.dmm_synth_beam_fill(FAKE, good, 2, 4, 4, 2, _lib.DMM_C64, _lib.DMM_B_FULL, 1, BUF), "bad tile 1")
    _arg_error(lib.dmm_synth_beam_fill(FAKE, good, 2, 4, 4, 5, _lib.DMM_C64, _lib.DMM_B_FULL, 1, None), "NULL argument")
    for fn in (lib.dmm_dirty_run, lib.dmm_project_run):
        _arg_error(fn(None, BUF, BUF, BUF, BUF), "NULL argument")
    _arg_error(lib.dmm_wiener_run(None, BUF, BUF, BUF, 1.0, 0.5, BUF, BUF), "NULL argument")
    _arg_error(lib.dmm_ml_run(None, BUF, BUF, BUF, 1e-4, 1e-3, BUF, BUF), "NULL argument")
    assert lib.dmm_wiener_workspace_bytes(None) <= 0 and lib.dmm_ml_workspace_bytes(None) <= 0


def test_transform_calls():
    lib = _lib.lib
    _arg_error(lib.dmm_mfft_pack(FAKE, BUF, -1, 8, BUF, 4, _lib.DMM_C128, BUF), "bad sizes")
    _arg_error(lib.dmm_mfft_pack(FAKE, BUF, 4, 0, BUF, 4, _lib.DMM_C128, BUF), "bad sizes")
    _arg_error(lib.dmm_mfft_pack(FAKE, BUF, 4, 8, BUF, 4, 3, BUF), "bad out_dtype")
    _arg_error(lib.dmm_mmode_weight(FAKE, BUF, 4, 0, BUF, 4, BUF), "bad sizes")
    _arg_error(lib.dmm_mmode_weight(FAKE, None, 4, 8, BUF, 4, BUF), "NULL argument")
    # inverse: limits must fit the m axis and the output length (transform.py:826-829)
    _arg_error(lib.dmm_mifft_unpack(FAKE, BUF, 5, 4, 16, 5, 4, BUF, BUF), "bad limits")   # m limit beyond the m axis
    _arg_error(lib.dmm_mifft_unpack(FAKE, BUF, 5, 4, 16, 3, 4, BUF, BUF), "bad limits")   # more -m than +m
    _arg_error(lib.dmm_mifft_unpack(FAKE, BUF, 5, 4, 6, 4, 3, BUF, BUF), "exceed nra")    # 2*4 > 6 samples
    n = C.c_int()
    _arg_error(lib.dmm_mrow_is_zero(FAKE, BUF, 5, 4, 5, 0, C.byref(n)), "bad index")
    _arg_error(lib.dmm_mrow_is_zero(FAKE, BUF, 5, 4, 1, 2, C.byref(n)), "bad index")
    _arg_error(lib.dmm_mask_mmode_weight(FAKE, BUF, -1, 4, 4, BUF, 0, 0, 0, 0), "bad sizes")
    _arg_error(lib.dmm_expand_products(FAKE, BUF, 2, -3, 4, 6, BUF, BUF, BUF, BUF), "bad sizes")


def test_sht_ringmap_svd_calls():
    lib = _lib.lib
    _arg_error(lib.dmm_alm2map(FAKE, BUF, 1, 4, 8, 9, 4, BUF), "bad sizes")
    _arg_error(lib.dmm_alm2map(FAKE, BUF, 1, 3, 8, 8, 4, BUF), "npol must be 1 or 4")
    _arg_error(lib.dmm_alm2map(FAKE, BUF, 1, 4, 8, 8, 12, BUF), "power of two")
    _arg_error(lib.dmm_map2alm(FAKE, BUF, 1, 4, 8, 8, 4, 99, BUF), "niter")
    _arg_error(lib.dmm_map2alm(FAKE, None, 1, 4, 8, 8, 4, 3, BUF), "NULL argument")
    _arg_error(lib.dmm_mmode_fill0(FAKE, BUF, BUF, -2, 10, BUF), "bad sizes")
    _arg_error(lib.dmm_mmode_fill0(FAKE, BUF, BUF, 2, 1 << 33, BUF), "do not fit")
    _arg_error(lib.dmm_mmode_svd(FAKE, BUF, BUF, 3, 0, 4, 1, 1, BUF, 0, 0.0, 0.0, 0.0, BUF, None, None), "bad sizes")
    _arg_error(lib.dmm_mmode_svd(FAKE, BUF, BUF, 3, 4, 4, 1, 1, BUF, 2, 0.0, 0.0, 0.0, BUF, None, None), "mode must be")
    _arg_error(lib.dmm_mmode_svd(FAKE, BUF, BUF, 3, 4, 4, 1, 1, BUF, 0, 0.0, 0.0, 0.0, BUF, BUF, None), "come together")
    _arg_error(lib.dmm_ringmap_window(FAKE, -1, 4, 4, BUF, BUF, None, BUF), "bad sizes")


def test_context_calls():
    lib = _lib.lib
    _arg_error(lib.dmm_ctx_set_option(FAKE, b"no_such_option", 1), "unknown option")
    v = C.c_int64()
    _arg_error(lib.dmm_ctx_get_counter(FAKE, b"no_such_counter", C.byref(v)), "unknown counter")
    _arg_error(lib.dmm_ctx_set_option(None, b"ml_eigen", 1), "NULL argument")
    _arg_error(lib.dmm_ctx_create(0, None), "ctx is NULL")
    assert lib.dmm_ctx_destroy(None) == 0
    ms = C.c_float()
    _arg_error(lib.dmm_timer_stop(FAKE, None), "NULL argument")


def test_collective_entry_points_validate_their_arguments():
    """dmm_comm_* / dmm_allgather_map (the map all-gather of SURVEY 8e) check their arguments before RCCL is even loaded."""
    lib = _lib.lib
    out = C.c_void_p()
    ident = (C.c_char * 128)()
    _arg_error(lib.dmm_comm_unique_id(None), "NULL argument")
    _arg_error(lib.dmm_comm_init(None, ident, 0, 1, C.byref(out)), "NULL argument")
    _arg_error(lib.dmm_comm_init(FAKE, None, 0, 1, C.byref(out)), "NULL argument")
    _arg_error(lib.dmm_comm_init(FAKE, ident, 2, 2, C.byref(out)), "rank 2 of 2")
    _arg_error(lib.dmm_comm_init(FAKE, ident, 0, 0, C.byref(out)), "rank 0 of 0")
    _arg_error(lib.dmm_allgather_map(FAKE, None, BUF, 8, BUF), "NULL argument")
    _arg_error(lib.dmm_allgather_map(FAKE, FAKE, None, 8, BUF), "NULL argument")
    _arg_error(lib.dmm_allgather_map(FAKE, FAKE, BUF, -1, BUF), "negative count")
    assert lib.dmm_comm_destroy(None) == 0  # destroying nothing is fine


def test_round4_entries_check_their_arguments():
    """`dmm_dirty_run_multi`, `dmm_ctx_set_ml_diag` and the new options / counters: NULL handles and bad counts are refused
    before anything is dereferenced."""
    lib = _lib.lib
    PA = C.c_void_p * 2
    pv = PA(BUF, BUF)
    _arg_error(lib.dmm_dirty_run_multi(None, BUF, pv, pv, pv, 2), "NULL argument")
    _arg_error(lib.dmm_dirty_run_multi(FAKE, None, pv, pv, pv, 2), "NULL argument")
    _arg_error(lib.dmm_dirty_run_multi(FAKE, BUF, None, pv, pv, 2), "NULL argument")
    _arg_error(lib.dmm_dirty_run_multi(FAKE, BUF, pv, pv, pv, 0), "nday = 0")
    _arg_error(lib.dmm_dirty_run_multi(FAKE, C.c_void_p(0x2008), pv, pv, pv, 2), "16-byte aligned")
    _arg_error(lib.dmm_dirty_run_multi(FAKE, BUF, pv, pv, PA(BUF, BUF), 2), "share their alm")
    _arg_error(lib.dmm_dirty_run_multi(FAKE, BUF, PA(BUF, None), pv, PA(BUF, C.c_void_p(0x3000)), 2), "NULL array of day 1")
    _arg_error(lib.dmm_ctx_set_ml_diag(None, BUF), "ctx is NULL")
    _arg_error(lib.dmm_ctx_set_option(None, b"ml_null", 1), "NULL")
    assert _lib.DMM_E_COMM == -5


def test_resident_product_entries_check_their_arguments():
    """`dmm_ctx_set_ml_gram_cache`, `dmm_ctx_set_ml_basis` and their sizing functions: NULL handles and inconsistent
    arrays are refused before anything is dereferenced; a NULL plan has no slots."""
    lib = _lib.lib
    _arg_error(lib.dmm_ctx_set_ml_gram_cache(None, BUF, BUF, 4, 0), "ctx is NULL")
    _arg_error(lib.dmm_ctx_set_ml_basis(None, BUF, BUF, BUF, 4, 448, 0), "ctx is NULL")
    assert lib.dmm_ml_gram_cache_slots(None) == 0 and lib.dmm_ml_gram_cache_bytes(None) == 0


def test_round5_option_names_check_their_context():
    """Options are refused on a NULL context before anything is dereferenced."""
    lib = _lib.lib
    for name in (b"sht_synth_form", b"ml_reduce"):
        _arg_error(lib.dmm_ctx_set_option(None, name, 1), "NULL")

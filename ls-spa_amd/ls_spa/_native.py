"""ctypes binding of liblsspa_hip.so (include/lsspa.h).

There is deliberately no CPU fallback here: if the HIP library cannot be loaded or
no gfx950 device is present, every entry point raises ``LSSPANativeError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

OK = 0
F64, F32 = 0, 1
HOST, DEVICE = 0, 1
INFO_NOT_PD = 1
KERNEL_CLASSES = ("gather", "chol_diag", "chol_panel", "strip", "lift", "stats", "gram", "error", "comm", "small_p")

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.normpath(os.path.join(_PKG_DIR, "..", "lib", "liblsspa_hip.so"))


class LSSPANativeError(RuntimeError):
    """The HIP engine is missing, failed to load, or a call into it failed."""


_lib = None

_i32, _i64, _dbl, _vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p
_pd = C.POINTER(C.c_double)
_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); mirrors include/lsspa.h one to one
SIGNATURES = {
    "lsspa_abi_version": (C.c_int, []),
    "lsspa_last_error": (C.c_char_p, [_vp]),
    "lsspa_create": (C.c_int, [_i32, C.POINTER(_vp)]),
    "lsspa_destroy": (C.c_int, [_vp]),
    "lsspa_set_stream": (C.c_int, [_vp, _vp]),
    "lsspa_synchronize": (C.c_int, [_vp]),
    "lsspa_reduce": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _dbl, _i32, _i32]),
    "lsspa_reduce_timing": (C.c_int, [_vp, _pd]),
    "lsspa_reduce_partial": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32]),
    "lsspa_reduce_buffer": (C.c_int, [_vp, C.POINTER(_vp), _pi64]),
    "lsspa_reduce_finish": (C.c_int, [_vp, _i64, _dbl]),
    "lsspa_set_reduced": (C.c_int, [_vp, _i32, _pd, _pd, _dbl, _i32, _pd, _pd, _i32, _pd, _pd, _dbl]),
    "lsspa_get_problem": (C.c_int, [_vp, _pi32, _pi32, _pi32, _pd]),
    "lsspa_get_gram": (C.c_int, [_vp, _pd, _pd, _pd, _pd]),
    "lsspa_full_fit": (C.c_int, [_vp, _pd, _pd, _pi32]),
    "lsspa_get_factors": (C.c_int, [_vp, _pd, _pd, _pd, _pd]),
    "lsspa_lift_batch": (C.c_int, [_vp, _pi32, _i32, _i32, _pd, _i32]),
    "lsspa_lift_launch": (C.c_int, [_vp, _pi32, _i32, _i32, _pi32]),
    "lsspa_lift_collect": (C.c_int, [_vp, _i32, _i32, _i32, _pd, _i32]),
    "lsspa_lift_collect_chunks": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32]),
    "lsspa_lift_discard": (C.c_int, [_vp, _i32]),
    "lsspa_set_lanes": (C.c_int, [_vp, _i32]),
    "lsspa_get_info": (C.c_int, [_vp, _pi32]),
    "lsspa_get_info_collected": (C.c_int, [_vp, _pi32]),
    "lsspa_get_sum_deviation": (C.c_int, [_vp, _pd]),
    "lsspa_stats_reset": (C.c_int, [_vp]),
    "lsspa_stats_pending": (C.c_int, [_vp, C.POINTER(_vp), _pi64]),
    "lsspa_stats_merge": (C.c_int, [_vp]),
    "lsspa_stats_get": (C.c_int, [_vp, _pi64, _pd, _pd]),
    "lsspa_stats_set": (C.c_int, [_vp, _i64, _pd, _pd]),
    "lsspa_history_enable": (C.c_int, [_vp, _i64]),
    "lsspa_history_get": (C.c_int, [_vp, _pi64, _pd]),
    "lsspa_history_append": (C.c_int, [_vp, _pd, _i64]),
    "lsspa_error_draws": (C.c_int, [_vp, _pd, _i64, _i64, _i64]),
    "lsspa_error_buffer": (C.c_int, [_vp, C.POINTER(_vp), _pi64]),
    "lsspa_error_quantiles": (C.c_int, [_vp, _pd, _pd]),
    "lsspa_error_running_enable": (C.c_int, [_vp, C.c_uint64]),
    "lsspa_error_advance": (C.c_int, [_vp, _i64, _i64]),
    "lsspa_error_running_draws": (C.c_int, [_vp, _i64]),
    "lsspa_error_quantiles_enqueue": (C.c_int, [_vp, _i32]),
    "lsspa_error_check_enqueue": (C.c_int, [_vp, _i64, _i32]),
    "lsspa_error_result": (C.c_int, [_vp, _i32, _i32, _pi32, _pd, _pd, _pd, _pi64]),
    "lsspa_group_collect": (C.c_int, [_vp, _i32, _i32, _pi32, _pi32, _pi64, _i64, _pi64, _pi32]),
    "lsspa_error_state_get": (C.c_int, [_vp, _pd, _pd]),
    "lsspa_error_state_set": (C.c_int, [_vp, _pd, _pd]),
    "lsspa_error_xi": (C.c_int, [_vp, C.c_uint64, _i64, _i64, _i64, _pd]),
    "lsspa_profile_enable": (C.c_int, [_vp, _i32]),
    "lsspa_profile_get": (C.c_int, [_vp, _i32, _pd, _pi64]),
    "lsspa_profile_reset": (C.c_int, [_vp]),
    "lsspa_set_flags": (C.c_int, [_vp, _i32]),
    "lsspa_set_precision": (C.c_int, [_vp, _i32]),
    "lsspa_debug_fail_alloc": (C.c_int, [_vp, _i32]),
    "lsspa_debug_pack_from": (C.c_int, [_vp, _i32]),
    "lsspa_host_argsort_rows": (C.c_int, [_pd, _i64, _i32, _pi32, C.POINTER(C.c_uint8), _i32, _pi64]),
    "lsspa_sampler_create": (C.c_int, [_i32, _i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _dbl, _i64, _i32, _i64,
                                      _i64, _i32, _i32, _i32, C.POINTER(_vp)]),
    "lsspa_sampler_take": (C.c_int, [_vp, _i64, _pi32, _i64, _pi64, _pi64, _pi64, _pi64, _pi64]),
    "lsspa_sampler_destroy": (C.c_int, [_vp]),
    "lsspa_debug_set_r2": (C.c_int, [_vp, _dbl]),
    "lsspa_debug_check_perms": (C.c_int, [_pi32, _i32, _i32, _i32]),
    "lsspa_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "lsspa_comm_init": (C.c_int, [_vp, C.POINTER(C.c_uint8), _i32, _i32]),
    "lsspa_comm_destroy": (C.c_int, [_vp]),
    "lsspa_comm_info": (C.c_int, [_vp, _pi32, _pi32]),
    "lsspa_stats_allreduce": (C.c_int, [_vp]),
    "lsspa_reduce_allreduce": (C.c_int, [_vp]),
    "lsspa_error_allreduce": (C.c_int, [_vp]),
    "lsspa_comm_sum_i64": (C.c_int, [_vp, _pi64, _i32]),
    "lsspa_comm_allgather": (C.c_int, [_vp, _pd, _i64, _pd]),
    "lsspa_mfma_probe": (C.c_int, [_vp, _pd, _pd, _pd, _i32]),
    "lsspa_debug_factor": (C.c_int, [_vp, _pi32, _pd, _pd, _pd, _pi32, _pi32, _pi32]),
}


def library_path() -> str:
    return _LIB_PATH


def _explicit_hip_runtime():
    """``LSSPA_HIP_RUNTIME=/path/to/libamdhip64.so``: bind the engine to that HIP runtime instead of the one the
    dynamic loader finds (the system ROCm).  One process must not hold two HIP runtimes; a host application that
    bundles its own (a PyTorch-ROCm wheel does) and is imported AFTER this package would load a second one.  Either
    import that application first -- the engine then shares its runtime, nothing to set -- or name its runtime
    here.  The package itself never looks for PyTorch."""
    path = os.environ.get("LSSPA_HIP_RUNTIME")
    if not path:
        return
    try:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as exc:
        raise LSSPANativeError(f"LSSPA_HIP_RUNTIME={path}: {exc}") from exc


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise LSSPANativeError(
            f"{_LIB_PATH} is missing: build it with `python ls-spa_amd/build.py` "
            "(needs hipcc; there is no CPU fallback)")
    _explicit_hip_runtime()
    try:
        lib = C.CDLL(_LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 not found
        raise LSSPANativeError(f"cannot load {_LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise LSSPANativeError(f"{_LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.lsspa_abi_version() != 1:
        raise LSSPANativeError("liblsspa_hip.so ABI version mismatch: rebuild it")
    _lib = lib
    return lib


def dptr(a):
    """double* of a C-contiguous float64 array (or NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_pd)


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_pi32)

"""ctypes mirrors of include/noahmp_hip.h, built from abi_spec.py.

Only plumbing lives here: structure definitions, the loader of the in-tree HIP
shared library, and helpers that pack numpy arrays / torch tensors into a
``noahmp_step_args`` block.  There is NO CPU fallback: if the HIP library is
missing, :func:`load_library` raises.
"""
import ctypes as C
import os

import numpy as np

from .abi_spec import STEP_FIELDS, TABLE_FIELDS, ERROR_CODES, NSNOW, WTABLE_FIELDS

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnoahmp_hip.so")

MEM_HOST, MEM_DEVICE = 0, 1
SORT_VEG, SORT_SNOW, SORT_SNOW_FIRST, SORT_TAIR, SORT_COST = 1, 2, 4, 8, 16
HALO_TCP, HALO_RCCL = 0, 1


def _step_ctype(kind):
    return {"i": C.c_int32, "f": C.c_float, "pf": C.c_void_p, "pi": C.c_void_p}[kind]


class StepArgs(C.Structure):
    """noahmp_step_args (include/noahmp_hip.h); mirrors the noahmplsm argument list, drv:11-44."""
    _fields_ = [(n, _step_ctype(k)) for n, k, lev, io, ln in STEP_FIELDS]


class WtableArgs(C.Structure):
    """noahmp_wtable_args; mirrors the WTABLE_mmf_noahmp argument list, gw:14-22."""
    _fields_ = [(n, _step_ctype(k)) for n, k, lev, io, ln in WTABLE_FIELDS]


WTABLE_INFO = {n: (k, lev, io) for n, k, lev, io, ln in WTABLE_FIELDS}
WTABLE_ARRAYS = [n for n, k, lev, io, ln in WTABLE_FIELDS if k in ("pf", "pi")]


FORCING_RECORD_FIELDS = ("t", "q", "u", "v", "p", "lw", "sw", "pcp", "fpar", "lai")


class ForcingRecord(C.Structure):
    """noahmp_forcing_record: one forcing file's planes as hrldas_input_read keeps them (netcdf_io:1228-1252)."""
    _fields_ = [(n, C.c_void_p) for n in FORCING_RECORD_FIELDS]


def _tbl_ctype(kind, shape):
    t = C.c_int32 if kind == "i" else C.c_float
    for d in shape:           # Fortran (a,b) -> C [b][a]: wrap fastest dim first
        t = t * d
    return t


class Tables(C.Structure):
    """noahmp_tables: image of the reference's module tables (lsm:43-103, 215-259, 417-424)."""
    _fields_ = [(n, _tbl_ctype(k, s)) for n, k, s, src in TABLE_FIELDS]


class Status(C.Structure):
    _fields_ = [("code", C.c_int32), ("i", C.c_int32), ("j", C.c_int32),
                ("n_land", C.c_int32), ("n_glacier", C.c_int32), ("n_skipped", C.c_int32),
                ("kernel_ms", C.c_float)]


FIELD_INFO = {n: (k, lev, io) for n, k, lev, io, ln in STEP_FIELDS}
ARRAY_FIELDS = [n for n, k, lev, io, ln in STEP_FIELDS if k in ("pf", "pi")]


def nlev(lev, nsoil, nk_atm=2):
    return {None: 1, "atm": nk_atm, "soil": nsoil, "snow": NSNOW, "snso": NSNOW + nsoil}[lev]


def tables_to_dict(t):
    """Tables -> {name: numpy array in Fortran index order (e.g. saim[veg, month])}."""
    out = {}
    for n, k, s, src in TABLE_FIELDS:
        v = getattr(t, n)
        if s:
            a = np.ctypeslib.as_array(v).copy()
            out[n] = a.T.copy() if len(s) > 1 else a
        else:
            out[n] = v
    return out


def tables_from_dict(d):
    t = Tables()
    for n, k, s, src in TABLE_FIELDS:
        if s:
            dt = np.int32 if k == "i" else np.float32
            a = np.asarray(d[n], dtype=dt)
            assert a.shape == tuple(s), (n, a.shape, s)
            src_arr = np.ascontiguousarray(a.T) if len(s) > 1 else np.ascontiguousarray(a)
            C.memmove(C.addressof(getattr(t, n)), src_arr.ctypes.data, src_arr.nbytes)
        else:
            setattr(t, n, int(d[n]) if k == "i" else float(d[n]))
    return t


_lib = None


def load_library(path=None):
    """Load the in-tree HIP engine.  Fails loudly when it is missing (no CPU fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "HIP engine %s not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; two HIP runtimes in one process
    # cannot both own the device.  Importing torch first makes the loader resolve this library's
    # libamdhip64.so.7 dependency to the copy torch already mapped (same SONAME).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    lib.noahmp_hip_abi_version.restype = C.c_int
    lib.noahmp_hip_nsoil.restype = C.c_int
    lib.noahmp_hip_nsoil.argtypes = []
    lib.noahmp_hip_sizeof_step_args.restype = C.c_size_t
    lib.noahmp_hip_sizeof_tables.restype = C.c_size_t
    lib.noahmp_hip_device_count.restype = C.c_int
    lib.noahmp_hip_set_device.argtypes = [C.c_int]
    lib.noahmp_hip_set_tables.argtypes = [C.POINTER(Tables)]
    lib.noahmp_hip_step.argtypes = [C.POINTER(StepArgs), C.c_int, C.c_void_p, C.POINTER(Status)]
    lib.noahmp_hip_fetch.argtypes = [C.POINTER(StepArgs)]
    lib.noahmp_hip_step_async.argtypes = [C.POINTER(StepArgs), C.c_void_p]
    lib.noahmp_hip_sync.argtypes = [C.POINTER(Status), C.POINTER(C.c_int)]
    lib.noahmp_hip_sync_timing.argtypes = [C.POINTER(C.c_float), C.c_int]
    lib.noahmp_hip_sync_step_timing.argtypes = [C.POINTER(C.c_float), C.c_int]
    lib.noahmp_hip_sync_counts.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.noahmp_hip_fetch_cost.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
    lib.noahmp_hip_fetch_cost.restype = C.c_long
    lib.noahmp_hip_init.argtypes = [C.POINTER(StepArgs), C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Status)]
    lib.noahmp_hip_forcing_prep.argtypes = [C.POINTER(StepArgs), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_float, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.POINTER(Status)]
    lib.noahmp_hip_forcing_interpolate.argtypes = [C.POINTER(StepArgs), C.POINTER(ForcingRecord), C.POINTER(ForcingRecord),
                                                   C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Status)]
    lib.noahmp_hip_forcing_interpolate_prep.argtypes = [C.POINTER(StepArgs), C.POINTER(ForcingRecord), C.POINTER(ForcingRecord), C.c_int, C.c_int,
                                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                                        C.POINTER(C.c_float), C.c_int, C.c_void_p, C.POINTER(Status)]
    lib.noahmp_hip_gather_fields.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                             C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.noahmp_hip_output_fields.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    lib.noahmp_hip_scatter_fields.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.noahmp_hip_sorted_exchange.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.noahmp_hip_sort_columns.argtypes = [C.POINTER(StepArgs), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int64), C.c_void_p]
    lib.noahmp_hip_sort_staleness.argtypes = [C.POINTER(StepArgs), C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
    lib.noahmp_hip_sort_staleness_async.argtypes = [C.POINTER(StepArgs), C.c_int, C.c_void_p, C.c_void_p]
    lib.noahmp_hip_sort_staleness_result.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.noahmp_hip_sort_set_band.argtypes = [C.c_void_p]
    lib.noahmp_hip_sort_set_veg_order.argtypes = [C.POINTER(C.c_int32), C.c_int]
    lib.noahmp_hip_permute_step_arrays.argtypes = [C.POINTER(StepArgs), C.POINTER(StepArgs), C.c_void_p, C.c_void_p]
    lib.noahmp_hip_scatter_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.noahmp_hip_stream_sync.argtypes = [C.c_void_p]
    lib.noahmp_hip_declination.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.noahmp_hip_declination.restype = C.c_float
    lib.noahmp_hip_wtable_mmf.argtypes = [C.POINTER(WtableArgs), C.c_int, C.c_void_p, C.POINTER(Status)]
    lib.noahmp_hip_wtable_mmf_async.argtypes = [C.POINTER(WtableArgs), C.c_void_p]
    lib.noahmp_hip_wtable_lateral_async.argtypes = [C.POINTER(WtableArgs), C.c_void_p, C.c_void_p]
    lib.noahmp_hip_wtable_columns_async.argtypes = [C.POINTER(WtableArgs), C.c_void_p, C.c_void_p]
    lib.noahmp_hip_halo_init.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
    lib.noahmp_hip_exchange_halo.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int, C.c_void_p]
    lib.noahmp_hip_groundwater_init.argtypes = [C.POINTER(WtableArgs), C.c_int, C.c_int, C.c_void_p, C.POINTER(Status)]
    lib.noahmp_hip_sizeof_wtable_args.restype = C.c_size_t
    lib.noahmp_hip_malloc.argtypes = [C.c_size_t]
    lib.noahmp_hip_malloc.restype = C.c_void_p
    lib.noahmp_hip_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.noahmp_hip_free.argtypes = [C.c_void_p]
    lib.noahmp_hip_free.restype = None
    lib.noahmp_hip_jit_compile_check.argtypes = [C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]
    lib.noahmp_hip_jit_cache_info.argtypes = [C.POINTER(C.c_int32)]
    lib.noahmp_hip_jit_cache_info.restype = C.c_char_p
    lib.noahmp_hip_jit_source_hash.restype = C.c_uint64
    lib.noahmp_hip_set_option.argtypes = [C.c_char_p, C.c_int]
    lib.noahmp_hip_debug_copy_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.noahmp_hip_debug_copy_stats.restype = None
    lib.noahmp_hip_error_string.argtypes = [C.c_int]
    lib.noahmp_hip_error_string.restype = C.c_char_p
    lib.noahmp_hip_last_error.restype = C.c_char_p
    lib.noahmp_hip_finalize.restype = None
    if lib.noahmp_hip_sizeof_step_args() != C.sizeof(StepArgs):
        raise RuntimeError("ABI drift: sizeof(noahmp_step_args) %d != ctypes %d"
                           % (lib.noahmp_hip_sizeof_step_args(), C.sizeof(StepArgs)))
    if lib.noahmp_hip_sizeof_wtable_args() != C.sizeof(WtableArgs):
        raise RuntimeError("ABI drift: sizeof(noahmp_wtable_args)")
    if lib.noahmp_hip_sizeof_tables() != C.sizeof(Tables):
        raise RuntimeError("ABI drift: sizeof(noahmp_tables)")
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "noahmp_hip_abi_version", "noahmp_hip_nsoil", "noahmp_hip_sizeof_step_args", "noahmp_hip_sizeof_tables",
    "noahmp_hip_device_count", "noahmp_hip_set_device", "noahmp_hip_set_tables",
    "noahmp_hip_step", "noahmp_hip_fetch", "noahmp_hip_step_async", "noahmp_hip_sync", "noahmp_hip_sync_timing", "noahmp_hip_sync_step_timing", "noahmp_hip_sync_counts", "noahmp_hip_fetch_cost", "noahmp_hip_init", "noahmp_hip_forcing_prep", "noahmp_hip_forcing_interpolate", "noahmp_hip_forcing_interpolate_prep", "noahmp_hip_declination", "noahmp_hip_gather_fields", "noahmp_hip_output_fields", "noahmp_hip_scatter_fields", "noahmp_hip_scatter_chunk", "noahmp_hip_scatter_chunk_of", "noahmp_hip_index_width", "noahmp_hip_sorted_exchange", "noahmp_hip_sort_columns", "noahmp_hip_sort_set_band", "noahmp_hip_sort_set_veg_order", "noahmp_hip_sort_staleness", "noahmp_hip_sort_staleness_async", "noahmp_hip_sort_staleness_result", "noahmp_hip_permute_step_arrays", "noahmp_hip_scatter_plan", "noahmp_hip_stream_sync", "noahmp_hip_wtable_mmf", "noahmp_hip_wtable_mmf_async", "noahmp_hip_wtable_lateral_async", "noahmp_hip_wtable_columns_async", "noahmp_hip_groundwater_init", "noahmp_hip_halo_init", "noahmp_hip_exchange_halo", "noahmp_hip_halo_finalize", "noahmp_hip_halo_selftest_rccl", "noahmp_hip_sizeof_wtable_args", "noahmp_hip_malloc", "noahmp_hip_memcpy", "noahmp_hip_free", "noahmp_hip_jit_compile_check", "noahmp_hip_jit_cache_info", "noahmp_hip_jit_source_hash", "noahmp_hip_set_option", "noahmp_hip_debug_live_host_registrations", "noahmp_hip_debug_copy_stats", "noahmp_hip_error_string",
    "noahmp_hip_last_error", "noahmp_hip_finalize",
]


def error_name(code):
    return ERROR_CODES.get(code, ("unknown", ""))[0]

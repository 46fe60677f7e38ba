"""
ctypes binding of libbfg_mi355.so (include/bfg_mi355.h).

There is NO CPU fallback: if the HIP library cannot be loaded, or no GPU is
visible, every product entry point raises.  (The CPU oracle under oracle/ is
test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

from . import _build

_i64 = C.c_int64
_dbl = C.c_double
_vp = C.c_void_p

BFG_OK = 0
BFG_MAX_DIM = 6
BFG_MAX_EXTRA = 3
BFG_ND_MAX_OUTER = 12            # z, M and up to 10 p_keys axes on the N-dimensional row path (csrc/bfg_ndtable.hpp)
BFG_TABLE_LOG_VALUES = 1

VARIANT_AUTO, VARIANT_SCATTER_WAVE, VARIANT_SCATTER_QUARTER, VARIANT_TILE_LDS = 0, 1, 2, 3
VARIANTS = {"auto": VARIANT_AUTO, "scatter_wave": VARIANT_SCATTER_WAVE,
            "scatter_quarter": VARIANT_SCATTER_QUARTER, "tile_lds": VARIANT_TILE_LDS}

WARN_Z_RANGE, WARN_M_RANGE, WARN_R_RANGE = 1, 2, 4
SHELL_OUT_IS_ZERO, SHELL_OUT_OVERWRITE, SHELL_REUSE_PLAN = 1, 2, 4

# every symbol include/bfg_mi355.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "bfg_abi_version", "bfg_status_string", "bfg_last_error", "bfg_device_count",
    "bfg_ctx_create", "bfg_ctx_destroy", "bfg_ctx_set_stream", "bfg_ctx_synchronize", "bfg_ctx_device_info",
    "bfg_dev_malloc", "bfg_dev_free", "bfg_memcpy_h2d", "bfg_memcpy_d2h", "bfg_dev_memset_zero",
    "bfg_table_create", "bfg_table_destroy", "bfg_table_eval",
    "bfg_spline_create", "bfg_spline_destroy",
    "bfg_paint_shell", "bfg_baryonify_offsets", "bfg_regrid_shell", "bfg_reduce_absmax_sum",
    "bfg_baryonify_snapshot", "bfg_deposit_grid", "bfg_paint_grid", "bfg_baryonify_grid_offsets", "bfg_regrid_grid",
    "bfg_build_displacement_table", "bfg_baryonify_snapshot_strided", "bfg_deposit_grid_strided",
    "bfg_stats_reset", "bfg_stats_read", "bfg_timing_enable", "bfg_timing_select", "bfg_timing_read",
    "bfg_comm_unique_id", "bfg_comm_init", "bfg_comm_destroy", "bfg_comm_info",
    "bfg_allreduce_f64", "bfg_allreduce_f64_begin", "bfg_comm_wait", "bfg_reduce_scatter_f64", "bfg_allgather_f64",
    "bfg_reduce_scatter_f64_begin", "bfg_paint_shell_sliced", "bfg_baryonify_offsets_sliced",
    "bfg_disc_enumerate_count", "bfg_disc_enumerate", "bfg_map_add_values", "bfg_offsets_add_displacements",
    "bfg_copy_to_mapped_host", "bfg_shell_slice_cuts",
    "bfg_ndtable_create", "bfg_ndtable_destroy", "bfg_ndtable_rows", "bfg_ndtable_read",
    "bfg_regrid_band_rings", "bfg_regrid_shell_bands", "bfg_plan_reuses",
]
ABI_VERSION = 6
# bfg_slice_fn: int (*)(void *user, int slice, int n_slices, int64_t elem_begin, int64_t elem_end)
SLICE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64)
BFG_COMM_ID_BYTES = 128


class MassDefStruct(C.Structure):
    _fields_ = [("Omega_m", _dbl), ("Omega_l", _dbl), ("Omega_r", _dbl), ("w0", _dbl), ("h", _dbl),
                ("rho_crit0_h2", _dbl), ("Delta", _dbl), ("rho_type", C.c_int32), ("reserved", C.c_int32)]


class ShellArgs(C.Structure):
    _fields_ = [("nside", _i64), ("n_halo", _i64), ("d_catalog", _vp), ("cat_stride", C.c_int32),
                ("n_extra", C.c_int32), ("epsilon_max", _dbl), ("runner_md", MassDefStruct),
                ("model_md", MassDefStruct), ("model_epsilon_max", _dbl),
                ("rdelta_sampling", C.c_int32), ("include_pixel_size", C.c_int32),
                ("variant", C.c_int32), ("flags", C.c_int32)]


class SnapshotArgs(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("rdelta_sampling", C.c_int32), ("n_part", _i64), ("n_halo", _i64),
                ("L", _dbl), ("a", _dbl), ("d_part", _vp), ("d_halo", _vp), ("halo_stride", C.c_int32),
                ("n_extra", C.c_int32), ("epsilon_max", _dbl), ("runner_md", MassDefStruct),
                ("model_md", MassDefStruct), ("model_epsilon_max", _dbl)]


class GridArgs(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("rdelta_sampling", C.c_int32), ("n_halo", _i64), ("npix", C.c_int32),
                ("reserved", C.c_int32), ("a", _dbl), ("d_bins", _vp), ("d_halo", _vp), ("halo_stride", C.c_int32),
                ("n_extra", C.c_int32), ("epsilon_max", _dbl), ("runner_md", MassDefStruct),
                ("model_md", MassDefStruct), ("model_epsilon_max", _dbl), ("d_rmat", _vp)]


class Stats(C.Structure):
    _fields_ = [("pixel_updates", C.c_uint64), ("halos_out_of_table", C.c_uint64),
                ("pixels_out_of_table", C.c_uint64), ("halos_fallback4", C.c_uint64),
                ("warn_mask", C.c_uint32), ("halos_scatter_fallback", C.c_uint32)]


class BFGError(RuntimeError):
    pass


_lib = None


def so_path():
    return _build.SO


def load(build_if_missing=True):
    """Load the shared library (building it with hipcc first if it is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing:
        try:
            _build.build()
        except Exception as exc:  # no hipcc on this box: fall through to the prebuilt .so if present
            if not os.path.exists(_build.SO):
                raise BFGError(f"libbfg_mi355.so is missing and could not be built: {exc}") from exc
    if not os.path.exists(_build.SO):
        raise BFGError("libbfg_mi355.so not found; run `python -m baryonforge_amd._build`")
    # PyTorch-ROCm bundles its own libamdhip64.so.7.  Import torch FIRST so that this library
    # binds to the same HIP runtime instance (one runtime per process: torch's streams and
    # allocations are then valid handles inside libbfg_mi355.so).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # BFG_SO: load an alternative build of the library (A/B timing of kernel variants)
    L = C.CDLL(os.environ.get("BFG_SO", _build.SO))
    L.bfg_abi_version.restype = C.c_int
    L.bfg_status_string.restype = C.c_char_p
    L.bfg_status_string.argtypes = [C.c_int]
    L.bfg_last_error.restype = C.c_char_p
    L.bfg_device_count.argtypes = [C.POINTER(C.c_int)]
    L.bfg_ctx_create.argtypes = [C.c_int, _vp, C.POINTER(_vp)]
    L.bfg_ctx_destroy.argtypes = [_vp]
    L.bfg_ctx_set_stream.argtypes = [_vp, _vp]
    L.bfg_comm_unique_id.argtypes = [C.c_char_p, C.c_size_t]
    L.bfg_comm_init.argtypes = [_vp, C.c_char_p, C.c_size_t, C.c_int, C.c_int]
    L.bfg_comm_destroy.argtypes = [_vp]
    L.bfg_comm_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bfg_allreduce_f64.argtypes = [_vp, _vp, _i64]
    L.bfg_allreduce_f64_begin.argtypes = [_vp, _vp, _i64, C.POINTER(_i64)]
    L.bfg_reduce_scatter_f64_begin.argtypes = [_vp, _vp, _i64, C.POINTER(_i64)]
    L.bfg_comm_wait.argtypes = [_vp, _i64]
    L.bfg_reduce_scatter_f64.argtypes = [_vp, _vp, _i64]
    L.bfg_allgather_f64.argtypes = [_vp, _vp, _i64]
    L.bfg_ctx_synchronize.argtypes = [_vp]
    L.bfg_ctx_device_info.argtypes = [_vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(_i64)]
    L.bfg_dev_malloc.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
    L.bfg_dev_free.argtypes = [_vp, _vp]
    L.bfg_memcpy_h2d.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.bfg_memcpy_d2h.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.bfg_dev_memset_zero.argtypes = [_vp, _vp, C.c_size_t]
    L.bfg_table_create.argtypes = [_vp, C.c_int, C.POINTER(_i64), C.POINTER(C.POINTER(_dbl)),
                                   C.POINTER(_dbl), C.c_uint32, C.POINTER(_vp)]
    L.bfg_table_destroy.argtypes = [_vp, _vp]
    L.bfg_table_eval.argtypes = [_vp, _vp, _i64, C.POINTER(_dbl), C.POINTER(_dbl)]
    L.bfg_spline_create.argtypes = [_vp, C.c_int, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_vp)]
    L.bfg_spline_destroy.argtypes = [_vp, _vp]
    L.bfg_paint_shell.argtypes = [_vp, C.POINTER(ShellArgs), _vp, _vp, _vp]
    L.bfg_baryonify_snapshot.argtypes = [_vp, C.POINTER(SnapshotArgs), _vp, _vp]
    L.bfg_baryonify_snapshot_strided.argtypes = [_vp, C.POINTER(SnapshotArgs), _vp, _vp, _i64, _i64]
    L.bfg_paint_grid.argtypes = [_vp, C.POINTER(GridArgs), _vp, _vp]
    L.bfg_baryonify_grid_offsets.argtypes = [_vp, C.POINTER(GridArgs), _vp, _vp]
    L.bfg_regrid_grid.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp]
    L.bfg_deposit_grid.argtypes = [_vp, C.c_int, _i64, _vp, _vp, _dbl, C.c_int, C.c_int, _vp]
    L.bfg_deposit_grid_strided.argtypes = [_vp, C.c_int, _i64, _vp, _i64, _vp, _i64, _dbl, C.c_int, C.c_int, _vp]
    L.bfg_build_displacement_table.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_dbl), _vp, _vp, C.c_int,
                                               C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl), _vp,
                                               C.POINTER(C.c_int32)]
    L.bfg_baryonify_offsets.argtypes = [_vp, C.POINTER(ShellArgs), _vp, _vp, _vp]
    L.bfg_paint_shell_sliced.argtypes = [_vp, C.POINTER(ShellArgs), _vp, _vp, _vp, C.c_int, SLICE_FN, _vp]
    L.bfg_baryonify_offsets_sliced.argtypes = [_vp, C.POINTER(ShellArgs), _vp, _vp, _vp, C.c_int, SLICE_FN, _vp]
    L.bfg_regrid_shell.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp]
    L.bfg_copy_to_mapped_host.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t]
    L.bfg_regrid_band_rings.argtypes = []
    L.bfg_regrid_shell_bands.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_uint32]
    L.bfg_shell_slice_cuts.argtypes = [_i64, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(C.c_int)]
    L.bfg_ndtable_create.argtypes = [_vp, C.c_int, C.POINTER(_i64), C.POINTER(C.POINTER(_dbl)), _i64, C.POINTER(_dbl),
                                     C.POINTER(_dbl), C.POINTER(_vp)]
    L.bfg_ndtable_destroy.argtypes = [_vp, _vp]
    L.bfg_ndtable_rows.argtypes = [_vp, _vp, _vp, _i64, C.c_int, _vp]
    L.bfg_ndtable_read.argtypes = [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp]
    L.bfg_disc_enumerate_count.argtypes = [_vp, C.POINTER(ShellArgs), _vp, C.c_int, _vp]
    L.bfg_disc_enumerate.argtypes = [_vp, C.POINTER(ShellArgs), _vp, C.c_int, _vp, _vp, _vp, _vp]
    L.bfg_map_add_values.argtypes = [_vp, _vp, _vp, _vp, _i64]
    L.bfg_offsets_add_displacements.argtypes = [_vp, C.POINTER(ShellArgs), _vp, _vp, _vp, _vp, _i64, _vp]
    L.bfg_reduce_absmax_sum.argtypes = [_vp, _i64, _vp, C.POINTER(_dbl), C.POINTER(_dbl)]
    L.bfg_plan_reuses.argtypes = [_vp, C.POINTER(_i64)]
    L.bfg_stats_reset.argtypes = [_vp]
    L.bfg_stats_read.argtypes = [_vp, C.POINTER(Stats)]
    L.bfg_timing_enable.argtypes = [_vp, C.c_int]
    L.bfg_timing_select.argtypes = [_vp, C.c_uint]
    L.bfg_timing_read.argtypes = [_vp, C.c_int, C.POINTER(_dbl), C.POINTER(_i64)]
    for name in SYMBOLS:
        if name not in ("bfg_status_string", "bfg_last_error"):
            getattr(L, name).restype = C.c_int
    if L.bfg_abi_version() != ABI_VERSION:
        raise BFGError("libbfg_mi355.so ABI version mismatch")
    _lib = L
    return L


def check(status, what=""):
    if status != BFG_OK:
        L = load()
        msg = L.bfg_status_string(status).decode()
        detail = L.bfg_last_error().decode() if status in (-2, -6) else ""
        raise BFGError(f"{what}: {msg} ({status}) {detail}".strip())


def dptr(arr):
    """ctypes double* view of a C-contiguous float64 numpy array"""
    assert arr.dtype == np.float64 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.POINTER(_dbl))


def shell_slice_cuts(nside, offsets, n_slices):
    """element cuts [0, ..., all] of the slices bfg_paint_shell_sliced (offsets=False) / bfg_baryonify_offsets_sliced (True) report
    for this NSIDE and slice count (bfg_shell_slice_cuts; no GPU needed)"""
    cuts = (_i64 * 17)()
    n = C.c_int(0)
    check(load().bfg_shell_slice_cuts(int(nside), 1 if offsets else 0, int(n_slices), cuts, C.byref(n)), "bfg_shell_slice_cuts")
    return [int(cuts[k]) for k in range(n.value + 1)]

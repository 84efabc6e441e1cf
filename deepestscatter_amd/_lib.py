"""ctypes binding of libcloudtrace.so (include/cloudtrace.h).  There is NO fallback: if the
HIP library is missing this raises, and every call that needs a GPU fails with the library's
own error code when no device is present."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG = Path(__file__).resolve().parent
# CT_LIBRARY: another build of the same library beside the product's (python -m deepestscatter_amd.build --variant ...): the
# experiments build libcloudtrace_exp.so (tests/test_exchange.py, tools/) or an A/B build.  File name only, always in-tree.
LIB_PATH = PKG / os.path.basename(os.environ.get("CT_LIBRARY", "libcloudtrace.so"))

CT_ABI_VERSION = 1
CT_OK, CT_E_INVAL, CT_E_HIP, CT_E_NOMEM, CT_E_STATE, CT_E_NODEVICE, CT_E_RCCL = 0, -1, -2, -3, -4, -5, -6
CT_MODE_SUN_AND_SKY_ALL_SCATTER, CT_MODE_SUN_MULTIPLE_SCATTER, CT_MODE_SUN_SINGLE_SCATTER = 0, 1, 2
CT_EST_MARCH, CT_EST_DELTA = 0, 1
CT_BUF_MEAN, CT_BUF_M2, CT_BUF_FRAME, CT_BUF_SCREEN, CT_BUF_INSCATTER, CT_BUF_DENSITY = range(6)
CT_FLAG_NONE, CT_FLAG_SIMPLE_KERNEL, CT_FLAG_LIGHT_NORMALIZED, CT_FLAG_SPARSE_BRICKS, CT_FLAG_VMM_BRICKS = 0, 1, 2, 4, 8

# every symbol include/cloudtrace.h declares (tests check the library exports all of them)
EXPORTS = [
    "ct_create", "ct_destroy", "ct_last_error", "ct_set_stream", "ct_set_camera", "ct_render_subframe",
    "ct_accumulate", "ct_render_accumulate", "ct_render_accumulate_async", "ct_synchronize", "ct_copy_to_device_async", "ct_point_radiance_launch", "ct_generate_scatter_samples", "ct_collect_descriptors", "ct_reset", "ct_tonemap", "ct_tonemap_async", "ct_set_render_ahead", "ct_rendered_subframes", "ct_set_stop_when_converged", "ct_converged_at", "ct_is_converged", "ct_tonemap_buffer", "ct_is_converged_buffers", "ct_download", "ct_upload",
    "ct_buffer_bytes", "ct_copy_to_device", "ct_device_ptr", "ct_subframes", "ct_set_subframes", "ct_counters", "ct_kernel_time",
    "ct_debug_cdf_inversion", "ct_debug_math_selftest", "ct_debug_fetch_probe", "ct_debug_fetch_probe_ws", "ct_debug_track_lines", "ct_debug_touched_lines", "ct_debug_stats", "ct_debug_stats_ex", "ct_debug_suspended", "ct_debug_timeline", "ct_debug_invariants", "ct_debug_memory", "ct_debug_delta_grid", "ct_fetch_counters", "ct_calculate_camera_variables", "ct_quantize_volume", "ct_load_vdb", "ct_generate_mipmaps",
    "ct_tile_owner", "ct_make_procedural_cloud",
    "ct_group_create", "ct_group_destroy", "ct_group_last_error", "ct_group_size", "ct_group_handle", "ct_group_set_camera",
    "ct_group_render_accumulate", "ct_group_reset", "ct_group_merge", "ct_group_download", "ct_group_tonemap",
    "ct_group_is_converged", "ct_group_counters",
]


class CtScene(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("dims", C.c_uint32 * 3),
        ("density_host", C.c_void_p),
        ("cloud_size_m", C.c_float),
        ("mean_free_path_m", C.c_float),
        ("sample_step", C.c_float),
        ("mode", C.c_int32),
        ("estimator", C.c_int32),
        ("max_depth", C.c_uint32),
        ("light_direction", C.c_float * 3),
        ("light_color", C.c_float * 3),
        ("light_intensity", C.c_float),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("mie_host", C.c_void_p),
        ("chopped_mie_host", C.c_void_p),
        ("mie_count", C.c_uint32),
        ("device", C.c_int32),
        ("shard_index", C.c_uint32),
        ("shard_count", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class CtCounters(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64),
        ("box_hits", C.c_uint64),
        ("density_lookups", C.c_uint64),
        ("inscatter_lookups", C.c_uint64),
        ("scatter_events", C.c_uint64),
        ("depth_capped", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class CtFetchCounters(C.Structure):
    _fields_ = [("density_fetches", C.c_uint64), ("inscatter_fetches", C.c_uint64)]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class CloudTraceError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libcloudtrace error {code}: {message}")
        self.code = code
        self.message = message


_LIB = None


def load():
    """Load the HIP library; raises if it has not been built (python -m deepestscatter_amd.build)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not LIB_PATH.exists():
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build the HIP extension with `python -m deepestscatter_amd.build` "
            "(there is no CPU fallback)")
    L = C.CDLL(str(LIB_PATH))
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_float
    sig = {
        "ct_create": (i32, [C.POINTER(CtScene), C.POINTER(vp)]),
        "ct_destroy": (i32, [vp]),
        "ct_last_error": (C.c_char_p, [vp]),
        "ct_set_stream": (i32, [vp, vp]),
        "ct_set_camera": (i32, [vp, vp, vp, vp, vp]),
        "ct_render_subframe": (i32, [vp, u32, vp]),
        "ct_accumulate": (i32, [vp, u32, vp]),
        "ct_render_accumulate": (i32, [vp, u32, u32]),
        "ct_render_accumulate_async": (i32, [vp, u32, u32]),
        "ct_synchronize": (i32, [vp]),
        "ct_point_radiance_launch": (i32, [vp, vp, u32, u32, u32]),
        "ct_generate_scatter_samples": (i32, [vp, u32, u32, vp, vp]),
        "ct_collect_descriptors": (i32, [vp, vp, vp, u32, vp]),
        "ct_reset": (i32, [vp]),
        "ct_tonemap": (i32, [vp, f32, vp, C.POINTER(f32)]),
        "ct_tonemap_async": (i32, [vp, f32]),
        "ct_set_render_ahead": (i32, [vp, u32]),
        "ct_rendered_subframes": (i32, [vp, C.POINTER(u32)]),
        "ct_set_stop_when_converged": (i32, [vp, u32, u32]),
        "ct_converged_at": (i32, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64)]),
        "ct_is_converged": (i32, [vp, C.POINTER(i32), C.POINTER(C.c_uint64)]),
        "ct_tonemap_buffer": (i32, [vp, vp, f32, vp, C.POINTER(f32)]),
        "ct_is_converged_buffers": (i32, [vp, vp, vp, u32, C.POINTER(i32), C.POINTER(C.c_uint64)]),
        "ct_download": (i32, [vp, i32, vp, C.c_size_t]),
        "ct_upload": (i32, [vp, i32, vp, C.c_size_t]),
        "ct_buffer_bytes": (i32, [vp, i32, C.POINTER(C.c_size_t)]),
        "ct_copy_to_device": (i32, [vp, i32, vp, C.c_size_t]),
        "ct_copy_to_device_async": (i32, [vp, i32, vp, C.c_size_t]),
        "ct_debug_suspended": (i32, [vp, vp]),
        "ct_debug_timeline": (i32, [vp, vp, u32]),
        "ct_device_ptr": (i32, [vp, i32, C.POINTER(vp)]),
        "ct_subframes": (i32, [vp, C.POINTER(u32)]),
        "ct_set_subframes": (i32, [vp, u32]),
        "ct_counters": (i32, [vp, C.POINTER(CtCounters)]),
        "ct_kernel_time": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "ct_debug_cdf_inversion": (i32, [vp, u32, u32, vp]),
        "ct_debug_math_selftest": (i32, [vp, i32, vp]),
        "ct_debug_stats": (i32, [vp, vp]),
        "ct_debug_stats_ex": (i32, [vp, vp, C.c_uint32]),
        "ct_debug_invariants": (i32, [vp, vp]),
        "ct_debug_memory": (i32, [vp, vp]),
        "ct_debug_delta_grid": (i32, [vp, vp]),
        "ct_fetch_counters": (i32, [vp, C.POINTER(CtFetchCounters)]),
        "ct_debug_fetch_probe": (i32, [i32, u32, u32, C.POINTER(C.c_uint64)]),
        "ct_debug_fetch_probe_ws": (i32, [i32, u32, C.c_uint64, u32, C.POINTER(C.c_uint64)]),
        "ct_debug_track_lines": (i32, [vp, i32]),
        "ct_debug_touched_lines": (i32, [vp, C.POINTER(C.c_uint64), i32]),
        "ct_calculate_camera_variables": (i32, [vp, vp, vp, f32, f32, vp, vp, vp]),
        "ct_quantize_volume": (i32, [vp, vp, vp]),
        "ct_load_vdb": (i32, [C.c_char_p, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
        "ct_generate_mipmaps": (i32, [vp, vp, vp, C.c_size_t, C.POINTER(u32), C.POINTER(C.c_size_t), vp]),
        "ct_tile_owner": (u32, [u32, u32, u32]),
        "ct_group_create": (i32, [C.POINTER(CtScene), vp, u32, C.POINTER(vp)]),
        "ct_group_destroy": (i32, [vp]),
        "ct_group_last_error": (C.c_char_p, [vp]),
        "ct_group_size": (i32, [vp, C.POINTER(u32)]),
        "ct_group_handle": (i32, [vp, u32, C.POINTER(vp)]),
        "ct_group_set_camera": (i32, [vp, vp, vp, vp, vp]),
        "ct_group_render_accumulate": (i32, [vp, u32, u32]),
        "ct_group_reset": (i32, [vp]),
        "ct_group_merge": (i32, [vp]),
        "ct_group_download": (i32, [vp, i32, vp, C.c_size_t]),
        "ct_group_tonemap": (i32, [vp, f32, vp, C.POINTER(f32)]),
        "ct_group_is_converged": (i32, [vp, C.POINTER(i32), C.POINTER(C.c_uint64)]),
        "ct_group_counters": (i32, [vp, C.POINTER(CtCounters)]),
        "ct_make_procedural_cloud": (i32, [u32, u32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def check(rc: int, handle=None):
    if rc != CT_OK:
        msg = load().ct_last_error(handle)
        raise CloudTraceError(rc, msg.decode("utf-8", "replace") if msg else "")

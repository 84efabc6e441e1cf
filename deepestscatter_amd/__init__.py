"""deepestscatter_amd -- MI355X (gfx950) native replacement for the Monte-Carlo cloud radiance
estimator of marsermd/DeepestScatter (the PathTracingRenderer hot path).

The product is libcloudtrace.so (hand-written HIP kernels behind the C ABI of
include/cloudtrace.h); this package is the host-side plumbing around it.
"""
from ._lib import CloudTraceError  # noqa: F401
from .cloudtrace import (LIGHT_DIRECTIONS, POINT_TASK_DTYPE, CloudTracer, SceneParams, TracerGroup, algorithmic_bytes,  # noqa: F401
                         calculate_camera_variables, generate_mipmaps, load_mie_raw, load_vdb, make_point_tasks, make_procedural_cloud,
                         quantize_volume, shard_mask, tile_owner)

__all__ = [
    "CloudTracer", "CloudTraceError", "SceneParams", "TracerGroup", "LIGHT_DIRECTIONS", "algorithmic_bytes", "calculate_camera_variables",
    "generate_mipmaps", "load_mie_raw", "load_vdb", "make_point_tasks", "make_procedural_cloud", "POINT_TASK_DTYPE", "quantize_volume", "shard_mask", "tile_owner",
]

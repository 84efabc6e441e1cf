"""Thin object wrapper over the C ABI (include/cloudtrace.h): one `CloudTracer` = one CtHandle.

This is plumbing only -- every numeric result comes from libcloudtrace.so (HIP kernels).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import _lib
from ._lib import (CT_BUF_DENSITY, CT_BUF_FRAME, CT_BUF_INSCATTER, CT_BUF_M2, CT_BUF_MEAN, CT_BUF_SCREEN,
                   CtCounters, CtFetchCounters, CtScene, check)

MIE_FILE = Path(__file__).resolve().parent / "data" / "mie_raw.f32"

# Tasks.cpp:52-65
LIGHT_DIRECTIONS = {
    "Front": (-0.586, -0.766, -0.271),
    "Side": (-0.03, -0.25, 0.8),
    "Back": (0.586, -0.766, -0.271),
}


def load_mie_raw() -> tuple[np.ndarray, np.ndarray]:
    """The two 4096-entry Lorenz-Mie tables (data of Mie.cpp:8-8203): (mie, choppedMie)."""
    raw = np.fromfile(MIE_FILE, dtype="<f4")
    if raw.size != 8192:
        raise RuntimeError(f"{MIE_FILE} is corrupt")
    return np.ascontiguousarray(raw[:4096]), np.ascontiguousarray(raw[4096:])


# Gpu::PointRadianceTask, PointRadianceTask.h:70-77 (40 bytes)
POINT_TASK_DTYPE = np.dtype([("id", "<i4"), ("experimentCount", "<u4"), ("radiance", "<f4"),
                             ("runningVariance", "<f4"), ("position", "<f4", 3), ("direction", "<f4", 3)])
assert POINT_TASK_DTYPE.itemsize == 40


def make_point_tasks(positions, directions, ids=None) -> np.ndarray:
    """PointRadianceTask(id, position, direction) x N (PointRadianceTask.h:15-18)."""
    positions = np.asarray(positions, np.float32).reshape(-1, 3)
    directions = np.asarray(directions, np.float32).reshape(-1, 3)
    t = np.zeros(len(positions), POINT_TASK_DTYPE)
    t["id"] = np.arange(len(positions)) if ids is None else ids
    t["position"] = positions
    t["direction"] = directions
    return t


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def calculate_camera_variables(eye, lookat, up, hfov_deg: float, aspect_ratio: float):
    """sutil::calculateCameraVariables(..., fov_is_vertical=false), sutil.cpp:501-524."""
    e, la, u = (np.asarray(v, np.float32) for v in (eye, lookat, up))
    U, V, W = (np.empty(3, np.float32) for _ in range(3))
    check(_lib.load().ct_calculate_camera_variables(_p(e), _p(la), _p(u), hfov_deg, aspect_ratio, _p(U), _p(V), _p(W)))
    return U, V, W


def quantize_volume(grid: np.ndarray) -> np.ndarray:
    """Resources::loadVolumeBuffer's quantiser (Resources.cpp:92-141): float [Z,Y,X] payload ->
    uint8 [Z+2,Y+2,X+2] texture with a zero border."""
    grid = np.ascontiguousarray(grid, np.float32)
    nz, ny, nx = grid.shape
    pd = np.array([nx, ny, nz], np.uint32)
    out = np.empty((nz + 2, ny + 2, nx + 2), np.uint8)
    check(_lib.load().ct_quantize_volume(_p(grid), _p(pd), _p(out)))
    return out


def load_vdb(path) -> np.ndarray:
    """Resources::loadVolumeBuffer for a .vdb file (Resources.cpp:82-143; ct_load_vdb): -> uint8 [Z,Y,X] texture."""
    L = _lib.load()
    dims = np.zeros(3, np.uint32)
    n = C.c_size_t(0)
    err = C.create_string_buffer(512)
    rc = L.ct_load_vdb(str(path).encode(), _p(dims), None, 0, C.byref(n), err, 512)
    if rc != _lib.CT_OK:
        raise _lib.CloudTraceError(rc, err.value.decode("utf-8", "replace"))
    out = np.empty((int(dims[2]), int(dims[1]), int(dims[0])), np.uint8)
    rc = L.ct_load_vdb(str(path).encode(), _p(dims), _p(out), out.nbytes, C.byref(n), err, 512)
    if rc != _lib.CT_OK:
        raise _lib.CloudTraceError(rc, err.value.decode("utf-8", "replace"))
    return out


def generate_mipmaps(level0: np.ndarray) -> list[np.ndarray]:
    """Resources::generateMipmaps (Resources.cpp:169-209): list of uint8 [z,y,x] levels."""
    level0 = np.ascontiguousarray(level0, np.uint8)
    nz, ny, nx = level0.shape
    dims = np.array([nx, ny, nz], np.uint32)
    L = _lib.load()
    levels, total = C.c_uint32(0), C.c_size_t(0)
    offs = (C.c_size_t * 32)()
    check(L.ct_generate_mipmaps(_p(level0), _p(dims), None, 0, C.byref(levels), C.byref(total), offs))
    buf = np.empty(total.value, np.uint8)
    check(L.ct_generate_mipmaps(_p(level0), _p(dims), _p(buf), buf.size, C.byref(levels), C.byref(total), offs))
    out = []
    for l in range(levels.value):
        shape = (max(nz >> l, 1), max(ny >> l, 1), max(nx >> l, 1))
        n = int(np.prod(shape))
        out.append(buf[offs[l]:offs[l] + n].reshape(shape).copy())
    return out


def make_procedural_cloud(n: int, seed: int = 0xC10D5EED) -> np.ndarray:
    """Synthetic benchmark cloud of SURVEY.md section 8(d): uint8 [n,n,n] texture."""
    out = np.empty((n, n, n), np.uint8)
    check(_lib.load().ct_make_procedural_cloud(n, seed & 0xFFFFFFFF, _p(out)))
    return out


def tile_owner(tx: int, ty: int, shard_count: int) -> int:
    return int(_lib.load().ct_tile_owner(tx, ty, shard_count))


def shard_mask(width: int, height: int, shard_index: int, shard_count: int) -> np.ndarray:
    """bool [H, W]: pixels whose 8x8 tile belongs to `shard_index`."""
    ty, tx = np.meshgrid(np.arange(height) // 8, np.arange(width) // 8, indexing="ij")
    if shard_count <= 1:
        return np.ones((height, width), bool)
    return ((tx + 3 * ty) % shard_count) == shard_index


@dataclass
class SceneParams:
    """Defaults = the reference's hard-coded configuration (SURVEY.md section 5 'config')."""
    width: int = 512                       # Tasks.cpp:49
    height: int = 256                      # Tasks.cpp:50
    mode: int = 0                          # SunAndSkyAllScatter, Tasks.cpp:92
    estimator: int = 0
    cloud_size_m: float = 7000.0           # main.cpp:63
    mean_free_path_m: float = 10.0         # SceneDescription.h:80
    sample_step: float = 1.0 / 512.0       # installers.cpp:86
    max_depth: int = 2000                  # cloudRadianceMaterials.cu:4
    light_direction: tuple = LIGHT_DIRECTIONS["Side"]
    light_color: tuple = (1.0, 1.0, 1.0)   # installers.cpp:99
    light_intensity: float = 1e6           # installers.cpp:100
    device: int = 0
    shard_index: int = 0
    shard_count: int = 1
    flags: int = 0


def make_scene(density: np.ndarray, p: SceneParams):
    """-> (CtScene, objects that must stay alive until ct_create / ct_group_create has copied the host data)."""
    density = np.ascontiguousarray(density, np.uint8)
    if density.ndim != 3:
        raise ValueError("density must be uint8 [Z, Y, X]")
    mie, chopped = load_mie_raw()
    s = CtScene()
    s.abi_version = _lib.CT_ABI_VERSION
    nz, ny, nx = density.shape
    s.dims[:] = (nx, ny, nz)
    s.density_host = density.ctypes.data
    s.cloud_size_m = p.cloud_size_m
    s.mean_free_path_m = p.mean_free_path_m
    s.sample_step = p.sample_step
    s.mode = p.mode
    s.estimator = p.estimator
    s.max_depth = p.max_depth
    s.light_direction[:] = p.light_direction
    s.light_color[:] = p.light_color
    s.light_intensity = p.light_intensity
    s.width, s.height = p.width, p.height
    s.mie_host = mie.ctypes.data
    s.chopped_mie_host = chopped.ctypes.data
    s.mie_count = 4096
    s.device = p.device
    s.shard_index, s.shard_count = p.shard_index, p.shard_count
    s.flags = p.flags
    return s, (density, mie, chopped)


class CloudTracer:
    """Owns one CtHandle.  `density` is the uint8 [Z,Y,X] texture incl. its zero border."""

    def __init__(self, density: np.ndarray, params: SceneParams | None = None, **kw):
        self.L = _lib.load()
        self.params = params or SceneParams(**kw)
        s, keep = make_scene(density, self.params)
        self.dims = tuple(int(v) for v in s.dims)
        self.width, self.height = self.params.width, self.params.height
        h = C.c_void_p()
        rc = self.L.ct_create(C.byref(s), C.byref(h))
        del keep
        check(rc, None)
        self.h = h

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.L.ct_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ARenderer verbs ------------------------------------------------------------------------
    def set_camera(self, eye, U, V, W):
        a = [np.asarray(v, np.float32) for v in (eye, U, V, W)]
        check(self.L.ct_set_camera(self.h, *[_p(v) for v in a]), self.h)

    def render_subframe(self, subframe_id: int, out_dev_ptr: int | None = None):
        check(self.L.ct_render_subframe(self.h, subframe_id, C.c_void_p(out_dev_ptr) if out_dev_ptr else None), self.h)

    def accumulate(self, subframe_id: int, frame_dev_ptr: int | None = None):
        check(self.L.ct_accumulate(self.h, subframe_id, C.c_void_p(frame_dev_ptr) if frame_dev_ptr else None), self.h)

    def render_accumulate(self, first_subframe_id: int, count: int):
        check(self.L.ct_render_accumulate(self.h, first_subframe_id, count), self.h)

    def render_accumulate_async(self, first_subframe_id: int, count: int):
        """Enqueue a batch and return (ct_render_accumulate_async): its paths may finish in later launches, its accumulate
        kernel follows them; anything that waits (synchronize, mean, tonemap ...) brings the image up to date."""
        check(self.L.ct_render_accumulate_async(self.h, first_subframe_id, count), self.h)

    def synchronize(self):
        check(self.L.ct_synchronize(self.h), self.h)

    def point_radiance_launch(self, tasks: np.ndarray, first_frame_id: int, launches: int) -> np.ndarray:
        """`launches` launches of estimateEmission (pointEmissionCamera.cu:20-40) over `tasks`
        (POINT_TASK_DTYPE), updated in place and returned."""
        assert tasks.dtype == POINT_TASK_DTYPE and tasks.flags.c_contiguous
        check(self.L.ct_point_radiance_launch(self.h, _p(tasks), len(tasks), first_frame_id, launches), self.h)
        return tasks

    def generate_scatter_samples(self, count: int, batch_seed: int = 0):
        """generatePoints + firstScatterPosition: -> (positions [count,3], view directions [count,3])."""
        pos = np.empty((count, 3), np.float32)
        d = np.empty((count, 3), np.float32)
        check(self.L.ct_generate_scatter_samples(self.h, count, batch_seed & 0xFFFFFFFF, _p(pos), _p(d)), self.h)
        return pos, d

    def collect_descriptors(self, positions: np.ndarray, directions: np.ndarray) -> np.ndarray:
        """setupHierarchicalDescriptor (DisneyDescriptor.cuh:71-112) for (position, view direction) samples:
        -> uint8 [count, 10, 9, 5, 5] (layer, z, y, x), the `grid` bytes of Persistance::DisneyDescriptor."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        if len(pos) != len(d):
            raise ValueError("positions and directions differ in length")
        out = np.empty((len(pos), 10, 9, 5, 5), np.uint8)
        check(self.L.ct_collect_descriptors(self.h, _p(pos), _p(d), len(pos), _p(out)), self.h)
        return out

    def reset(self):
        check(self.L.ct_reset(self.h), self.h)

    def tonemap(self, exposure: float = 0.4):
        """-> (uint8 [H,W,4] screen, average luminance)."""
        screen = np.empty((self.height, self.width, 4), np.uint8)
        avg = C.c_float(0)
        check(self.L.ct_tonemap(self.h, exposure, _p(screen), C.byref(avg)), self.h)
        return screen, float(avg.value)

    def tonemap_async(self, exposure: float = 0.4):
        """ct_tonemap_async: the display update enqueued behind the batches in flight; the screen stays on the device
        (download(CT_BUF_SCREEN) after synchronize())."""
        check(self.L.ct_tonemap_async(self.h, exposure), self.h)

    def set_render_ahead(self, subframes: int):
        """ct_set_render_ahead: enqueued calls of fewer subframes than this are served by launches of this many, each call
        accumulating its own share (the reference's 10-subframe display cadence at the speed of long launches)."""
        check(self.L.ct_set_render_ahead(self.h, subframes), self.h)

    def rendered_subframes(self) -> int:
        n = C.c_uint32(0)
        check(self.L.ct_rendered_subframes(self.h, C.byref(n)), self.h)
        return int(n.value)

    def set_stop_when_converged(self, cadence: int = 10, min_subframes: int = 100):
        """ct_set_stop_when_converged: Camera::isConverged tested on the device behind every `cadence`-th subframe; the running
        mean freezes at the first count that passes (the reference's stopping point, Camera.cpp:179,232-268)."""
        check(self.L.ct_set_stop_when_converged(self.h, cadence, min_subframes), self.h)

    def converged_at(self):
        """-> (subframes the image was frozen at or 0, count of the last finished test, its unconverged pixels); never waits."""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
        check(self.L.ct_converged_at(self.h, C.byref(a), C.byref(b), C.byref(c)), self.h)
        return int(a.value), int(b.value), int(c.value)

    def is_converged(self):
        ok, bad = C.c_int32(0), C.c_uint64(0)
        check(self.L.ct_is_converged(self.h, C.byref(ok), C.byref(bad)), self.h)
        return bool(ok.value), int(bad.value)

    def tonemap_buffer(self, mean_dev_ptr: int, exposure: float = 0.4):
        """ct_tonemap_buffer: the tonemap of a caller-owned W*H float4 device buffer (a merged multi-GPU frame)."""
        screen = np.empty((self.height, self.width, 4), np.uint8)
        avg = C.c_float(0)
        check(self.L.ct_tonemap_buffer(self.h, C.c_void_p(mean_dev_ptr), exposure, _p(screen), C.byref(avg)), self.h)
        return screen, float(avg.value)

    def is_converged_buffers(self, mean_dev_ptr: int, m2_dev_ptr: int, subframes: int):
        ok, bad = C.c_int32(0), C.c_uint64(0)
        check(self.L.ct_is_converged_buffers(self.h, C.c_void_p(mean_dev_ptr), C.c_void_p(m2_dev_ptr), subframes,
                                             C.byref(ok), C.byref(bad)), self.h)
        return bool(ok.value), int(bad.value)

    # -- data ---------------------------------------------------------------------------------------
    def download(self, which: int) -> np.ndarray:
        n = C.c_size_t(0)
        check(self.L.ct_buffer_bytes(self.h, which, C.byref(n)), self.h)
        if which in (CT_BUF_MEAN, CT_BUF_M2, CT_BUF_FRAME):
            out = np.empty((self.height, self.width, 4), np.float32)
        elif which == CT_BUF_SCREEN:
            out = np.empty((self.height, self.width, 4), np.uint8)
        else:
            nx, ny, nz = self.dims
            out = np.empty((nz, ny, nx), np.uint8)
        assert out.nbytes == n.value
        check(self.L.ct_download(self.h, which, _p(out), out.nbytes), self.h)
        return out

    def upload(self, which: int, data: np.ndarray):
        """ct_upload: set the running mean or M2 (checkpoint / resume, with set_subframes)."""
        a = np.ascontiguousarray(data, np.float32)
        check(self.L.ct_upload(self.h, which, _p(a), a.nbytes), self.h)

    @staticmethod
    def _state_path(path):
        """np.savez appends '.npz' to a name without it; np.load does not: both sides use the same file name."""
        from pathlib import Path
        p = Path(path)
        return p if p.suffix == ".npz" else p.with_name(p.name + ".npz")

    def save_state(self, path):
        """(mean, M2, subframe count) -> .npz: everything a progressive render needs to continue exactly.  The count is that of
        the samples IN the buffers: with ct_set_stop_when_converged the running mean freezes at the reference's stopping count
        while the host goes on submitting (and ct_subframes goes on counting) -- then the frozen count is what is saved."""
        n = C.c_uint32(0)
        check(self.L.ct_subframes(self.h, C.byref(n)), self.h)
        mean, m2 = self.mean(), self.m2()          # (these wait: the counts read here and below are what the buffers hold)
        frozen_at = self.converged_at()[0]
        count = frozen_at if frozen_at else n.value
        np.savez(self._state_path(path), mean=mean, m2=m2, subframes=np.uint32(count))

    def load_state(self, path) -> int:
        """The reverse: the handle continues with subframe `count + 1`.  A handle whose image had frozen is released (ct_upload
        and ct_set_subframes clear the flag: the image they describe is a new one)."""
        with np.load(self._state_path(path)) as z:
            self.upload(CT_BUF_MEAN, z["mean"])
            self.upload(CT_BUF_M2, z["m2"])
            n = int(z["subframes"])
        self.set_subframes(n)
        return n

    def track_lines(self, enable: bool = True):
        """ct_debug_track_lines: record which 128-B lines of the density and shadow arrays the launches read (CT_STATS=1)."""
        check(self.L.ct_debug_track_lines(self.h, 1 if enable else 0), self.h)

    def touched_lines(self, clear: bool = False) -> dict:
        out = (C.c_uint64 * 4)()
        check(self.L.ct_debug_touched_lines(self.h, out, 1 if clear else 0), self.h)
        return {"density_lines_touched": int(out[0]), "shadow_lines_touched": int(out[1]), "density_lines": int(out[2]), "shadow_lines": int(out[3]),
                "touched_MiB": (int(out[0]) + int(out[1])) * 128 / 2 ** 20}

    def mean(self):
        return self.download(CT_BUF_MEAN)

    def m2(self):
        return self.download(CT_BUF_M2)

    def frame(self):
        return self.download(CT_BUF_FRAME)

    def inscatter(self):
        return self.download(CT_BUF_INSCATTER)

    def copy_to_device(self, which: int, dst_dev_ptr: int, nbytes: int):
        """D2D copy of a handle buffer into caller-owned device memory (e.g. a torch tensor)."""
        check(self.L.ct_copy_to_device(self.h, which, C.c_void_p(dst_dev_ptr), nbytes), self.h)

    def copy_to_device_async(self, which: int, dst_dev_ptr: int, nbytes: int):
        """The same, enqueued on the handle's stream behind the accumulate kernels, not waited for."""
        check(self.L.ct_copy_to_device_async(self.h, which, C.c_void_p(dst_dev_ptr), nbytes), self.h)

    def device_ptr(self, which: int) -> int:
        p = C.c_void_p()
        check(self.L.ct_device_ptr(self.h, which, C.byref(p)), self.h)
        return int(p.value)

    @property
    def subframes(self) -> int:
        n = C.c_uint32(0)
        check(self.L.ct_subframes(self.h, C.byref(n)), self.h)
        return int(n.value)

    def set_subframes(self, n: int):
        check(self.L.ct_set_subframes(self.h, n), self.h)

    def counters(self) -> dict:
        c = CtCounters()
        check(self.L.ct_counters(self.h, C.byref(c)), self.h)
        return c.as_dict()

    def fetch_counters(self) -> dict:
        """What the kernels issued for the lookups `counters()` reports (ct_fetch_counters)."""
        c = CtFetchCounters()
        check(self.L.ct_fetch_counters(self.h, C.byref(c)), self.h)
        return c.as_dict()

    def debug_invariants(self) -> dict:
        """Path conservation / sample integrity tallies (ct_debug_invariants)."""
        out = np.zeros(8, np.uint64)
        check(self.L.ct_debug_invariants(self.h, _p(out)), self.h)
        names = ["armed", "checks", "violations", "samples_without_alpha_1", "dealt", "resumed", "written", "suspended"]
        return {n: int(v) for n, v in zip(names, out)}

    def debug_memory(self) -> dict:
        """Bytes of the volume representations on the device (ct_debug_memory)."""
        out = np.zeros(8, np.uint64)
        check(self.L.ct_debug_memory(self.h, _p(out)), self.h)
        names = ["raw_texture", "density_bricks", "inscatter_bricks", "march_bricks_dense", "march_bricks_stored", "sparse",
                 "row_table", "coarse_clearance"]
        return {n: int(v) for n, v in zip(names, out)}

    def delta_grid(self) -> dict:
        """The DELTA estimator's majorant grid and kernel variant (ct_debug_delta_grid)."""
        out = np.zeros(8, np.uint32)
        check(self.L.ct_debug_delta_grid(self.h, _p(out)), self.h)
        return {"cell": int(out[0]), "stored": tuple(int(v) for v in out[1:4]), "origin": tuple(int(v) for v in out[4:7]),
                "nee": int(out[7] & 0xff), "interior": bool(out[7] & 0x100)}

    def kernel_time(self):
        """-> (estimator kernel ms, accumulate kernel ms, estimator launches) since create/reset."""
        a, b, n = C.c_double(0), C.c_double(0), C.c_uint64(0)
        check(self.L.ct_kernel_time(self.h, C.byref(a), C.byref(b), C.byref(n)), self.h)
        return float(a.value), float(b.value), int(n.value)

    def set_stream(self, hip_stream: int | None):
        check(self.L.ct_set_stream(self.h, C.c_void_p(hip_stream) if hip_stream else None), self.h)

    def debug_stats(self) -> dict:
        """Scheduler diagnostics (only filled when the process runs with CT_STATS=1)."""
        out = np.zeros(64, np.uint64)
        check(self.L.ct_debug_stats(self.h, _p(out)), self.h)
        names = ["regen_phases", "regen_lanes", "march_phases", "march_lanes", "scatter_phases", "scatter_lanes",
                 "fetched_steps", "fetched_zero_cells", "skipped_steps", "zero_cells_nonfree_brick",
                 "zero_cells_free_brick_d1", "skip_loop_wave_iterations", "nee_footprints_reused", "waves",
                 "stolen_jobs", "max_scheduler_visits_of_a_wave"]
        d = {n: int(v) for n, v in zip(names, out)}
        # waves by the time they ended (5 ms bins from their own start), and by how long they kept
        # running after they had found the job queue empty (0.5 ms bins)
        d["wave_end_hist_5ms"] = [int(v) for v in out[16:40]]
        d["wave_end_minus_drained_hist_0p5ms"] = [int(v) for v in out[40:64]]
        ex = np.zeros(72, np.uint64)
        check(self.L.ct_debug_stats_ex(self.h, _p(ex), 72), self.h)
        # of the march fetches: the lane's previous fetch was in the same 128-B brick line / a lower lane of the wave
        # fetches the same line in the same instruction (what a brick cache in LDS could find: DESIGN.md 4.3)
        d["march_fetch_same_line_as_lanes_previous"] = int(ex[68])
        d["march_fetch_line_shared_with_a_lower_lane"] = int(ex[69])
        d["raw"] = [int(v) for v in out]
        d["watchdog"] = int(out[63])     # exchange kernels: waves that gave up on a bounded wait (must be 0)
        return d

    def debug_suspended(self) -> int:
        """Paths handed from one async launch to the next so far."""
        n = C.c_uint64(0)
        check(self.L.ct_debug_suspended(self.h, C.byref(n)), self.h)
        return int(n.value)

    def debug_math_selftest(self, which: int) -> dict:
        """ct_debug_math_selftest: 0 = reciprocal, 1 = square root -> {tested, mismatches, first_bad_bits}."""
        out = np.zeros(3, np.uint64)
        check(self.L.ct_debug_math_selftest(self.h, which, _p(out)), self.h)
        return {"tested": int(out[0]), "mismatches": int(out[1]), "first_bad_bits": int(out[2])}

    def debug_cdf_inversion(self, first_u24: int, count: int) -> np.ndarray:
        out = np.empty(count, np.uint32)
        check(self.L.ct_debug_cdf_inversion(self.h, first_u24, count, _p(out)), self.h)
        return out


class TracerGroup:
    """Owns one CtGroup: a multi-GPU job driven by ONE process below the C ABI (ct_group_*; RCCL frame reduce).
    `devices` may repeat a device to rehearse an N-GPU job on fewer GPUs (merged without a collective then)."""

    def __init__(self, density: np.ndarray, devices, params: SceneParams | None = None, **kw):
        self.L = _lib.load()
        self.params = params or SceneParams(**kw)
        s, keep = make_scene(density, self.params)
        self.width, self.height = self.params.width, self.params.height
        dev = np.asarray(list(devices), np.int32)
        g = C.c_void_p()
        rc = self.L.ct_group_create(C.byref(s), _p(dev), len(dev), C.byref(g))
        del keep
        if rc != _lib.CT_OK:
            msg = self.L.ct_group_last_error(None)
            raise _lib.CloudTraceError(rc, msg.decode("utf-8", "replace") if msg else "")
        self.g = g

    def _check(self, rc):
        if rc != _lib.CT_OK:
            msg = self.L.ct_group_last_error(self.g)
            raise _lib.CloudTraceError(rc, msg.decode("utf-8", "replace") if msg else "")

    def close(self):
        if getattr(self, "g", None):
            self.L.ct_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, eye, U, V, W):
        a = [np.asarray(v, np.float32) for v in (eye, U, V, W)]
        self._check(self.L.ct_group_set_camera(self.g, *[_p(v) for v in a]))

    def render_accumulate(self, first_subframe_id: int, count: int):
        self._check(self.L.ct_group_render_accumulate(self.g, first_subframe_id, count))

    def reset(self):
        self._check(self.L.ct_group_reset(self.g))

    def merge(self):
        self._check(self.L.ct_group_merge(self.g))

    def _download(self, which):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self.L.ct_group_download(self.g, which, _p(out), out.nbytes))
        return out

    def mean(self):
        return self._download(CT_BUF_MEAN)

    def m2(self):
        return self._download(CT_BUF_M2)

    def tonemap(self, exposure: float = 0.4):
        screen = np.empty((self.height, self.width, 4), np.uint8)
        avg = C.c_float(0)
        self._check(self.L.ct_group_tonemap(self.g, exposure, _p(screen), C.byref(avg)))
        return screen, float(avg.value)

    def is_converged(self):
        ok, bad = C.c_int32(0), C.c_uint64(0)
        self._check(self.L.ct_group_is_converged(self.g, C.byref(ok), C.byref(bad)))
        return bool(ok.value), int(bad.value)

    def counters(self) -> dict:
        c = CtCounters()
        self._check(self.L.ct_group_counters(self.g, C.byref(c)))
        return c.as_dict()


def algorithmic_bytes(counters: dict, pixels_times_spp: int) -> int:
    """8 B per trilinear lookup (density or shadow volume) + 64 B per
    pixel per subframe of accumulation."""
    return 8 * (counters["density_lookups"] + counters["inscatter_lookups"]) + 64 * pixels_times_spp

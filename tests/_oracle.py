"""ctypes binding of the CPU oracle (oracle/ct_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
nothing under deepestscatter_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
ORACLE_DIR = ROOT / "oracle"
MIE_FILE = ROOT / "deepestscatter_amd" / "data" / "mie_raw.f32"


class OrcScene(C.Structure):
    _fields_ = [
        ("dims", C.c_uint32 * 3),
        ("density", C.c_void_p),
        ("inscatter", C.c_void_p),
        ("cloud_size_m", C.c_float),
        ("mean_free_path_m", C.c_float),
        ("sample_step", C.c_float),
        ("mode", C.c_int32),
        ("max_depth", C.c_uint32),
        ("light_direction", C.c_float * 3),
        ("light_color", C.c_float * 3),
        ("light_intensity", C.c_float),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("eye", C.c_float * 3),
        ("U", C.c_float * 3),
        ("V", C.c_float * 3),
        ("W", C.c_float * 3),
        ("mie_tex", C.c_void_p),
        ("chopped_mie_tex", C.c_void_p),
        ("chopped_cdf_tex", C.c_void_p),
        ("mie_count", C.c_uint32),
        ("estimator", C.c_int32),
        ("majorant", C.c_void_p),
        ("maj_bias", C.c_int32),
        ("maj_gx", C.c_int32),
        ("maj_gy", C.c_int32),
        ("maj_gz", C.c_int32),
        ("maj_cell", C.c_int32),
        ("maj_codes", C.c_void_p),
        ("inscatter_valid", C.c_void_p),
        ("maj_origin", C.c_int32 * 3),
        ("maj_virtual", C.c_int32 * 3),
    ]


class OrcCounters(C.Structure):
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


def _cpu_has(flag: str) -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    """Compile the oracle in place with gcc (no GPU, no reference needed)."""
    so = ORACLE_DIR / "libct_oracle.so"
    src = ORACLE_DIR / "ct_oracle.c"
    hdr = ROOT / "include" / "ct_fmath.h"
    if force or not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["make", "-C", str(ORACLE_DIR), "-s", "-B"], check=True)


_LIB = None


def lib(fast=False):
    """Load the oracle. fast=True picks the -mavx2 -mfma build when this CPU supports it; fast="libm" the build on libm's math."""
    global _LIB
    key = "fma" if (fast and _cpu_has("fma") and _cpu_has("avx2")) else "base"
    if fast == "libm":
        key = "libm"      # the restatement on the C library's math (statistical agreement only; ORC_LIBM in ct_oracle.c)
    if fast == "fixed8":
        key = "fixed8"    # ... with 1.8 fixed-point filter weights, like the reference's texture unit (ORC_TEX_FIXED8)
    if _LIB is None:
        _LIB = {}
    if key in _LIB:
        return _LIB[key]
    name = {"fma": "libct_oracle_fma.so", "libm": "libct_oracle_libm.so", "fixed8": "libct_oracle_fixed8.so"}.get(key, "libct_oracle.so")
    path = ORACLE_DIR / name
    build(force=not path.exists())   # also rebuilds a library older than its source before it is loaded
    if os.environ.get("CT_ORACLE_SANITIZE") == "1" and key in ("base", "fma"):
        # tests/test_sanitizers.py: this process has libasan preloaded and runs the oracle's own tests on the
        # AddressSanitizer + UndefinedBehaviorSanitizer build of the same source (make -C oracle asan)
        subprocess.run(["make", "-C", str(ORACLE_DIR), "-s", "asan"], check=True)
        path = ORACLE_DIR / "libct_oracle_asan.so"
    L = C.CDLL(str(path))
    f32p = C.POINTER(C.c_float)
    L.orc_tea4.restype = C.c_uint32
    L.orc_tea4.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_lcg.restype = C.c_uint32
    L.orc_lcg.argtypes = [C.POINTER(C.c_uint32)]
    L.orc_rnd.restype = C.c_float
    L.orc_rnd.argtypes = [C.POINTER(C.c_uint32)]
    L.orc_mie_phase_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_mie_integral_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_tex1d.restype = C.c_float
    L.orc_tex1d.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
    L.orc_tex3d.restype = C.c_float
    L.orc_tex3d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_derived_uniforms.argtypes = [C.POINTER(OrcScene), C.c_void_p]
    L.orc_render_subframe.argtypes = [C.POINTER(OrcScene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_void_p, C.POINTER(OrcCounters), C.c_int32]
    L.orc_point_radiance.argtypes = [C.POINTER(OrcScene), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(OrcCounters)]
    L.orc_inscatter.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_int32]
    L.orc_inscatter_texels.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32]
    L.orc_point_radiance_launch.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.POINTER(OrcCounters), C.c_int32]
    L.orc_majorant_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.orc_build_majorants.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.orc_generate_scatter_samples.argtypes = [C.POINTER(OrcScene), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.orc_collect_descriptors.argtypes = [C.POINTER(OrcScene), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_collect_descriptors.restype = None
    L.orc_point_task_merge.restype = C.c_int32
    L.orc_point_task_merge.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_camera_variables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_quantize_volume.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_mip_levels.restype = C.c_uint32
    L.orc_mip_levels.argtypes = [C.c_void_p]
    L.orc_generate_mipmaps.restype = C.c_size_t
    L.orc_generate_mipmaps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t]
    L.orc_reinhard.restype = C.c_float
    L.orc_reinhard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
    L.orc_is_converged.restype = C.c_int32
    L.orc_is_converged.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.POINTER(C.c_uint64)]
    for fn in ("orc_expf", "orc_logf"):
        getattr(L, fn).restype = C.c_float
        getattr(L, fn).argtypes = [C.c_float]
    L.orc_powf.restype = C.c_float
    L.orc_powf.argtypes = [C.c_float, C.c_float]
    L.orc_sincosf.argtypes = [C.c_float, f32p, f32p]
    L.orc_cdf_bisect_k.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32]
    L.orc_max_threads.restype = C.c_int32
    _LIB[key] = L
    return L


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def load_mie_raw() -> tuple[np.ndarray, np.ndarray]:
    raw = np.fromfile(MIE_FILE, dtype="<f4")
    assert raw.size == 8192
    return np.ascontiguousarray(raw[:4096]), np.ascontiguousarray(raw[4096:])


def mie_textures(fast: bool = False) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(mie phase, chopped phase, chopped CDF) textures per Mie.cpp:8206-8282."""
    L = lib(fast)
    mie, chopped = load_mie_raw()
    out = [np.empty(4096, np.float32) for _ in range(3)]
    L.orc_mie_phase_texture(_ptr(mie), 4096, _ptr(out[0]))
    L.orc_mie_phase_texture(_ptr(chopped), 4096, _ptr(out[1]))
    L.orc_mie_integral_texture(_ptr(chopped), 4096, _ptr(out[2]))
    return out[0], out[1], out[2]


def camera_variables(eye=(2.5, -0.4, 0.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                     hfov=30.0, aspect=1.0):
    L = lib()
    e, la, u = (np.asarray(v, np.float32) for v in (eye, lookat, up))
    U, V, W = (np.empty(3, np.float32) for _ in range(3))
    L.orc_camera_variables(_ptr(e), _ptr(la), _ptr(u), hfov, aspect, _ptr(U), _ptr(V), _ptr(W))
    return U, V, W


@dataclass
class Oracle:
    """A scene bound to the oracle; mirrors the parameters of CtScene."""
    density: np.ndarray                      # uint8 [Z, Y, X]
    width: int
    height: int
    mode: int = 0
    cloud_size_m: float = 7000.0
    mean_free_path_m: float = 10.0
    sample_step: float = 1.0 / 512.0
    max_depth: int = 2000
    light_direction: tuple = (-0.03, -0.25, 0.8)
    light_color: tuple = (1.0, 1.0, 1.0)
    light_intensity: float = 1e6
    eye: tuple = (2.5, -0.4, 0.0)
    estimator: int = 0
    threads: int = 0
    fast: bool = False
    inscatter: np.ndarray | str | None = None   # None: computed here; "lazy": texel by texel as paths touch it; "none": never
    counters: OrcCounters = field(default_factory=OrcCounters)

    def __post_init__(self):
        self.L = lib(self.fast)
        self.density = np.ascontiguousarray(self.density, dtype=np.uint8)
        assert self.density.ndim == 3
        self._tex = mie_textures(self.fast)
        U, V, W = camera_variables(self.eye, aspect=self.width / self.height)
        self.U, self.V, self.W = U, V, W
        s = OrcScene()
        nz, ny, nx = self.density.shape
        s.dims[:] = (nx, ny, nz)
        s.density = self.density.ctypes.data
        s.cloud_size_m = self.cloud_size_m
        s.mean_free_path_m = self.mean_free_path_m
        s.sample_step = self.sample_step
        s.mode = self.mode
        s.max_depth = self.max_depth
        s.light_direction[:] = self.light_direction
        s.light_color[:] = self.light_color
        s.light_intensity = self.light_intensity
        s.width, s.height = self.width, self.height
        s.eye[:] = self.eye
        s.U[:] = U.tolist()
        s.V[:] = V.tolist()
        s.W[:] = W.tolist()
        s.mie_tex = self._tex[0].ctypes.data
        s.chopped_mie_tex = self._tex[1].ctypes.data
        s.chopped_cdf_tex = self._tex[2].ctypes.data
        s.mie_count = 4096
        s.estimator = self.estimator
        if self.estimator == 1:
            dims = np.array([nx, ny, nz], np.uint32)
            grid = np.zeros(11, np.int32)
            self.L.orc_majorant_grid(_ptr(self.density), _ptr(dims), self.sample_step, _ptr(grid))
            bias, cell = int(grid[0]), int(grid[1])
            origin = np.ascontiguousarray(grid[2:5])
            gx, gy, gz = (int(v) for v in grid[5:8])
            self.majorant = np.empty((gz, gy, gx), np.uint8)
            self.majorant_codes = np.empty((gz, gy, gx), np.uint8)
            self.L.orc_build_majorants(_ptr(self.density), _ptr(dims), bias, cell, _ptr(origin), gx, gy, gz, _ptr(self.majorant),
                                       _ptr(self.majorant_codes))
            s.majorant = self.majorant.ctypes.data
            s.maj_codes = self.majorant_codes.ctypes.data
            s.maj_bias, s.maj_gx, s.maj_gy, s.maj_gz, s.maj_cell = bias, gx, gy, gz, cell
            s.maj_origin[:] = [int(v) for v in grid[2:5]]
            s.maj_virtual[:] = [int(v) for v in grid[8:11]]
        self.scene = s
        self.inscatter_valid = None
        if self.inscatter is None:
            self.inscatter = np.empty_like(self.density)
            self.L.orc_inscatter(C.byref(s), _ptr(self.inscatter), self.threads)
        elif isinstance(self.inscatter, str) and self.inscatter == "lazy":
            # the oracle's own shadow values for the texels its paths touch (OrcScene::inscatter_valid)
            self.inscatter = np.zeros_like(self.density)
            self.inscatter_valid = np.zeros_like(self.density)
            s.inscatter_valid = self.inscatter_valid.ctypes.data
        elif isinstance(self.inscatter, str) and self.inscatter == "none":
            self.inscatter = np.zeros_like(self.density)     # for callers that only use inscatter_texels()
        else:
            self.inscatter = np.ascontiguousarray(self.inscatter, dtype=np.uint8)
        s.inscatter = self.inscatter.ctypes.data

    def inscatter_texels(self, xyz: np.ndarray) -> np.ndarray:
        """inScatter (inScatter.cu:40-66) at the texels xyz[i] = (x, y, z): uint8 [len(xyz)]."""
        xyz = np.ascontiguousarray(xyz, np.uint32).reshape(-1, 3)
        out = np.empty(len(xyz), np.uint8)
        self.L.orc_inscatter_texels(C.byref(self.scene), _ptr(xyz), len(xyz), _ptr(out), self.threads)
        return out

    def set_camera(self, eye, U, V, W):
        self.scene.eye[:] = [float(v) for v in eye]
        self.scene.U[:] = [float(v) for v in U]
        self.scene.V[:] = [float(v) for v in V]
        self.scene.W[:] = [float(v) for v in W]

    def derived_uniforms(self) -> np.ndarray:
        out = np.empty(16, np.float32)
        self.L.orc_derived_uniforms(C.byref(self.scene), _ptr(out))
        return out

    def render_subframe(self, subframe_id: int, window=None) -> np.ndarray:
        """float32 [H, W, 4]; pixels outside `window`=(x0,y0,x1,y1) stay 0."""
        frame = np.zeros((self.height, self.width, 4), np.float32)
        x0, y0, x1, y1 = window or (0, 0, self.width, self.height)
        self.L.orc_render_subframe(C.byref(self.scene), subframe_id, x0, y0, x1, y1, _ptr(frame),
                                   C.byref(self.counters), self.threads)
        return frame

    def render(self, spp: int, first: int = 1, window=None):
        """Progressive render: returns (mean, m2) float32 [H, W, 4] after `spp` subframes."""
        mean = np.zeros((self.height, self.width, 4), np.float32)
        m2 = np.zeros_like(mean)
        for sid in range(first, first + spp):
            frame = self.render_subframe(sid, window)
            if window is not None:
                x0, y0, x1, y1 = window
                mask = np.zeros((self.height, self.width), bool)
                mask[y0:y1, x0:x1] = True
                frame[~mask] = 0
            self.L.orc_accumulate(_ptr(frame), _ptr(mean), _ptr(m2), sid, self.width * self.height)
        return mean, m2

    def generate_scatter_samples(self, count: int, batch_seed: int = 0):
        pos = np.empty((count, 3), np.float32)
        d = np.empty((count, 3), np.float32)
        self.L.orc_generate_scatter_samples(C.byref(self.scene), count, batch_seed, _ptr(pos), _ptr(d))
        return pos, d

    def collect_descriptors(self, positions: np.ndarray, directions: np.ndarray) -> np.ndarray:
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        out = np.empty((len(pos), 10, 9, 5, 5), np.uint8)
        self.L.orc_collect_descriptors(C.byref(self.scene), _ptr(pos), _ptr(d), len(pos), _ptr(out))
        return out

    def point_radiance_launch(self, tasks: np.ndarray, first_frame: int, launches: int) -> np.ndarray:
        """Oracle twin of ct_point_radiance_launch; `tasks` has the 40-byte PointRadianceTask layout."""
        assert tasks.dtype.itemsize == 40 and tasks.flags.c_contiguous
        self.L.orc_point_radiance_launch(C.byref(self.scene), _ptr(tasks), len(tasks), first_frame, launches,
                                         C.byref(self.counters), self.threads)
        return tasks

    def point_radiance(self, launch_id: int, subframe_id: int, origin, direction) -> np.ndarray:
        o = np.asarray(origin, np.float32)
        d = np.asarray(direction, np.float32)
        out = np.empty(3, np.float32)
        self.L.orc_point_radiance(C.byref(self.scene), launch_id, subframe_id, _ptr(o), _ptr(d), _ptr(out),
                                  C.byref(self.counters))
        return out


def accumulate(frame, mean, m2, subframe_id):
    lib().orc_accumulate(_ptr(frame), _ptr(mean), _ptr(m2), subframe_id, frame.size // 4)


def reinhard(mean: np.ndarray, exposure: float = 0.4):
    h, w, _ = mean.shape
    mean = np.ascontiguousarray(mean, np.float32)
    screen = np.empty((h, w, 4), np.uint8)
    avg = lib().orc_reinhard(_ptr(mean), w, h, exposure, _ptr(screen))
    return screen, float(avg)


def is_converged(mean, m2, subframe_id):
    n = C.c_uint64(0)
    mean = np.ascontiguousarray(mean, np.float32)
    m2 = np.ascontiguousarray(m2, np.float32)
    ok = lib().orc_is_converged(_ptr(mean), _ptr(m2), subframe_id, mean.size // 4, C.byref(n))
    return bool(ok), int(n.value)


def quantize_volume(grid: np.ndarray) -> np.ndarray:
    grid = np.ascontiguousarray(grid, np.float32)
    nz, ny, nx = grid.shape
    pd = np.array([nx, ny, nz], np.uint32)
    out = np.empty((nz + 2, ny + 2, nx + 2), np.uint8)
    lib().orc_quantize_volume(_ptr(grid), _ptr(pd), _ptr(out))
    return out


def generate_mipmaps(level0: np.ndarray) -> list[np.ndarray]:
    level0 = np.ascontiguousarray(level0, np.uint8)
    nz, ny, nx = level0.shape
    dims = np.array([nx, ny, nz], np.uint32)
    L = lib()
    levels = L.orc_mip_levels(_ptr(dims))
    shapes = [(max(nz >> l, 1), max(ny >> l, 1), max(nx >> l, 1)) for l in range(levels)]
    total = sum(int(np.prod(s)) for s in shapes)
    buf = np.empty(total, np.uint8)
    n = L.orc_generate_mipmaps(_ptr(level0), _ptr(dims), _ptr(buf))
    assert n == total
    out, off = [], 0
    for s in shapes:
        sz = int(np.prod(s))
        out.append(buf[off:off + sz].reshape(s).copy())
        off += sz
    return out


def tex3d(texels: np.ndarray, pos_box) -> float:
    texels = np.ascontiguousarray(texels, np.uint8)
    nz, ny, nx = texels.shape
    dims = np.array([nx, ny, nz], np.uint32)
    p = np.asarray(pos_box, np.float32)
    return float(lib().orc_tex3d(_ptr(texels), _ptr(dims), _ptr(p)))


def cdf_bisect_k(first_u24: int, count: int, fast: bool = True) -> np.ndarray:
    """k of the literal 16-step bisection (cloud.cuh:162-180) for a range of 24-bit randoms."""
    L = lib(fast)
    cdf = mie_textures(fast)[2]
    out = np.empty(count, np.uint32)
    L.orc_cdf_bisect_k(_ptr(cdf), 4096, first_u24, count, _ptr(out), 0)
    return out

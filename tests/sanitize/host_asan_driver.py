"""Runs INSIDE a child process that has libasan preloaded (tests/test_sanitizers.py starts it): loads the ASan + UBSan build of
the device-free part of the C ABI (tests/sanitize/_build/libcloudtrace_host_asan.so = csrc/ct_host.cpp + host/VdbReader.h)
with ctypes and drives it.  TEST INFRASTRUCTURE.

    python host_asan_driver.py selftest            camera, quantiser, mipmaps, procedural cloud, every .vdb variant of
                                                   tests/test_vdb.py against the reference loader's arithmetic
    python host_asan_driver.py fuzz <examples> [seed]   hypothesis-driven mutations (truncation, bit flips, length-field edits,
                                                   spliced noise) of .vdb files in the four compression modes: every call
                                                   must return CT_OK or a CT_E_* code with a message

A sanitizer report aborts the process (non-zero exit, report on stderr), which is what the parent asserts on."""
import ctypes as C
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import _vdb  # noqa: E402

L = C.CDLL(str(HERE / "_build" / "libcloudtrace_host_asan.so"))
vp = C.c_void_p
L.ct_load_vdb.argtypes = [C.c_char_p, vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
L.ct_quantize_volume.argtypes = [vp, vp, vp]
L.ct_generate_mipmaps.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t), vp]
L.ct_make_procedural_cloud.argtypes = [C.c_uint32, C.c_uint32, vp]
L.ct_calculate_camera_variables.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]


def _p(a):
    return a.ctypes.data_as(vp)


def load_vdb(path):
    """-> (rc, message, texture or None); the same two-call protocol as deepestscatter_amd.load_vdb"""
    dims = np.zeros(3, np.uint32)
    n = C.c_size_t(0)
    err = C.create_string_buffer(512)
    rc = L.ct_load_vdb(str(path).encode(), _p(dims), None, 0, C.byref(n), err, 512)
    if rc != 0:
        return rc, err.value.decode("utf-8", "replace"), None
    if n.value != int(dims[0]) * int(dims[1]) * int(dims[2]) or n.value > (1 << 31):
        return -1, "inconsistent size", None
    out = np.empty((int(dims[2]), int(dims[1]), int(dims[0])), np.uint8)
    rc = L.ct_load_vdb(str(path).encode(), _p(dims), _p(out), out.nbytes, C.byref(n), err, 512)
    return rc, err.value.decode("utf-8", "replace"), out if rc == 0 else None


def cloud(shape, seed):
    rng = np.random.default_rng(seed)
    nx, ny, nz = shape
    x, y, z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    v = np.zeros(shape, np.float32)
    for _ in range(3):
        c = rng.uniform(0.2, 0.8, 3) * shape
        r = rng.uniform(0.2, 0.4) * min(shape)
        v += np.clip(1.0 - np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / r, 0, 1).astype(np.float32)
    return v, v > 0


MODES = [
    dict(compression=_vdb.COMPRESS_NONE),
    dict(compression=_vdb.COMPRESS_ACTIVE_MASK),
    dict(compression=_vdb.COMPRESS_ZIP | _vdb.COMPRESS_ACTIVE_MASK),
    dict(compression=_vdb.COMPRESS_BLOSC | _vdb.COMPRESS_ACTIVE_MASK),
]
EXTRA = [
    dict(compression=_vdb.COMPRESS_ZIP),
    dict(compression=_vdb.COMPRESS_BLOSC),
    dict(compression=_vdb.COMPRESS_BLOSC, blosc_memcpy=True),
    dict(compression=_vdb.COMPRESS_ZIP | _vdb.COMPRESS_ACTIVE_MASK, grid_offsets=False),
    dict(compression=_vdb.COMPRESS_ZIP | _vdb.COMPRESS_ACTIVE_MASK, version=222),
    dict(compression=_vdb.COMPRESS_BLOSC | _vdb.COMPRESS_ACTIVE_MASK, version=223, transform="UniformScaleTranslateMap"),
    dict(compression=_vdb.COMPRESS_ZIP, version=221, transform="AffineMap"),
    dict(compression=_vdb.COMPRESS_ACTIVE_MASK, force_all_values=True, extra_grids=2),
    dict(compression=_vdb.COMPRESS_ZIP | _vdb.COMPRESS_ACTIVE_MASK, half=True),
]


def selftest():
    # camera frame (sutil.cpp:501-524): |W| = |lookat - eye|, |U| = |W| tan(hfov / 2)
    eye, at, up = (np.array(v, np.float32) for v in ((2.5, -0.4, 0.0), (0, 0, 0), (0, 1, 0)))
    U, V, W = (np.zeros(3, np.float32) for _ in range(3))
    assert L.ct_calculate_camera_variables(_p(eye), _p(at), _p(up), 30.0, 2.0, _p(U), _p(V), _p(W)) == 0
    assert abs(np.linalg.norm(W) - np.linalg.norm(eye)) < 1e-6 and abs(np.linalg.norm(U) - np.linalg.norm(W) * np.tan(np.radians(15))) < 1e-5
    # quantiser + mip pyramid (Resources.cpp:92-141, 169-209) on a ragged grid
    pd = np.array([5, 3, 7], np.uint32)
    grid = np.random.default_rng(1).random(5 * 3 * 7).astype(np.float32)
    tex = np.zeros((9, 5, 7), np.uint8)
    assert L.ct_quantize_volume(_p(grid), _p(pd), _p(tex)) == 0
    want = (grid.astype(np.float64) / float(grid.max()) * 255).astype(np.uint8).reshape(7, 3, 5)
    assert np.array_equal(tex[1:-1, 1:-1, 1:-1], want) and tex[0].max() == 0 and tex[:, 0].max() == 0 and tex[:, :, -1].max() == 0
    dims = np.array([7, 5, 9], np.uint32)
    levels, total = C.c_uint32(0), C.c_size_t(0)
    offs = (C.c_size_t * 32)()
    assert L.ct_generate_mipmaps(_p(tex), _p(dims), None, 0, C.byref(levels), C.byref(total), offs) == 0
    buf = np.zeros(total.value, np.uint8)
    assert L.ct_generate_mipmaps(_p(tex), _p(dims), _p(buf), buf.size, C.byref(levels), C.byref(total), offs) == 0
    assert levels.value == 4 and np.array_equal(buf[:tex.size], tex.reshape(-1))
    # (a capacity one byte short must be refused, not overrun)
    assert L.ct_generate_mipmaps(_p(tex), _p(dims), _p(buf), buf.size - 1, C.byref(levels), C.byref(total), offs) != 0
    cloud_tex = np.zeros((24, 24, 24), np.uint8)
    assert L.ct_make_procedural_cloud(24, 0, _p(cloud_tex)) == 0 and cloud_tex.max() == 255 and cloud_tex[0].max() == 0
    # every .vdb variant: loader == the reference loader's arithmetic on the dense arrays
    v, a = cloud((37, 22, 41), 7)
    origin = (-21, 100, 4070)
    tiles = [(1, (-16, 136, 4096), 2.5, True), (1, (40, 96, 4072), 0.5, False)]
    with tempfile.TemporaryDirectory() as d:
        for i, kw in enumerate(MODES + EXTRA):
            path = Path(d) / f"c{i}.vdb"
            kw = dict(kw)
            half = kw.pop("half", False)
            _vdb.write_vdb(path, v, a, origin=origin, tiles=tiles, half=half, **kw)
            rc, msg, tex = load_vdb(path)
            assert rc == 0, (kw, msg)
            if not half:
                want = _vdb.reference_texture(v, a, origin, tiles)
                assert tex.shape == want.shape and np.array_equal(tex, want), kw
        rc, msg, _ = load_vdb(Path(d) / "missing.vdb")
        assert rc != 0 and msg
    print("selftest ok")


def fuzz(examples, seed=None):
    import hypothesis
    from hypothesis import HealthCheck, given, settings, strategies as st
    v, a = cloud((20, 12, 17), 3)
    base = []
    tmp = tempfile.mkdtemp(prefix="ct_fuzz_")
    for i, kw in enumerate(MODES):
        path = Path(tmp) / f"base{i}.vdb"
        _vdb.write_vdb(path, v, a, origin=(-5, 3, 4090), tiles=[(1, (-16, 8, 4096), 1.5, True)], **kw)
        base.append(path.read_bytes())
        assert load_vdb(path)[0] == 0
    target = Path(tmp) / "m.vdb"
    stats = {"ok": 0, "rejected": 0}
    extreme = [0, 1, 0x7F, 0x80, 0xFF, 0xFFFF, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0x7FFFFFFFFFFFFFFF]

    @settings(max_examples=examples, deadline=None, database=None, derandomize=seed is None,
              suppress_health_check=list(HealthCheck))
    @given(st.data())
    def run(data):
        b = bytearray(base[data.draw(st.integers(0, len(base) - 1), label="mode")])
        for _ in range(data.draw(st.integers(1, 4), label="mutations")):
            kind = data.draw(st.sampled_from(("truncate", "bitflip", "length", "noise", "dup")), label="kind")
            at = data.draw(st.integers(0, max(len(b) - 1, 0)), label="at")
            if kind == "truncate":
                del b[at:]
            elif kind == "bitflip" and b:
                b[at] ^= 1 << data.draw(st.integers(0, 7))
            elif kind == "length" and b:
                # a 4- or 8-byte little-endian field overwritten with an extreme or a small value (counts, sizes, offsets)
                width = data.draw(st.sampled_from((4, 8)))
                val = data.draw(st.one_of(st.sampled_from(extreme), st.integers(0, 4096))) & ((1 << (8 * width)) - 1)
                b[at:at + width] = val.to_bytes(width, "little")
            elif kind == "noise":
                n = data.draw(st.integers(1, 64))
                b[at:at + n] = data.draw(st.binary(min_size=n, max_size=n))
            elif kind == "dup" and b:
                n = data.draw(st.integers(1, 256))
                b[at:at] = b[at:at + n]
            if not b:
                break
        target.write_bytes(bytes(b))
        rc, msg, tex = load_vdb(target)
        if rc == 0:
            stats["ok"] += 1
        else:
            assert rc in (-1, -3) and msg, (rc, msg)     # CT_E_INVAL / CT_E_NOMEM with a message
            stats["rejected"] += 1

    if seed is not None:
        run = hypothesis.seed(seed)(run)
    run()
    print(f"fuzz: {stats['ok']} still loaded, {stats['rejected']} rejected with a message")


if __name__ == "__main__":
    if sys.argv[1] == "selftest":
        selftest()
    else:
        fuzz(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else None)

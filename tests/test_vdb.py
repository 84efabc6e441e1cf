"""The .vdb loader (Resources::loadVolumeBuffer, Resources.cpp:82-143) without OpenVDB:
deepestscatter_amd/host/VdbReader.h through ct_load_vdb, against files produced by the independent writer tests/_vdb.py
and against the reference loader's arithmetic applied to the dense arrays directly.  Plainly: no file read here was
written by OpenVDB or Houdini (none exists on the image, none ships with the reference)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import _vdb
import deepestscatter_amd as ds

ROOT = Path(__file__).resolve().parents[1]


def _cloud(shape, seed, leak=True):
    """Dense test block [x, y, z]: a blobby fog volume, active where non-zero, plus (leak) some inactive voxels
    that still hold a value -- getValue returns those too (Resources.cpp:136)."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = shape
    x, y, z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    v = np.zeros(shape, np.float32)
    for _ in range(4):
        c = rng.uniform(0.2, 0.8, 3) * shape
        r = rng.uniform(0.15, 0.35) * min(shape)
        v += np.clip(1.0 - np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / r, 0, 1).astype(np.float32)
    v *= rng.uniform(0.5, 1.0, shape).astype(np.float32) * 3.7
    active = v > 0
    if leak:
        ghost = (rng.random(shape) < 0.01) & ~active
        v[ghost] = 0.25
    return v, active


VARIANTS = [
    dict(compression=_vdb.COMPRESS_NONE),
    dict(compression=_vdb.COMPRESS_ACTIVE_MASK),
    dict(compression=_vdb.COMPRESS_ZIP | _vdb.COMPRESS_ACTIVE_MASK),
    dict(compression=_vdb.COMPRESS_ZIP),
    dict(compression=_vdb.COMPRESS_BLOSC | _vdb.COMPRESS_ACTIVE_MASK),
    dict(compression=_vdb.COMPRESS_BLOSC),
    dict(compression=_vdb.COMPRESS_BLOSC, blosc_memcpy=True),
    dict(compression=_vdb.COMPRESS_ZIP | _vdb.COMPRESS_ACTIVE_MASK, grid_offsets=False),
    dict(compression=_vdb.COMPRESS_ZIP | _vdb.COMPRESS_ACTIVE_MASK, version=222),
    dict(compression=_vdb.COMPRESS_BLOSC | _vdb.COMPRESS_ACTIVE_MASK, version=223, transform="UniformScaleTranslateMap"),
    dict(compression=_vdb.COMPRESS_ZIP, version=221, transform="AffineMap"),
    dict(compression=_vdb.COMPRESS_ACTIVE_MASK, force_all_values=True, extra_grids=2),
]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_vdb_loader_matches_the_reference_semantics(tmp_path, kw):
    v, a = _cloud((45, 30, 52), seed=7)
    origin = (-21, 100, 4070)          # negative coordinates; spans two 4096^3 root children and several 128^3 nodes
    tiles = [(1, (-16, 136, 4096), 2.5, True),      # an active 8^3 tile beside the leaves
             (1, (40, 96, 4072), 0.5, False)]       # an inactive tile with a value: outside the active box or inside, it is read
    path = tmp_path / "cloud.vdb"
    _vdb.write_vdb(path, v, a, origin=origin, tiles=tiles, **kw)
    want = _vdb.reference_texture(v, a, origin, tiles)
    got = ds.load_vdb(path)
    assert got.shape == want.shape and got.dtype == np.uint8
    assert np.array_equal(got, want)
    assert got.max() == 255


def test_blosc_frames_with_lz4_streams_from_the_real_library(tmp_path, monkeypatch):
    """The Blosc path again with the LZ4 streams produced by the system's liblz4 (the real encoder: long matches,
    overlapping copies, its own parsing choices) instead of the test writer's: the product's LZ4 decoder must read what
    the real library writes.  (liblz4.so.1 is on the image as a runtime library; whole frames from the real Blosc library: the next test.)"""
    real = _vdb.system_lz4()
    if real is None:
        pytest.skip("liblz4.so.1 not found")
    monkeypatch.setattr(_vdb, "LZ4_ENCODER", real[0])
    v, a = _cloud((60, 41, 37), seed=13)
    v[5:25, 5:25, 5:25] = 1.5                      # long runs: the real encoder emits long and overlapping matches
    a = v > 0
    for comp in (_vdb.COMPRESS_BLOSC, _vdb.COMPRESS_BLOSC | _vdb.COMPRESS_ACTIVE_MASK):
        path = tmp_path / f"real{comp}.vdb"
        _vdb.write_vdb(path, v, a, origin=(3, -70, 120), compression=comp)
        assert np.array_equal(ds.load_vdb(path), _vdb.reference_texture(v, a, (3, -70, 120)))


def test_blosc_frames_from_the_real_blosc_library(tmp_path, monkeypatch):
    """Every Blosc frame of the file made by the real Blosc library (1.21 in a conda environment of the image), called
    with OpenVDB's own arguments (bloscToStream: clevel 9, shuffle, typesize 4, LZ4, one block): its header flags, its
    split streams, its choice between compressed and stored blocks.  Smooth, noisy and constant leaf contents, float and
    half grids, with and without active-mask compression."""
    real = _vdb.system_blosc()
    if real is None:
        pytest.skip("no Blosc library on this machine")
    monkeypatch.setattr(_vdb, "BLOSC_ENCODER", real)
    rng = np.random.default_rng(23)
    v, a = _cloud((70, 45, 52), seed=31)
    v[8:40, 8:40, 8:40] = 0.75                                       # whole leaves of one value
    v[40:60, 10:30, 10:40] = rng.random((20, 20, 30), dtype=np.float32) + 0.01   # incompressible mantissas
    a = v > 0
    for half in (False, True):
        for comp in (_vdb.COMPRESS_BLOSC, _vdb.COMPRESS_BLOSC | _vdb.COMPRESS_ACTIVE_MASK):
            path = tmp_path / f"blosc{int(half)}{comp}.vdb"
            vv = v.astype(np.float16).astype(np.float32) if half else v
            _vdb.write_vdb(path, vv, a, origin=(-9, 200, 17), compression=comp, half=half)
            assert np.array_equal(ds.load_vdb(path), _vdb.reference_texture(vv, a, (-9, 200, 17))), (half, comp, real.version)


def test_vdb_half_float_grid_and_a_large_tile(tmp_path):
    v, a = _cloud((20, 24, 18), seed=3, leak=False)
    v = v.astype(np.float16).astype(np.float32)
    a = v > 0
    tiles = [(2, (128, 0, 0), float(np.float16(1.75)), True)]   # an active 128^3 tile: the box grows to hold all of it
    path = tmp_path / "half.vdb"
    _vdb.write_vdb(path, v, a, origin=(100, 3, 5), tiles=tiles, half=True, compression=_vdb.COMPRESS_BLOSC | _vdb.COMPRESS_ACTIVE_MASK)
    got = ds.load_vdb(path)
    want = _vdb.reference_texture(v, a, (100, 3, 5), tiles)
    assert got.shape == want.shape and got.shape[:2] == (130, 130)
    assert np.array_equal(got, want)


def test_vdb_non_zero_background_and_two_inactive_values(tmp_path):
    """Mask compression with a background that is not zero and two distinct inactive values per node (metadata 4 and 5
    of io::writeCompressedValues)."""
    rng = np.random.default_rng(11)
    v = np.full((16, 16, 16), 0.125, np.float32)
    a = rng.random(v.shape) < 0.4
    v[a] = rng.uniform(0.5, 2.0, int(a.sum())).astype(np.float32)
    v[(~a) & (rng.random(v.shape) < 0.3)] = 0.375
    for comp in (_vdb.COMPRESS_ACTIVE_MASK, _vdb.COMPRESS_ZIP | _vdb.COMPRESS_ACTIVE_MASK):
        path = tmp_path / f"bg{comp}.vdb"
        _vdb.write_vdb(path, v, a, origin=(8, 8, 8), background=0.125, compression=comp)
        assert np.array_equal(ds.load_vdb(path), _vdb.reference_texture(v, a, (8, 8, 8), background=0.125))


def test_vdb_errors_are_reported(tmp_path):
    v, a = _cloud((9, 9, 9), seed=1)
    bad = tmp_path / "bad.vdb"
    bad.write_bytes(b"not a vdb file at all, but long enough to hold a header" * 4)
    with pytest.raises(ds.CloudTraceError, match="magic"):
        ds.load_vdb(bad)
    with pytest.raises(ds.CloudTraceError, match="cannot open"):
        ds.load_vdb(tmp_path / "missing.vdb")
    p = tmp_path / "vec.vdb"
    _vdb.write_vdb(p, v, a, grid_type="Tree_vec3s_5_4_3")
    with pytest.raises(ds.CloudTraceError, match="expected a FloatGrid"):
        ds.load_vdb(p)
    p = tmp_path / "old.vdb"
    _vdb.write_vdb(p, v, a, version=212)
    with pytest.raises(ds.CloudTraceError, match="version 212"):
        ds.load_vdb(p)
    p = tmp_path / "frustum.vdb"
    _vdb.write_vdb(p, v, a, transform="NonlinearFrustumMap")
    with pytest.raises(ds.CloudTraceError, match="NonlinearFrustumMap"):
        ds.load_vdb(p)
    p = tmp_path / "empty.vdb"
    _vdb.write_vdb(p, np.zeros((8, 8, 8), np.float32), np.zeros((8, 8, 8), bool))
    with pytest.raises(ds.CloudTraceError, match="no active voxel"):
        ds.load_vdb(p)
    p = tmp_path / "trunc.vdb"
    n = _vdb.write_vdb(p, v, a, compression=_vdb.COMPRESS_ZIP)
    p.write_bytes(p.read_bytes()[: n - 40])
    with pytest.raises(ds.CloudTraceError):
        ds.load_vdb(p)


def test_lz4_and_blosc_frames_of_the_test_writer_are_well_formed():
    """The writer's own LZ4 encoder against a plain-Python decoder written from the LZ4 block format description
    (so that a reader bug cannot hide behind a matching writer bug)."""
    def lz4_decode(src, n):
        out = bytearray()
        i = 0
        while i < len(src):
            tok = src[i]; i += 1
            ll = tok >> 4
            if ll == 15:
                while True:
                    b = src[i]; i += 1; ll += b
                    if b != 255:
                        break
            out += src[i:i + ll]; i += ll
            if i >= len(src):
                break
            off = src[i] | src[i + 1] << 8; i += 2
            ml = tok & 15
            if ml == 15:
                while True:
                    b = src[i]; i += 1; ml += b
                    if b != 255:
                        break
            for _ in range(ml + 4):
                out.append(out[-off])
        assert len(out) == n
        return bytes(out)
    rng = np.random.default_rng(5)
    for data in (bytes(4096), bytes(rng.integers(0, 4, 5000, dtype=np.uint8)), bytes(rng.integers(0, 256, 777, dtype=np.uint8)),
                 b"abcd" * 300 + b"xyz", b"short"):
        enc = _vdb.lz4_encode(data)
        assert lz4_decode(enc, len(data)) == data
    assert len(_vdb.lz4_encode(bytes(4096))) < 64
    real = _vdb.system_lz4()
    if real is not None:                            # and against the real library, both ways
        for data in (bytes(4096), bytes(rng.integers(0, 4, 5000, dtype=np.uint8)), b"abcd" * 300 + b"xyz"):
            assert real[1](_vdb.lz4_encode(data), len(data)) == data
            assert lz4_decode(real[0](data), len(data)) == data


@pytest.mark.gpu
def test_cpp_cli_renders_a_vdb_file(tmp_path):
    """The C++ host (deepestscatter_amd/host: the reference's Scene / VDBCloud / Resources / Camera mirror) reads a .vdb
    the way Tasks::renderCloud hands it one, and renders the same picture as the Python path on the loaded texture."""
    from deepestscatter_amd import build, exr
    cli = build.build_cli()
    v, a = _cloud((40, 28, 36), seed=9, leak=False)
    path = tmp_path / "cloud.vdb"
    _vdb.write_vdb(path, v, a, origin=(-5, 17, 300), compression=_vdb.COMPRESS_BLOSC | _vdb.COMPRESS_ACTIVE_MASK)
    tex = ds.load_vdb(path)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([str(cli), str(path), "--size", "64x48", "--spp", "20", "--light", "Side", "--out", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"Creating buffer of size {tex.shape[2]}x{tex.shape[1]}x{tex.shape[0]}" in r.stdout      # Resources.cpp:119
    tr = ds.CloudTracer(tex, width=64, height=48, light_direction=ds.LIGHT_DIRECTIONS["Side"])
    tr.render_accumulate(1, 20)
    img = exr.read_exr(out / "cloud.vdb.Side.PT.exr")               # <cloud>.<Light>.PT.exr, Tasks.cpp:88-90
    assert np.array_equal(img, tr.mean()[..., :3])
    assert img.max() > 0
    tr.close()

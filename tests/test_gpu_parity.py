"""GPU parity tests proper (run with -m gpu on an MI355X): every result comes from the HIP
kernels through the C ABI and is compared with the CPU oracle on the same seeded inputs.

Tolerance: the north star allows 1e-3 relative L2; because the kernels and the oracle share the
deterministic float32 contract of include/ct_fmath.h the comparison is BIT-EXACT here (radiance,
mean, M2, shadow volume, screen bytes, work counters).
"""
import numpy as np
import pytest

import _oracle as O
import deepestscatter_amd as ds
from deepestscatter_amd import _lib
from conftest import sphere_volume

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def make_pair(tex, w, h, mode=0, **kw):
    tr = ds.CloudTracer(tex, width=w, height=h, mode=mode, **kw)
    okw = {k: v for k, v in kw.items() if k in ("cloud_size_m", "mean_free_path_m", "sample_step", "max_depth",
                                                 "light_direction", "light_color", "light_intensity", "estimator")}
    orc = O.Oracle(tex, w, h, mode=mode, fast=True, **okw)
    return tr, orc


# ---------------------------------------------------------------------------------------------------
def test_loaded_library_is_the_in_tree_hip_extension():
    L = _lib.load()
    assert str(_lib.LIB_PATH).endswith("deepestscatter_amd/libcloudtrace.so")
    with open("/proc/self/maps") as f:
        assert any("deepestscatter_amd/libcloudtrace.so" in line for line in f)
    assert L.ct_tile_owner(1, 1, 4) == 0


@pytest.mark.parametrize("dims", [(32, 32, 32), (20, 28, 36), (40, 24, 16)])
def test_shadow_volume_bit_exact(dims):
    tex = sphere_volume(dims=dims, seed=2)
    tr, orc = make_pair(tex, 8, 8)
    got = tr.inscatter()
    assert got.shape == orc.inscatter.shape
    assert np.array_equal(got, orc.inscatter)
    assert np.array_equal(tr.download(_lib.CT_BUF_DENSITY), tex)
    tr.close()


def test_shadow_volume_other_light_and_step():
    tex = sphere_volume(24, seed=4)
    tr, orc = make_pair(tex, 8, 8, light_direction=(0.586, -0.766, -0.271), sample_step=1.0 / 128.0,
                        cloud_size_m=3000.0)
    assert np.array_equal(tr.inscatter(), orc.inscatter)
    tr.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_subframe_radiance_bit_exact_all_modes(mode):
    tex = sphere_volume(32, seed=1)
    w, h = 32, 24
    tr, orc = make_pair(tex, w, h, mode=mode)
    for sid in (1, 2, 7):
        tr.render_subframe(sid)
        got = tr.frame()
        ref = orc.render_subframe(sid)
        assert np.array_equal(got, ref), (mode, sid, rel_l2(got, ref))
    assert tr.counters() == orc.counters.as_dict()
    assert np.isfinite(tr.frame()).all()
    tr.close()


def test_progressive_mean_and_m2_bit_exact():
    tex = sphere_volume(32, seed=3)
    w, h = 24, 24
    tr, orc = make_pair(tex, w, h, mode=0)
    mean, m2 = orc.render(16)
    tr.render_accumulate(1, 5)      # two batches of different size
    tr.render_accumulate(6, 11)
    assert tr.subframes == 16
    assert np.array_equal(tr.mean(), mean)
    assert np.array_equal(tr.m2(), m2)
    assert rel_l2(tr.mean(), mean) <= 1e-3       # the north-star bar, trivially
    assert tr.counters() == orc.counters.as_dict()
    # separate render + accumulate calls (the reference's two launches) give the same thing
    tr.reset()
    for sid in range(1, 5):
        tr.render_subframe(sid)
        tr.accumulate(sid)
    mean4, m24 = O.Oracle(tex, w, h, mode=0, fast=True, inscatter=orc.inscatter).render(4)
    assert np.array_equal(tr.mean(), mean4) and np.array_equal(tr.m2(), m24)
    tr.close()


def test_ragged_frame_and_noncubic_volume():
    tex = sphere_volume(dims=(20, 28, 36), seed=5)
    w, h = 37, 21                      # not multiples of the 8x8 tile
    tr, orc = make_pair(tex, w, h, mode=0)
    mean, m2 = orc.render(3)
    tr.render_accumulate(1, 3)
    assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2)
    tr.close()


@pytest.mark.parametrize("dims,radius", [((64, 64, 64), 0.16), ((96, 72, 80), 0.12)])
def test_free_space_skipping_is_bit_exact(dims, radius):
    """Small cloud in a big box: most bricks are free, so the march replays position updates
    without fetching.  Radiance AND the algorithmic lookup counters must still equal the oracle's
    (which fetches at every step) and the plain one-thread-per-pixel kernel's."""
    tex = sphere_volume(dims=dims, radius=radius, seed=21)
    w, h = 48, 40
    tr, orc = make_pair(tex, w, h, mode=0)
    mean, m2 = orc.render(3)
    tr.render_accumulate(1, 3)
    assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2)
    assert tr.counters() == orc.counters.as_dict()
    plain = ds.CloudTracer(tex, width=w, height=h, mode=0, flags=_lib.CT_FLAG_SIMPLE_KERNEL)
    plain.render_accumulate(1, 3)
    assert np.array_equal(plain.mean(), mean) and plain.counters() == tr.counters()
    # other view directions exercise the other axes of the skip bound
    for eye in ((0.2, 2.4, 0.3), (-0.5, -0.3, -2.3)):
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
        tr.set_camera(eye, U, V, W)
        orc.set_camera(eye, U, V, W)
        tr.render_subframe(5)
        assert np.array_equal(tr.frame(), orc.render_subframe(5))
    tr.close()
    plain.close()


def _speck_volume(dims, seed):
    """Isolated non-zero texels, thin shells, a slab that touches two faces of the volume (no zero
    border there) -- everything the clearance codes of the march bricks can get wrong."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = dims
    t = np.zeros((nz, ny, nx), np.uint8)
    specks = rng.random(t.shape) < 0.004
    t[specks] = rng.integers(1, 256, int(specks.sum()), dtype=np.uint8)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    r = np.sqrt((x - 0.5 * nx) ** 2 + (y - 0.45 * ny) ** 2 + (z - 0.55 * nz) ** 2)
    t[np.abs(r - 0.30 * min(dims)) < 0.6] = 200          # one-texel shell
    t[np.abs(r - 0.12 * min(dims)) < 1.2] = 90           # thicker inner shell
    t[nz // 3, :, :] = np.maximum(t[nz // 3, :, :], 40)   # slab through the whole box, faces included
    t[:, :, 0] = np.maximum(t[:, :, 0], 3)                # a non-zero face
    return t


@pytest.mark.parametrize("dims", [(49, 50, 52), (64, 37, 45)])
def test_march_bricks_on_adversarial_volumes(dims):
    """3x4x4 march bricks: x extents that are not multiples of 3, clearance codes next to isolated
    texels and one-texel shells, non-zero texels on the faces (isInBox ends those paths, not the
    zero border).  Radiance and the algorithm's lookup counts must equal the oracle's."""
    tex = _speck_volume(dims, seed=77)
    w, h = 40, 32
    tr, orc = make_pair(tex, w, h, mode=0, cloud_size_m=900.0)
    mean, m2 = orc.render(2)
    tr.render_accumulate(1, 2)
    assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2)
    assert tr.counters() == orc.counters.as_dict()
    for eye in ((0.3, 2.2, 0.4), (-0.4, -0.2, -2.4), (-2.3, 0.3, 0.2)):
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
        tr.set_camera(eye, U, V, W)
        orc.set_camera(eye, U, V, W)
        tr.render_subframe(7)
        assert np.array_equal(tr.frame(), orc.render_subframe(7))
    assert tr.counters() == orc.counters.as_dict()
    tr.close()


@pytest.mark.parametrize("dims", [(2, 2, 2), (3, 5, 4), (4, 3, 9), (7, 8, 5)])
def test_tiny_random_volumes_without_zero_border(dims):
    """Smallest legal volumes, every texel random (no zero border, so the march ends by isInBox alone) and
    fewer texels than a brick on some axis.  1x1x1 is refused like any dimension below 2."""
    rng = np.random.default_rng(sum(dims))
    tex = rng.integers(0, 256, dims[::-1]).astype(np.uint8)
    tr, orc = make_pair(tex, 24, 16, mode=0, cloud_size_m=50.0)
    tr.render_accumulate_async(1, 3)
    tr.render_accumulate_async(4, 2)
    mean, m2 = orc.render(5)
    assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2)
    assert tr.counters() == orc.counters.as_dict()
    assert np.array_equal(tr.inscatter(), orc.inscatter)
    tr.close()
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.CloudTracer(np.zeros((1, 1, 1), np.uint8), width=8, height=8)
    assert e.value.code == _lib.CT_E_INVAL


def test_simple_kernel_equals_persistent_kernel():
    tex = sphere_volume(32, seed=6)
    a = ds.CloudTracer(tex, width=40, height=32, mode=0)
    b = ds.CloudTracer(tex, width=40, height=32, mode=0, flags=_lib.CT_FLAG_SIMPLE_KERNEL)
    a.render_accumulate(1, 4)
    b.render_accumulate(1, 4)
    assert np.array_equal(a.mean(), b.mean()) and np.array_equal(a.m2(), b.m2())
    assert a.counters() == b.counters()
    a.close()
    b.close()


def test_depth_cap_and_thick_medium():
    tex = sphere_volume(24)
    tr, orc = make_pair(tex, 16, 16, mode=0, max_depth=6, cloud_size_m=20000.0)
    mean, _ = orc.render(2)
    tr.render_accumulate(1, 2)
    assert np.array_equal(tr.mean(), mean)
    c = tr.counters()
    assert c == orc.counters.as_dict() and c["depth_capped"] > 0
    tr.close()


def test_empty_volume_and_camera_inside_box():
    tex = np.zeros((16, 16, 16), np.uint8)
    tr, orc = make_pair(tex, 16, 16)
    tr.render_accumulate(1, 2)
    assert np.all(tr.mean()[..., :3] == 0) and np.all(tr.mean()[..., 3] == 1)
    assert np.all(tr.inscatter() == 255)
    tr.close()
    # camera inside the cloud's box: intersect reports minimalRayDistance (cloudBBox.cu:29-35)
    tex = sphere_volume(24, seed=8)
    tr, orc = make_pair(tex, 16, 16)
    eye = (0.1, 0.05, -0.2)
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 60.0, 1.0)
    tr.set_camera(eye, U, V, W)
    orc.set_camera(eye, U, V, W)
    tr.render_subframe(3)
    assert np.array_equal(tr.frame(), orc.render_subframe(3))
    tr.close()


def test_coloured_light_and_intensity_linearity():
    tex = sphere_volume(24, seed=9)
    tr, orc = make_pair(tex, 16, 16, light_color=(1.0, 0.5, 0.25), light_intensity=3e5)
    tr.render_subframe(1)
    f = tr.frame()
    assert np.array_equal(f, orc.render_subframe(1))
    assert not np.array_equal(f[..., 0], f[..., 1])
    # doubling the intensity doubles every sample exactly (power of two, no rounding)
    tr2 = ds.CloudTracer(tex, width=16, height=16, light_color=(1.0, 0.5, 0.25), light_intensity=6e5)
    tr2.render_subframe(1)
    assert np.array_equal(tr2.frame()[..., :3], f[..., :3] * 2)
    tr.close()
    tr2.close()


@pytest.mark.parametrize("count", [2, 3])
def test_pixel_tile_shards_sum_to_the_whole(count):
    tex = sphere_volume(32, seed=10)
    w, h = 40, 24
    whole = ds.CloudTracer(tex, width=w, height=h)
    whole.render_accumulate(1, 3)
    total_mean = np.zeros((h, w, 4), np.float32)
    total_m2 = np.zeros_like(total_mean)
    ctr = None
    for i in range(count):
        sh = ds.CloudTracer(tex, width=w, height=h, shard_index=i, shard_count=count)
        sh.render_accumulate(1, 3)
        m = sh.mean()
        mask = ds.shard_mask(w, h, i, count)
        assert np.all(m[~mask] == 0)                     # foreign pixels untouched: sum == merge
        total_mean += m
        total_m2 += sh.m2()
        c = sh.counters()
        ctr = c if ctr is None else {k: ctr[k] + c[k] for k in c}
        sh.close()
    assert np.array_equal(total_mean, whole.mean())
    assert np.array_equal(total_m2, whole.m2())
    assert ctr == whole.counters()
    whole.close()


def test_tonemap_and_convergence_bit_exact():
    tex = sphere_volume(32, seed=12)
    w, h = 32, 32
    tr, orc = make_pair(tex, w, h)
    mean, m2 = orc.render(8)
    tr.render_accumulate(1, 8)
    screen, avg = tr.tonemap(0.4)
    ref_screen, ref_avg = O.reinhard(mean, 0.4)
    assert avg == ref_avg
    assert np.array_equal(screen, ref_screen)
    assert np.array_equal(tr.download(_lib.CT_BUF_SCREEN), ref_screen)
    assert tr.is_converged() == (False, w * h)          # < 100 subframes (Camera.cpp:234)
    tr.set_subframes(100)                                 # same buffers, pretend N = 100
    assert tr.is_converged() == O.is_converged(mean, m2, 100)
    tr.close()


def test_cdf_inversion_exhaustive():
    """All 2^24 values of the random number give the same cos(theta) as the literal 16-step
    bisection of cloud.cuh:162-180."""
    tr = ds.CloudTracer(sphere_volume(16), width=8, height=8)
    ref = O.cdf_bisect_k(0, 1 << 24)
    got = tr.debug_cdf_inversion(0, 1 << 24)
    assert np.array_equal(got, ref)
    tr.close()


def test_error_behaviour_on_device():
    tex = sphere_volume(16)
    tr = ds.CloudTracer(tex, width=16, height=16)
    with pytest.raises(_lib.CloudTraceError) as e:
        tr.render_accumulate(5, 1)                      # out of order
    assert e.value.code == _lib.CT_E_STATE
    with pytest.raises(_lib.CloudTraceError) as e:
        tr.accumulate(0)
    assert e.value.code == _lib.CT_E_INVAL
    with pytest.raises(_lib.CloudTraceError):
        tr.download(99)
    tr.render_accumulate(1, 1)                           # still usable after errors
    tr.reset()
    assert tr.subframes == 0 and tr.counters()["paths"] == 0
    tr.close()


def test_cpp_host_cli_matches_python_path(tmp_path):
    """The C++ mirror of the reference's Scene/Camera/PathTracingRenderer classes (headless
    `cloudtrace`, = main.cpp + Tasks::renderCloud) drives the same C ABI: its EXR (and PFM) output equals the
    running mean obtained through the Python binding, for the fused and the two-launch loop."""
    import subprocess
    from deepestscatter_amd import build
    cli = build.build_cli()
    tex = ds.make_procedural_cloud(32)
    from deepestscatter_amd import exr
    tr = ds.CloudTracer(tex, width=40, height=24, light_direction=ds.LIGHT_DIRECTIONS["Back"])
    tr.render_accumulate(1, 5)
    want = tr.mean()[..., :3]
    tr.close()
    for extra in ([], ["--unfused", "--format", "pfm"]):
        out = tmp_path / ("u" if extra else "f")
        out.mkdir()
        r = subprocess.run([str(cli), "procedural:32", "--size", "40x24", "--spp", "5", "--light", "Back",
                            "--out", str(out), *extra], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "rendering subframe 5" in r.stdout and "MS/FRAME" in r.stdout
        if extra:
            raw = (out / "procedural_32.Back.PT.pfm").read_bytes()
            header_end = 0
            for _ in range(3):
                header_end = raw.index(b"\n", header_end) + 1
            assert raw[:header_end].split() == [b"PF", b"40", b"24", b"-1.0"]
            img = np.frombuffer(raw[header_end:], "<f4").reshape(24, 40, 3)
        else:
            # the reference's output: <cloud>.<Light>.PT.exr, R/G/B FLOAT, DECREASING_Y (Camera.cpp:149-175)
            path = out / "procedural_32.Back.PT.exr"
            img = exr.read_exr(path)
            raw = path.read_bytes()
            assert raw[:8] == b"\x76\x2f\x31\x01\x02\x00\x00\x00" and b"lineOrder\0lineOrder\0\x01\0\0\0\x01" in raw
        assert np.array_equal(img, want)
    # --estimator delta: the same host classes around Woodcock tracking
    tr = ds.CloudTracer(tex, width=40, height=24, light_direction=ds.LIGHT_DIRECTIONS["Back"], estimator=1)
    tr.render_accumulate(1, 5)
    want_delta = tr.mean()[..., :3]
    tr.close()
    out = tmp_path / "d"
    out.mkdir()
    r = subprocess.run([str(cli), "procedural:32", "--size", "40x24", "--spp", "5", "--light", "Back", "--estimator", "delta",
                        "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(exr.read_exr(out / "procedural_32.Back.PT.exr"), want_delta)
    assert not np.array_equal(want_delta, want)
    bad = subprocess.run([str(cli), "procedural:32", "--mode", "bogus"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Invalid Render Mode" in bad.stdout      # CloudMaterial.cpp:62 / main.cpp:65-76


@pytest.mark.gpu
def test_cpp_host_runs_until_converged_and_stops_where_the_reference_loop_stops(tmp_path):
    """`cloudtrace` without --spp renders until Camera::isConverged holds (Camera.cpp:179).  The headless Camera enqueues
    batches up to its save points and leaves the test to the device (ct_set_stop_when_converged); `--display` runs the
    reference's own loop -- 10 subframes, tonemap, test, every call waited for.  Same stopping count (100: the first test,
    not a save point, so the headless run learns of it at 120 and has rendered 20 subframes for nothing), same image."""
    import subprocess
    from deepestscatter_amd import build, exr
    cli = build.build_cli()
    tex = ds.make_procedural_cloud(32)
    tr = ds.CloudTracer(tex, width=40, height=24, light_direction=ds.LIGHT_DIRECTIONS["Back"])
    tr.render_accumulate(1, 100)
    assert tr.is_converged()[0]
    want = tr.mean()[..., :3]
    tr.close()
    for extra in ([], ["--display"]):
        out = tmp_path / ("d" if extra else "h")
        out.mkdir()
        r = subprocess.run([str(cli), "procedural:32", "--size", "40x24", "--light", "Back", "--out", str(out), *extra],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [l for l in r.stdout.splitlines() if l.startswith("rendering subframe")]
        assert lines[-1] == "rendering subframe 100", lines[-3:]
        assert np.array_equal(exr.read_exr(out / "procedural_32.Back.PT.exr"), want)


# ---- DELTA estimator (Woodcock tracking; oracle twin: delta_flight in oracle/ct_oracle.c) -------------
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_delta_estimator_bit_exact_vs_oracle(mode):
    tex = sphere_volume(32, seed=31)
    w, h = 32, 24
    tr, orc = make_pair(tex, w, h, mode=mode, estimator=1)
    mean, m2 = orc.render(6)
    tr.render_accumulate(1, 2)
    tr.render_accumulate(3, 4)
    assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2)
    assert tr.counters() == orc.counters.as_dict()
    tr.close()


@pytest.mark.parametrize("dims,radius", [((64, 64, 64), 0.16), ((96, 72, 80), 0.12)])
def test_delta_brick_replay_is_bit_exact(dims, radius):
    """Small cloud in a big box: the kernel crosses runs of empty bricks by replaying DDA steps
    without loading their majorants; the oracle walks them one by one."""
    tex = sphere_volume(dims=dims, radius=radius, seed=33)
    w, h = 40, 32
    tr, orc = make_pair(tex, w, h, mode=0, estimator=1)
    mean, _ = orc.render(3)
    tr.render_accumulate(1, 3)
    assert np.array_equal(tr.mean(), mean)
    assert tr.counters() == orc.counters.as_dict()
    for eye in ((0.2, 2.4, 0.3), (-0.5, -0.3, -2.3), (0.05, 0.0, 0.1)):     # last one: camera inside the box
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0.3), (0, 1, 0), 40.0, w / h)
        tr.set_camera(eye, U, V, W)
        orc.set_camera(eye, U, V, W)
        tr.render_subframe(5)
        assert np.array_equal(tr.frame(), orc.render_subframe(5))
    tr.close()


def test_delta_flight_at_the_edge_of_the_brick_grid():
    """A volume without a zero border: its clamp-to-edge values fill the apron bricks, so Woodcock flights
    keep drawing tentative collisions all the way to the outer face of the brick grid, where a position
    within an ulp of the face rounds into a brick that does not exist (found by tools/soak_continuation.py
    as a GPU memory fault: dims (14,5,34), 288x141).  The fetch clamps the texel index to the grid; the
    value is the oracle's clamp-to-edge one."""
    rng = np.random.default_rng(84)
    tex = rng.integers(0, 256, (34, 5, 14)).astype(np.uint8)
    w, h = 288, 141
    eye = (0.13926167059960207, -0.06436481666191757, 1.1023804706499314)
    for mode, depth in ((2, 50), (0, 300)):
        tr, orc = make_pair(tex, w, h, mode=mode, estimator=1, cloud_size_m=20000.0, sample_step=1 / 512, max_depth=depth,
                            light_direction=(-0.14960121569810259, -1.138979689131218, 0.7051263148774981))
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
        tr.set_camera(eye, U, V, W)
        orc.set_camera(eye, U, V, W)
        mean, m2 = orc.render(3)
        tr.render_accumulate(1, 2)
        tr.render_accumulate_async(3, 1)
        assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2)
        assert tr.counters() == orc.counters.as_dict()
        tr.close()


def test_delta_shards_and_point_tasks():
    tex = sphere_volume(32, seed=35)
    w, h = 40, 24
    whole = ds.CloudTracer(tex, width=w, height=h, estimator=1)
    whole.render_accumulate(1, 3)
    merged = np.zeros((h, w, 4), np.float32)
    for i in range(2):
        sh = ds.CloudTracer(tex, width=w, height=h, estimator=1, shard_index=i, shard_count=2)
        sh.render_accumulate(1, 3)
        merged += sh.mean()
        sh.close()
    assert np.array_equal(merged, whole.mean())
    whole.close()
    rng = np.random.default_rng(3)
    pos = (rng.random((100, 3), dtype=np.float32) - 0.5) * 0.5
    d = rng.normal(size=(100, 3)).astype(np.float32)
    tr = ds.CloudTracer(tex, width=8, height=8, mode=1, estimator=1)
    orc = O.Oracle(tex, 8, 8, mode=1, estimator=1, fast=True)
    got = tr.point_radiance_launch(ds.make_point_tasks(pos, d), 1, 5)
    ref = orc.point_radiance_launch(ds.make_point_tasks(pos, d), 1, 5)
    assert got.tobytes() == ref.tobytes()
    tr.close()


def test_delta_agrees_with_march_statistically():
    """DELTA is an unbiased estimator of the medium that the reference's march samples with an
    O(step) bias.  With the reference's 7000 m cloud the optical depth per step reaches 1.37 and the
    march reads 5-10 % low (tools/march_delta_convergence.py); at a tenth of
    that density the bias is below the noise and the two images must agree."""
    tex = ds.make_procedural_cloud(96)
    w = h = 48
    spp = 2048
    imgs = []
    for est in (0, 1):
        tr = ds.CloudTracer(tex, width=w, height=h, estimator=est, cloud_size_m=700.0)
        tr.render_accumulate(1, spp)
        imgs.append((tr.mean()[..., 0].astype(np.float64), tr.m2()[..., 0].astype(np.float64)))
        tr.close()
    (m0, v0), (m1, v1) = imgs
    se_mean = np.sqrt((v0 + v1).sum()) / (spp * m0.size)
    assert abs(m1.mean() - m0.mean()) < 4 * se_mean + 0.01 * m0.mean(), (m0.mean(), m1.mean(), se_mean)
    # 8x8-pixel block means against the combined 4-sigma confidence interval of the two estimates
    b0 = m0.reshape(6, 8, 6, 8).mean(axis=(1, 3))
    b1 = m1.reshape(6, 8, 6, 8).mean(axis=(1, 3))
    se = np.sqrt((v0 + v1).reshape(6, 8, 6, 8).sum(axis=(1, 3))) / (64 * spp)
    assert np.all(np.abs(b1 - b0) <= 4 * se + 0.02 * np.maximum(b0, 1e-3)), (np.abs(b1 - b0) / (se + 1e-12)).max()


def test_pipelined_batches_equal_synchronous_batches():
    """ct_render_accumulate_async: two batches in flight on two streams, accumulate in subframe order --
    mean, M2 and counters must equal the synchronous calls' and the oracle's, whatever is interleaved."""
    tex = sphere_volume(40, seed=13)
    w, h = 56, 40
    tr, orc = make_pair(tex, w, h, mode=0)
    ref = ds.CloudTracer(tex, width=w, height=h, mode=0)
    sizes = [3, 2, 4, 1, 5, 2]
    first = 1
    for n in sizes:
        tr.render_accumulate_async(first, n)
        ref.render_accumulate(first, n)
        first += n
    assert tr.subframes == sum(sizes)
    tr.synchronize()
    mean, m2 = orc.render(sum(sizes))
    assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2)
    assert np.array_equal(ref.mean(), mean) and np.array_equal(ref.m2(), m2)
    assert tr.counters() == ref.counters() == orc.counters.as_dict()
    # entry points that read results wait for the batches in flight by themselves
    tr.render_accumulate_async(first, 3)
    tr.render_accumulate_async(first + 3, 3)
    got = tr.mean()                                   # no explicit synchronize
    ref.render_accumulate(first, 6)
    assert np.array_equal(got, ref.mean())
    # a new pose while batches are in flight
    eye = (0.3, 2.2, 0.4)
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    tr.render_accumulate_async(first + 6, 2)
    tr.set_camera(eye, U, V, W)
    ref.render_accumulate(first + 6, 2)
    ref.set_camera(eye, U, V, W)
    tr.reset()
    ref.reset()
    tr.render_accumulate_async(1, 4)
    tr.render_accumulate_async(5, 4)
    ref.render_accumulate(1, 8)
    assert np.array_equal(tr.mean(), ref.mean()) and np.array_equal(tr.m2(), ref.m2())
    render_ms, accum_ms, launches = tr.kernel_time()
    assert launches == 2 and render_ms > 0 and accum_ms > 0
    tr.close()
    ref.close()


def test_path_continuation_across_launches_is_bit_exact():
    """Async batches of the MARCH estimator suspend their surviving paths when the job list is empty and the
    next launch resumes them.  A thick medium (every path bounces for long) and small batches make sure that
    thousands of paths cross launch boundaries; mean, M2 and the counters must equal the synchronous run's."""
    tex = sphere_volume(48, radius=0.42, seed=17)
    w, h = 256, 192
    kw = dict(mode=0, cloud_size_m=30000.0, max_depth=300)
    tr = ds.CloudTracer(tex, width=w, height=h, **kw)
    ref = ds.CloudTracer(tex, width=w, height=h, **kw)
    first = 1
    for n in (4, 3, 5, 2, 6, 4):
        tr.render_accumulate_async(first, n)
        ref.render_accumulate(first, n)
        first += n
    tr.synchronize()
    assert tr.debug_suspended() > 1000
    assert np.array_equal(tr.mean(), ref.mean()) and np.array_equal(tr.m2(), ref.m2())
    assert tr.counters() == ref.counters()
    assert tr.counters()["depth_capped"] > 0
    # and against the oracle on a window
    orc = O.Oracle(tex, w, h, fast=True, **kw)
    win = (96, 64, 160, 128)
    mean, _ = orc.render(8, window=win)
    tr2 = ds.CloudTracer(tex, width=w, height=h, **kw)
    tr2.render_accumulate_async(1, 5)
    tr2.render_accumulate_async(6, 3)
    got = tr2.mean()
    assert np.array_equal(got[win[1]:win[3], win[0]:win[2]], mean[win[1]:win[3], win[0]:win[2]])
    for t in (tr, ref, tr2):
        t.close()


def test_shards_with_enqueued_batches_sum_to_the_whole():
    """Pixel-tile shards (one handle per GPU in production) rendered with enqueued batches and path
    continuation merge into the unsharded image bit for bit."""
    tex = sphere_volume(40, radius=0.4, seed=23)
    w, h = 120, 88
    kw = dict(mode=0, cloud_size_m=20000.0, max_depth=400)
    whole = ds.CloudTracer(tex, width=w, height=h, **kw)
    whole.render_accumulate(1, 9)
    merged = np.zeros_like(whole.mean())
    paths = 0
    for i in range(3):
        sh = ds.CloudTracer(tex, width=w, height=h, shard_index=i, shard_count=3, **kw)
        sh.render_accumulate_async(1, 4)
        sh.render_accumulate_async(5, 3)
        sh.render_accumulate_async(8, 2)
        merged += sh.mean()
        paths += sh.counters()["paths"]
        sh.close()
    assert np.array_equal(merged, whole.mean())
    assert paths == whole.counters()["paths"]
    whole.close()


def test_sharded_tracer_staged_async_path_single_rank():
    """The multi-GPU step without a second GPU: the handle shares a torch stream, every step enqueues
    render + accumulate + the copy of the running mean into the staging tensor (the RCCL reduce is a
    no-op for one rank).  The staging tensor must hold the running mean after synchronize()."""
    torch = pytest.importorskip("torch")
    from deepestscatter_amd.distributed import ShardedTracer
    tex = sphere_volume(32, seed=8)
    w, h = 40, 24
    st = ShardedTracer(tex, ds.SceneParams(width=w, height=h, mode=0), 0, 1, 0, stage_always=True)
    ref = ds.CloudTracer(tex, width=w, height=h, mode=0)
    first = 1
    for n in (2, 3, 2, 4):
        st.step_async(first, n)
        ref.render_accumulate(first, n)
        first += n
    st.synchronize()
    assert np.array_equal(st.merged_mean.cpu().numpy(), ref.mean())
    assert np.array_equal(st.merged_m2.cpu().numpy(), ref.m2())
    st.step(first, 2)
    ref.render_accumulate(first, 2)
    st.synchronize()
    assert np.array_equal(st.merged_mean.cpu().numpy(), ref.mean())
    assert np.array_equal(st.merged_m2.cpu().numpy(), ref.m2())
    # whole-frame quantities come from the merged buffers
    screen, avg = st.tonemap(0.4)
    ref_screen, ref_avg = ref.tonemap(0.4)
    assert np.array_equal(screen, ref_screen) and avg == ref_avg
    st.close()
    ref.close()


def test_merged_shards_tonemap_and_convergence_equal_the_single_gpu_job():
    """What an N-GPU job does at its end, on one GPU: three shard handles render their tiles (enqueued batches),
    their mean and M2 buffers are summed into one device buffer each (the RCCL reduce's arithmetic: foreign
    pixels are exactly 0), and rank 0's handle tonemaps the merged frame and counts its unconverged pixels
    (ct_tonemap_buffer / ct_is_converged_buffers).  Screen bytes, average luminance and the count must equal
    the single-GPU job's -- which a shard's own buffers cannot give (its foreign pixels are black and count as
    converged)."""
    torch = pytest.importorskip("torch")
    tex = sphere_volume(40, radius=0.42, seed=21)
    w, h, spp = 96, 72, 120
    kw = dict(mode=0, cloud_size_m=3000.0, max_depth=200)
    whole = ds.CloudTracer(tex, width=w, height=h, **kw)
    whole.render_accumulate(1, spp)
    want_screen, want_avg = whole.tonemap(0.4)
    want_conv = whole.is_converged()
    world = 3
    merged = torch.zeros((2, h, w, 4), dtype=torch.float32, device="cuda")
    shards = []
    for r in range(world):
        sh = ds.CloudTracer(tex, width=w, height=h, shard_index=r, shard_count=world, **kw)
        sh.render_accumulate_async(1, 50)
        sh.render_accumulate_async(51, spp - 50)
        merged[0] += torch.from_numpy(sh.mean()).cuda()
        merged[1] += torch.from_numpy(sh.m2()).cuda()
        shards.append(sh)
    torch.cuda.synchronize()
    assert np.array_equal(merged[0].cpu().numpy(), whole.mean()) and np.array_equal(merged[1].cpu().numpy(), whole.m2())
    screen, avg = shards[0].tonemap_buffer(merged[0].data_ptr(), 0.4)
    conv = shards[0].is_converged_buffers(merged[0].data_ptr(), merged[1].data_ptr(), spp)
    assert np.array_equal(screen, want_screen) and avg == want_avg
    assert conv == want_conv and want_conv[1] > 0
    own_screen, own_avg = shards[0].tonemap(0.4)              # a shard's own view is a different picture
    assert own_avg != want_avg and shards[0].is_converged()[1] < want_conv[1]
    for sh in shards:
        sh.close()
    whole.close()


def test_error_behaviour_of_the_newer_entry_points():
    """Descriptor gather, async batches: argument checks and state errors come back as CtStatus codes."""
    tex = sphere_volume(16)
    tr = ds.CloudTracer(tex, width=16, height=16)
    with pytest.raises(_lib.CloudTraceError) as e:
        tr.render_accumulate_async(3, 2)                 # out of order
    assert e.value.code == _lib.CT_E_STATE
    with pytest.raises(ValueError):
        tr.collect_descriptors(np.zeros((2, 3), np.float32), np.zeros((3, 3), np.float32))
    L = _lib.load()
    assert L.ct_collect_descriptors(tr.h, None, None, 4, None) == _lib.CT_E_INVAL
    assert L.ct_collect_descriptors(tr.h, None, None, 0, None) == _lib.CT_E_INVAL
    assert L.ct_copy_to_device_async(tr.h, _lib.CT_BUF_MEAN, None, 16 * 16 * 16) == _lib.CT_E_INVAL
    assert L.ct_synchronize(None) == _lib.CT_E_INVAL
    tr.render_accumulate_async(1, 2)                     # still usable
    tr.synchronize()
    assert tr.subframes == 2
    # a descriptor far outside the box is all zero; one at the centre of a cloud is not
    d = tr.collect_descriptors([[5.0, 5.0, 5.0], [0.0, 0.0, 0.0]], [[0, 0, 1], [0, 0, 1]])
    assert not d[0].any() and d[1].any()
    tr.close()


def test_descriptors_bit_exact_vs_oracle():
    """ct_collect_descriptors (setupHierarchicalDescriptor, DisneyDescriptor.cuh:71-112): the mip pyramid,
    the mip-linear trilinear sampler, the light/view frame and the fade outside the box, byte for byte."""
    for dims, size_m in (((40, 40, 40), 700.0), ((36, 52, 44), 3000.0), ((64, 64, 64), 7000.0)):
        tex = sphere_volume(dims=dims, seed=31)
        tr, orc = make_pair(tex, 8, 8, cloud_size_m=size_m)
        pos, view = tr.generate_scatter_samples(48, batch_seed=5)
        rng = np.random.default_rng(9)
        # plus points near / on / outside the faces and a view direction (anti)parallel-ish to the light
        extra = rng.uniform(-0.75, 0.75, (16, 3)).astype(np.float32)
        extra_v = rng.normal(size=(16, 3)).astype(np.float32)
        extra_v /= np.linalg.norm(extra_v, axis=1, keepdims=True)
        pos = np.concatenate([pos, extra])
        view = np.concatenate([view, extra_v])
        got = tr.collect_descriptors(pos, view)
        want = orc.collect_descriptors(pos, view)
        assert got.shape == (64, 10, 9, 5, 5)
        assert np.array_equal(got, want)
        assert got.any()
        tr.close()


def test_golden_fixture_on_gpu():
    from test_golden import load_golden
    g = load_golden()
    for case in g["cases"]:
        tex = g["volumes"][case["volume"]]
        tr = ds.CloudTracer(tex, width=case["width"], height=case["height"], mode=case["mode"])
        tr.render_accumulate(1, case["spp"])
        assert np.array_equal(tr.mean(), case["mean"]), case["name"]
        assert np.array_equal(tr.m2(), case["m2"]), case["name"]
        c = tr.counters()
        assert [c[k] for k in ("paths", "box_hits", "density_lookups", "inscatter_lookups", "scatter_events",
                               "depth_capped")] == list(case["counters"]), case["name"]
        assert np.array_equal(tr.inscatter(), g["inscatter"][case["volume"]])
        tr.close()


# ---- full-size properties (BASELINE.json config 2: 512^3 volume, 1024^2 frame) -------------------------
def test_golden_v2_fixture_on_gpu():
    """tests/golden/golden_v2.npz through the HIP path: DELTA renders, generated scatter samples, descriptors
    and point-radiance tasks, bit for bit."""
    from test_golden import GOLDEN_V2, COUNTER_KEYS
    from deepestscatter_amd.cloudtrace import make_point_tasks
    z = np.load(GOLDEN_V2)
    for entry in z["delta_case_names"]:
        cname, vname = str(entry).split(":")
        mode, w, h, spp = (int(v) for v in z[f"case_{cname}_meta"])
        tr = ds.CloudTracer(z[f"vol_{vname}"], width=w, height=h, mode=mode, estimator=1)
        tr.render_accumulate(1, spp)
        assert np.array_equal(tr.mean(), z[f"case_{cname}_mean"]) and np.array_equal(tr.m2(), z[f"case_{cname}_m2"])
        c = tr.counters()
        assert [c[k] for k in COUNTER_KEYS] == [int(v) for v in z[f"case_{cname}_counters"]]
        tr.close()
    tr = ds.CloudTracer(z["vol_v0"], width=8, height=8, mode=1, cloud_size_m=700.0)
    pos, view = tr.generate_scatter_samples(12, batch_seed=7)
    assert np.array_equal(pos, z["samples_pos"]) and np.array_equal(view, z["samples_dir"])
    assert np.array_equal(tr.collect_descriptors(pos, view), z["descriptors"])
    tasks = make_point_tasks(pos, view)
    tr.point_radiance_launch(tasks, 1, 6)
    assert np.array_equal(tasks.view(np.uint8).reshape(len(tasks), 40), z["point_tasks"])
    tr.close()


def test_full_size_properties_512_1024():
    n = 512
    tex = ds.make_procedural_cloud(n)
    w = h = 1024
    tr = ds.CloudTracer(tex, width=w, height=h)
    tr.render_accumulate(1, 2)
    m = tr.mean()
    c = tr.counters()
    assert np.isfinite(m).all() and np.all(m[..., :3] >= 0) and np.all(m[..., 3] == 1)
    assert c["paths"] == 2 * w * h and c["box_hits"] <= c["paths"]
    assert c["scatter_events"] == c["inscatter_lookups"] <= c["density_lookups"]
    # determinism: the schedule (persistent waves, atomics on the queue) never changes a pixel
    tr.reset()
    tr.render_accumulate(1, 2)
    assert np.array_equal(tr.mean(), m) and tr.counters() == c
    # oracle on a window of the same job (the full frame would take the CPU minutes)
    ins = tr.inscatter()
    orc = O.Oracle(tex, w, h, fast=True, inscatter=ins)
    win = (480, 500, 544, 532)
    ref, _ = orc.render(2, window=win)
    x0, y0, x1, y1 = win
    assert np.array_equal(m[y0:y1, x0:x1], ref[y0:y1, x0:x1])
    # shards tile the frame: two half-jobs merge into the whole
    merged = np.zeros_like(m)
    for i in range(2):
        sh = ds.CloudTracer(tex, width=w, height=h, shard_index=i, shard_count=2)
        sh.render_accumulate(1, 2)
        merged += sh.mean()
        sh.close()
    assert np.array_equal(merged, m)
    tr.close()


def test_north_star_parity_after_1024_spp_on_a_window():
    """BASELINE.json: "match the reference ... within 1e-3 relative L2 per pixel after 1024 spp" at 512^3 /
    1024^2.  The GPU renders the whole 1024-spp image the way bench.py does (four enqueued 256-subframe
    batches with path continuation); the oracle renders a 16x16 window in the dense body of the cloud for all
    1024 subframes.  Every pixel of the window must be bit-identical (relative L2 = 0 <= 1e-3)."""
    tex = ds.make_procedural_cloud(512)
    w = h = 1024
    tr = ds.CloudTracer(tex, width=w, height=h)
    first = 1
    for _ in range(4):
        tr.render_accumulate_async(first, 256)
        first += 256
    tr.synchronize()
    assert tr.debug_suspended() > 10000
    mean, m2 = tr.mean(), tr.m2()
    # The oracle does NOT borrow the product's shadow volume here: it evaluates inScatter (inScatter.cu:40-66) itself for
    # every texel its paths' NEE lookups touch (OrcScene::inscatter_valid), so the window checks the march AND the
    # shadow volume it reads; the touched texels are then compared with the product's volume one by one.
    orc = O.Oracle(tex, w, h, fast=True, inscatter="lazy")
    x0, y0 = 500, 520
    win = (x0, y0, x0 + 16, y0 + 16)
    ref_mean, ref_m2 = orc.render(1024, window=win)
    got, ref = mean[y0:y0 + 16, x0:x0 + 16], ref_mean[y0:y0 + 16, x0:x0 + 16]
    assert ref[..., 0].mean() > 0.5                     # the window is inside the cloud
    assert rel_l2(got, ref) <= 1e-3
    assert np.array_equal(got, ref)
    assert np.array_equal(m2[y0:y0 + 16, x0:x0 + 16], ref_m2[y0:y0 + 16, x0:x0 + 16])
    touched = orc.inscatter_valid.astype(bool)
    assert touched.sum() > 1_000_000                    # the window's paths wander through the whole body of the cloud
    assert np.array_equal(tr.inscatter()[touched], orc.inscatter[touched])
    tr.close()


@pytest.mark.parametrize("n,size,spp", [(512, 1024, 128), (144, 320, 24)])
def test_delta_estimator_on_a_window_with_coarse_majorant_cells(n, size, spp):
    """The DELTA estimator where its majorant grid has to coarsen to stay in LDS: 16-texel cells at 512^3 (the
    benchmark scene, rendered the way bench.py's DELTA leg does: enqueued batches with path continuation), 8-texel
    cells at 144^3.  The oracle renders a 16x16 window in the body of the cloud; mean and M2 must be bit-identical
    (the small volumes of the other DELTA tests all have 4-texel cells)."""
    tex = ds.make_procedural_cloud(n)
    w = h = size
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=1)
    half = spp // 2
    tr.render_accumulate_async(1, half)
    tr.render_accumulate_async(1 + half, spp - half)
    tr.synchronize()
    mean, m2 = tr.mean(), tr.m2()
    orc = O.Oracle(tex, w, h, fast=True, estimator=1, inscatter=tr.inscatter())
    assert orc.scene.maj_cell == (10 if n == 512 else 4)      # (round 4: the grid is cropped to the cloud; 16 / 8 before)
    x0, y0 = int(w * 0.49), int(h * 0.51)
    win = (x0, y0, x0 + 16, y0 + 16)
    ref_mean, ref_m2 = orc.render(spp, window=win)
    got, ref = mean[y0:y0 + 16, x0:x0 + 16], ref_mean[y0:y0 + 16, x0:x0 + 16]
    assert ref[..., 0].mean() > 0.3                     # the window is inside the cloud
    assert np.array_equal(got, ref)
    assert np.array_equal(m2[y0:y0 + 16, x0:x0 + 16], ref_m2[y0:y0 + 16, x0:x0 + 16])
    tr.close()


def _random_scene(rng):
    dims = tuple(int(v) for v in rng.integers(5, 41, 3))
    nz, ny, nx = dims[::-1]
    kind = int(rng.integers(0, 4))
    if kind == 0:                                         # blobs
        t = np.zeros((nz, ny, nx), np.float32)
        z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        for _ in range(int(rng.integers(1, 5))):
            c = rng.uniform(0.2, 0.8, 3) * (nz, ny, nx)
            r = rng.uniform(0.1, 0.35) * min(dims)
            t += np.clip(1.0 - np.sqrt((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) / r, 0, 1)
        tex = (np.clip(t, 0, 1) * 255).astype(np.uint8)
    elif kind == 1:                                       # sparse specks
        tex = np.where(rng.random((nz, ny, nx)) < rng.uniform(0.002, 0.05), rng.integers(1, 256, (nz, ny, nx)), 0).astype(np.uint8)
    elif kind == 2:                                       # dense noise, no border
        tex = rng.integers(0, 256, (nz, ny, nx)).astype(np.uint8)
    else:                                                 # slabs and holes
        tex = np.zeros((nz, ny, nx), np.uint8)
        tex[:, :, nx // 3: 2 * nx // 3] = rng.integers(20, 255)
        tex[nz // 4: nz // 2, ny // 4: ny // 2, :] = 0
    if kind != 2 and rng.random() < 0.7:                  # the reference's zero border
        tex[0], tex[-1], tex[:, 0], tex[:, -1], tex[:, :, 0], tex[:, :, -1] = 0, 0, 0, 0, 0, 0
    light = rng.normal(size=3)
    eye = rng.normal(size=3)
    eye = eye / np.linalg.norm(eye) * rng.uniform(0.3, 3.0)    # inside the box now and then
    return dict(tex=tex, width=int(rng.integers(9, 49)), height=int(rng.integers(9, 41)),
                mode=int(rng.integers(0, 3)), estimator=int(rng.random() < 0.3),
                cloud_size_m=float(rng.choice([80.0, 700.0, 7000.0, 40000.0])),
                sample_step=float(rng.choice([1 / 512, 1 / 256, 1 / 100])),
                max_depth=int(rng.choice([2, 17, 300, 2000])), light_direction=tuple(float(v) for v in light)), tuple(float(v) for v in eye)


def test_randomized_differential_against_the_oracle():
    """Random volumes (blobs, specks, dense noise without a border, slabs with holes), frame sizes, modes,
    estimators, step sizes, depth caps, lights and eyes (inside the box too), rendered with a random mix of
    synchronous and enqueued batches: radiance, M2 and every counter must equal the oracle's."""
    rng = np.random.default_rng(20261003)
    for case in range(24):
        kw, eye = _random_scene(rng)
        tex = kw.pop("tex")
        w, h = kw.pop("width"), kw.pop("height")
        tr, orc = make_pair(tex, w, h, **kw)
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
        tr.set_camera(eye, U, V, W)
        orc.set_camera(eye, U, V, W)
        first = 1
        for n in rng.integers(1, 5, 4):
            if rng.random() < 0.6:
                tr.render_accumulate_async(first, int(n))
            else:
                tr.render_accumulate(first, int(n))
            first += int(n)
        mean, m2 = orc.render(first - 1)
        tag = f"case {case}: dims {tex.shape[::-1]} {w}x{h} {kw} eye {eye}"
        assert np.array_equal(tr.mean(), mean), tag
        assert np.array_equal(tr.m2(), m2), tag
        assert tr.counters() == orc.counters.as_dict(), tag
        tr.close()


def test_randomized_continuation_soak():
    """Random scenes at frame sizes where tens of thousands of paths cross launch boundaries: enqueued batches
    (with an occasional synchronize in between) against synchronous ones on the HIP path -- mean, M2 and every
    counter bit for bit.  (tools/soak_continuation.py runs hundreds of these; an early version that let a lane
    suspend several paths per launch lost samples in 4 cases of 150 and was found this way.)"""
    rng = np.random.default_rng(77)
    suspended = 0
    for case in range(12):
        kw, eye = _random_scene(rng)
        tex = kw.pop("tex")
        kw.pop("width"), kw.pop("height")
        w, h = int(rng.integers(96, 320)), int(rng.integers(64, 256))
        kw.update(estimator=0, sample_step=1.0 / 512, cloud_size_m=float(rng.choice([7000.0, 20000.0, 40000.0])),
                  max_depth=int(rng.choice([50, 300, 2000])))
        a = ds.CloudTracer(tex, width=w, height=h, **kw)
        b = ds.CloudTracer(tex, width=w, height=h, **kw)
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
        a.set_camera(eye, U, V, W)
        b.set_camera(eye, U, V, W)
        first = 1
        for n in rng.integers(1, 9, int(rng.integers(2, 7))):
            a.render_accumulate_async(first, int(n))
            b.render_accumulate(first, int(n))
            first += int(n)
            if rng.random() < 0.2:
                a.synchronize()
        tag = f"case {case}: dims {tex.shape[::-1]} {w}x{h} {kw} eye {eye}"
        assert np.array_equal(a.mean(), b.mean()) and np.array_equal(a.m2(), b.m2()), tag
        assert a.counters() == b.counters(), tag
        suspended += a.debug_suspended()
        a.close()
        b.close()
    assert suspended > 100000


@pytest.mark.parametrize("estimator", [0, 1])
def test_scheduler_knobs_never_change_a_result(monkeypatch, estimator):
    """Every CT_* knob only changes the schedule (burst lengths, regeneration and scatter thresholds, per-XCD
    queues, blocks per CU, the pre-walked prefix, path continuation, the queue-empty hint): mean, M2 and the
    counters of enqueued batches must be identical under any of them -- for both estimators (round 1 ran MARCH
    only, and the DELTA kernel lost the jobs of seven of the eight per-XCD queues in enqueued batches: it read the
    "last job taken" flag as "every queue is empty").  The runs under a knob also arm CT_DEBUG_INVARIANTS: NaN-filled
    scratch, path conservation, no sample without alpha 1."""
    tex = sphere_volume(44, radius=0.38, seed=41)
    w, h = 200, 144
    kw = dict(mode=0, cloud_size_m=20000.0, max_depth=500, estimator=estimator)

    def run():
        tr = ds.CloudTracer(tex, width=w, height=h, **kw)
        tr.render_accumulate_async(1, 5)
        tr.render_accumulate_async(6, 3)
        tr.render_accumulate_async(9, 6)
        out = (tr.mean(), tr.m2(), tr.counters())
        iv = tr.debug_invariants()
        assert iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
        if iv["armed"]:
            assert iv["checks"] >= 1 and iv["dealt"] == iv["written"] > 0 and iv["resumed"] == iv["suspended"], iv
        tr.close()
        return out

    base = run()
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    armed = run()
    assert np.array_equal(armed[0], base[0]) and np.array_equal(armed[1], base[1]) and armed[2] == base[2]
    settings = [
        {"CT_MARCH_BURST": "1"}, {"CT_MARCH_BURST": "3", "CT_BURST_SCATTER": "5", "CT_BURST_IDLE": "7"},
        {"CT_MARCH_BURST": "64", "CT_BURST_SCATTER": "65", "CT_BURST_IDLE": "65"},
        {"CT_REGEN_MIN": "1"}, {"CT_REGEN_MIN": "64"}, {"CT_SCATTER_MIN": "20"}, {"CT_SCATTER_RATIO": "1/2"},
        {"CT_XCD_QUEUES": "1"}, {"CT_XCD_QUEUES": "1", "CT_XCD_REGIONS": "8", "CT_SHARED_DEPTH": "4"},
        {"CT_BLOCKS_PER_CU": "1"}, {"CT_BLOCKS_PER_CU": "3"},
        {"CT_NO_ADVANCE": "1"}, {"CT_CONTINUATION": "0"}, {"CT_HINT_PERIOD": "1"}, {"CT_HINT_PERIOD": "0"},
        {"CT_TAIL_BURST": "1"}, {"CT_BURST_MARCH_MIN": "20"}, {"CT_NEE_CACHE": "0"},
        {"CT_SPARSE": "1"}, {"CT_SPARSE": "1", "CT_XCD_QUEUES": "1", "CT_NO_ADVANCE": "1"},
        # the DELTA kernel's fetch layouts (round 4): separate bricks / shadow footprint requested at the collision / twin bricks
        {"CT_DELTA_NEE": "0"}, {"CT_DELTA_NEE": "1"}, {"CT_DELTA_NEE": "2"}, {"CT_DELTA_NEE": "2", "CT_CONTINUATION": "0"},
        {"CT_DELTA_NEE": "2", "CT_XCD_QUEUES": "1", "CT_NO_ADVANCE": "1"},
        # the DELTA kernel with the box test of a real collision (this sphere lies inside the volume: the default runs without it)
        {"CT_DELTA_INTERIOR": "0"}, {"CT_DELTA_INTERIOR": "0", "CT_CONTINUATION": "0"},
    ]
    for env in settings:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), env
        assert got[2] == base[2], env


@pytest.mark.parametrize("nee", ["0", "1", "2"])
def test_delta_fetch_layouts_against_the_oracle(nee, monkeypatch):
    """render_delta_kernel<.., NEE>: where a collision's density and shadow-volume footprints come from -- their own apron
    bricks with the shadow fetch in the scatter phase (0), the same arrays with the shadow footprint requested in the tracking
    visit (1), twin bricks of 3^3 base texels that hold both in one 128-byte line (2).  Same values whatever the layout: mean,
    M2 and counters equal the oracle twin's on non-cubic volumes, a volume without a zero border (flights run into the apron,
    where the twin grid's clamp applies), all three radiance programs, and with paths suspended between their collision and
    their bounce (enqueued batches: the resumed path requests its shadow footprint again)."""
    monkeypatch.setenv("CT_DELTA_NEE", nee)
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    rng = np.random.default_rng(7)
    cases = [
        (sphere_volume(dims=(23, 31, 17), seed=5), 40, 28, dict(mode=0, cloud_size_m=9000.0)),
        (rng.integers(0, 256, (14, 19, 26)).astype(np.uint8), 33, 21, dict(mode=0, cloud_size_m=300.0, max_depth=60)),   # no border
        (sphere_volume(dims=(40, 40, 40), radius=0.45, seed=9), 36, 36, dict(mode=1, cloud_size_m=4000.0, max_depth=300)),
        (sphere_volume(dims=(29, 29, 29), seed=11), 32, 24, dict(mode=2)),
    ]
    for tex, w, h, kw in cases:
        tr, orc = make_pair(tex, w, h, estimator=1, **kw)
        tr.render_accumulate_async(1, 3)
        tr.render_accumulate_async(4, 2)
        tr.render_accumulate(6, 4)
        tr.render_accumulate_async(10, 5)
        mean, m2 = orc.render(14)
        assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2), (nee, tex.shape, kw)
        assert tr.counters() == orc.counters.as_dict(), (nee, tex.shape, kw)
        iv = tr.debug_invariants()
        assert iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
        tr.close()


@pytest.mark.parametrize("nee", ["1", "2"])
def test_delta_kernel_without_the_box_test_where_the_cloud_lies_inside_the_volume(monkeypatch, nee):
    """DevScene::delta_interior: when every non-zero texel lies two texels or more inside the volume's faces a REAL collision --
    a position with a non-zero footprint -- is inside the box, and render_delta_kernel<.., INTERIOR = true> (no isInBox after
    a real collision) runs.  The flag is set exactly at that margin; with it the results equal the oracle twin's (which always
    tests) bit for bit on a volume that is dense right up to the margin, seen from outside and from inside, in all three
    radiance programs; CT_DELTA_INTERIOR=0 runs the kernel with the test and gives the same."""
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    monkeypatch.setenv("CT_DELTA_NEE", nee)   # (apron bricks with the shadow footprint requested at the collision / twin bricks)
    rng = np.random.default_rng(23)

    def slab(n, lo, hi):
        t = np.zeros((n, n, n), np.uint8)
        t[lo:hi + 1, lo:hi + 1, lo:hi + 1] = rng.integers(1, 256, (hi - lo + 1,) * 3).astype(np.uint8)
        return t

    for tex, want in ((slab(28, 6, 21), True), (slab(28, 1, 21), False), (slab(28, 6, 26), False), (rng.integers(0, 256, (14, 19, 26)).astype(np.uint8), False),
                      (sphere_volume(44, radius=0.38, seed=41), True)):
        tr = ds.CloudTracer(tex, width=16, height=16, estimator=1)
        assert tr.delta_grid()["interior"] is want and tr.delta_grid()["nee"] == int(nee)
        tr.close()
    for tex, mode, eye in [(slab(28, 6, 21), m, e) for m in (0, 1, 2) for e in ((0.3, 0.25, 3.0), (0.05, 0.02, 0.1))] + \
                          [(slab(20, 2, 17), 0, (0.3, 0.25, 3.0)), (slab(24, 3, 20), 0, (-2.0, 0.4, 0.3))]:   # (at and near the margin: whichever kernel runs)
        kw = dict(mode=mode, cloud_size_m=600.0, max_depth=200, estimator=1)
        outs = []
        for knob in (None, "0"):
            if knob is not None:
                monkeypatch.setenv("CT_DELTA_INTERIOR", knob)
            tr, orc = make_pair(tex, 40, 30, **kw)
            assert tr.delta_grid()["interior"] is (knob is None) or tex.shape[0] != 28
            assert knob is None or not tr.delta_grid()["interior"]
            U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, 40 / 30)
            tr.set_camera(eye, U, V, W)
            orc.set_camera(eye, U, V, W)
            tr.render_accumulate_async(1, 4)
            tr.render_accumulate(5, 3)
            tr.render_accumulate_async(8, 5)
            mean, m2 = orc.render(12)
            outs.append((tr.mean(), tr.m2(), tr.counters()))
            assert np.array_equal(outs[-1][0], mean) and np.array_equal(outs[-1][1], m2), (mode, eye, knob)
            assert outs[-1][2] == orc.counters.as_dict(), (mode, eye, knob)
            iv = tr.debug_invariants()
            assert iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
            tr.close()
            if knob is not None:
                monkeypatch.delenv("CT_DELTA_INTERIOR")
        assert outs[0][2] == outs[1][2] and outs[0][2]["scatter_events"] > 0


def test_multi_gpu_step_on_the_rccl_backend_single_rank():
    """tools/nccl_single_rank_check.py in a child process: the enqueued multi-GPU step (accumulate, copy of the
    running mean, dist.reduce on the shared torch stream) with backend "nccl" = RCCL, world size 1."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parents[1] / "tools" / "nccl_single_rank_check.py"
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL single-rank check: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("estimator", [0, 1])
def test_xcd_queues_with_enqueued_batches_lose_no_job(monkeypatch, estimator):
    """Per-XCD job queues (the default for volumes >= 768^3, CT_XCD_QUEUES=1 anywhere) with enqueued batches whose
    paths cross launches, invariants armed: results and every counter equal the synchronous single-queue run's and
    the oracle's on a window.  This is the configuration in which the DELTA kernel dropped jobs in round 1."""
    tex = sphere_volume(56, radius=0.4, seed=5)
    w, h = 320, 240
    kw = dict(mode=0, cloud_size_m=20000.0, max_depth=800, estimator=estimator)
    ref = ds.CloudTracer(tex, width=w, height=h, **kw)
    ref.render_accumulate(1, 6)
    ref.render_accumulate(7, 10)
    want = (ref.mean(), ref.m2(), ref.counters())
    ins = ref.inscatter()
    ref.close()
    monkeypatch.setenv("CT_XCD_QUEUES", "1")
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    tr = ds.CloudTracer(tex, width=w, height=h, **kw)
    tr.render_accumulate_async(1, 6)
    tr.render_accumulate_async(7, 10)
    got = (tr.mean(), tr.m2(), tr.counters())
    iv = tr.debug_invariants()
    assert iv["armed"] == 1 and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
    assert iv["dealt"] == iv["written"] == want[2]["box_hits"] and iv["resumed"] == iv["suspended"] > 0, iv
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    orc = O.Oracle(tex, w, h, fast=True, estimator=estimator, inscatter=ins, cloud_size_m=20000.0, max_depth=800)
    win = (150, 110, 166, 126)
    ref_mean, ref_m2 = orc.render(16, window=win)
    x0, y0, x1, y1 = win
    assert np.array_equal(got[0][y0:y1, x0:x1], ref_mean[y0:y1, x0:x1])
    assert np.array_equal(got[1][y0:y1, x0:x1], ref_m2[y0:y1, x0:x1])
    tr.close()


def test_fetch_counters_are_what_the_kernels_issue(monkeypatch):
    """ct_fetch_counters: the MARCH estimator issues fewer fetches than the algorithm counts lookups (free-space
    replay, pre-walked prefix, reused shadow-volume footprints) and exactly as many as its diagnostics build tallies
    lane by lane; DELTA and the one-thread-per-pixel kernel issue one fetch per lookup."""
    tex = ds.make_procedural_cloud(96)
    w, h = 160, 128
    monkeypatch.setenv("CT_STATS", "1")
    tr = ds.CloudTracer(tex, width=w, height=h)
    tr.render_accumulate(1, 40)
    tr.render_accumulate_async(41, 24)
    c, f, st = tr.counters(), tr.fetch_counters(), tr.debug_stats()
    tr.close()
    monkeypatch.delenv("CT_STATS")
    assert 0 < f["density_fetches"] < c["density_lookups"]
    assert 0 < f["inscatter_fetches"] <= c["inscatter_lookups"]
    assert f["density_fetches"] == st["fetched_steps"]
    assert f["inscatter_fetches"] == c["inscatter_lookups"] - st["nee_footprints_reused"]
    plain = ds.CloudTracer(tex, width=w, height=h)          # the production build counts the same
    plain.render_accumulate(1, 40)
    plain.render_accumulate_async(41, 24)
    assert plain.fetch_counters() == f and plain.counters() == c
    plain.close()
    for kw in (dict(estimator=1), dict(flags=_lib.CT_FLAG_SIMPLE_KERNEL)):
        t = ds.CloudTracer(tex, width=w, height=h, **kw)
        t.render_accumulate(1, 4)
        c, f = t.counters(), t.fetch_counters()
        assert f == {"density_fetches": c["density_lookups"], "inscatter_fetches": c["inscatter_lookups"]}, kw
        t.close()


def test_sparse_march_bricks_are_bit_exact_and_smaller(monkeypatch):
    """Sparse march bricks (the storage of volumes >= 768^3 texels; CT_SPARSE=1 forces it on small ones): only the
    bricks between the first and the last non-empty one of every brick row are stored, a footprint outside that extent
    is zero by construction and takes its clearance from the coarse grid.  Radiance, M2 and the algorithm's counters
    must equal the oracle's on the volumes that stress the clearance codes (isolated texels, one-texel shells,
    non-zero faces, x extents that are not multiples of 3, no zero border, a small cloud in a big box, a slab with a
    hole, fewer texels than a brick), from several directions, with and without the pre-walked prefix."""
    monkeypatch.setenv("CT_SPARSE", "1")
    rng = np.random.default_rng(4242)
    hole = np.zeros((40, 36, 44), np.uint8)
    hole[:, :, 14:30] = 120
    hole[10:20, 9:18, :] = 0
    hole[0], hole[-1], hole[:, 0], hole[:, -1], hole[:, :, 0], hole[:, :, -1] = 0, 0, 0, 0, 0, 0
    two = np.zeros((48, 40, 64), np.uint8)                      # two blobs in one brick row: the gap between them is stored
    two[20:28, 16:24, 6:14] = 200
    two[22:26, 18:22, 44:58] = 90
    cases = [(_speck_volume((49, 50, 52), seed=77), 900.0), (_speck_volume((64, 37, 45), seed=78), 900.0),
             (sphere_volume(dims=(96, 72, 80), radius=0.12, seed=21), 7000.0), (hole, 3000.0), (two, 7000.0),
             (rng.integers(0, 256, (5, 7, 4)).astype(np.uint8), 50.0), (rng.integers(0, 256, (2, 2, 2)).astype(np.uint8), 50.0),
             (np.zeros((20, 20, 20), np.uint8), 7000.0)]
    stored = dense = 0
    for i, (tex, size) in enumerate(cases):
        w, h = 40, 32
        tr, orc = make_pair(tex, w, h, mode=0, cloud_size_m=size)
        mem = tr.debug_memory()
        assert mem["sparse"] == 1 and mem["march_bricks_stored"] <= mem["march_bricks_dense"]
        stored += mem["march_bricks_stored"]
        dense += mem["march_bricks_dense"]
        mean, m2 = orc.render(3)
        tr.render_accumulate(1, 2)
        tr.render_accumulate_async(3, 1)
        assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2), i
        assert tr.counters() == orc.counters.as_dict(), i
        for eye in ((0.3, 2.2, 0.4), (-0.4, -0.2, -2.4), (-2.3, 0.3, 0.2), (0.1, 0.05, -0.2)):
            U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
            tr.set_camera(eye, U, V, W)
            orc.set_camera(eye, U, V, W)
            tr.render_subframe(7)
            assert np.array_equal(tr.frame(), orc.render_subframe(7)), (i, eye)
        assert tr.counters() == orc.counters.as_dict(), i
        tr.close()
    assert stored < 0.6 * dense
    # the same through the dense array: identical everything (the storage is not part of the result)
    monkeypatch.setenv("CT_SPARSE", "0")
    tex, size = cases[2]
    a = ds.CloudTracer(tex, width=64, height=48, cloud_size_m=size)
    monkeypatch.setenv("CT_SPARSE", "1")
    b = ds.CloudTracer(tex, width=64, height=48, cloud_size_m=size)
    assert a.debug_memory()["sparse"] == 0 and b.debug_memory()["sparse"] == 1
    for t in (a, b):
        t.render_accumulate_async(1, 6)
        t.render_accumulate_async(7, 5)
    assert np.array_equal(a.mean(), b.mean()) and np.array_equal(a.m2(), b.m2()) and a.counters() == b.counters()
    # fewer line fetches are issued for the same lookups?  not necessarily fewer -- but never a different algorithm count
    a.close()
    b.close()


def test_group_api_merges_shards_below_the_c_abi(tmp_path):
    """ct_group_*: one process, one handle per entry of the device list, frame reduce of [mean | M2] onto the first
    device.  With one GPU on the box: (a) a group of one device goes through a real RCCL communicator (ncclCommInitAll
    + ncclReduce of one rank); (b) a device list that repeats device 0 three times rehearses three shards -- RCCL
    refuses two ranks on one device, so that group merges with copies and an add kernel.  Merged mean, M2, counters,
    tonemapped screen, average luminance and the convergence count must equal the single handle's; so must the C++
    host's picture with --gpus 0,0."""
    tex = sphere_volume(40, radius=0.42, seed=33)
    w, h, spp = 88, 64, 110
    kw = dict(mode=0, cloud_size_m=3000.0, max_depth=200)
    one = ds.CloudTracer(tex, width=w, height=h, **kw)
    one.render_accumulate(1, 60)
    one.render_accumulate(61, spp - 60)
    want = (one.mean(), one.m2(), one.counters(), one.tonemap(0.4), one.is_converged())
    one.close()
    for devices in ([0], [0, 0, 0]):
        g = ds.TracerGroup(tex, devices, width=w, height=h, **kw)
        g.render_accumulate(1, 60)
        g.render_accumulate(61, spp - 60)
        assert np.array_equal(g.mean(), want[0]) and np.array_equal(g.m2(), want[1]), devices
        assert g.counters() == want[2], devices
        screen, avg = g.tonemap(0.4)
        assert np.array_equal(screen, want[3][0]) and avg == want[3][1], devices
        assert g.is_converged() == want[4] and want[4][1] > 0, devices
        g.reset()
        g.render_accumulate(1, 3)
        assert g.counters()["paths"] == 3 * w * h
        g.close()
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.TracerGroup(tex, [0, 99], width=w, height=h, **kw)
    assert e.value.code == _lib.CT_E_INVAL and "shard 1" in e.value.message
    # the C++ host: cloudtrace --gpus 0,0 against the single-GPU picture
    import subprocess
    from deepestscatter_amd import build, exr
    cli = build.build_cli()
    outs = []
    for extra in ([], ["--gpus", "0,0"]):
        out = tmp_path / ("g" if extra else "s")
        out.mkdir()
        r = subprocess.run([str(cli), "procedural:32", "--size", "48x32", "--spp", "45", "--light", "Side", "--out", str(out), *extra],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        outs.append(exr.read_exr(out / "procedural_32.Side.PT.exr"))
    assert np.array_equal(outs[0], outs[1]) and outs[0].max() > 0


def test_delta_single_scatter_from_a_camera_that_looks_away_from_the_box():
    """Found by the armed soak (seed 3003, case 634; 61 500 cases in round 2): the reference's box test reports a hit at
    t = 1e-6 for a ray whose BACKWARD extension meets the box (cloudBBox.cu:26-33), so a camera just outside a box it
    looks away from starts its paths outside the slack box.  The march does nothing there (its loop condition,
    cloud.cuh:87), and so does the DELTA kernel; the DELTA oracle twin lacked that test in singleScatterSunRadiance
    (no bounce loop around the flight, cloudRadianceMaterials.cu:134) and walked the apron cells, counting lookups
    that can never contribute: identical images, 5 % more density lookups.  Fixed in the oracle; this is the case."""
    rng = np.random.default_rng(3003)
    for _ in range(635):                                   # tools/soak.py's draws, up to the reported case
        kw, eye = _random_scene(rng)
        rng.random()
        pattern = [(int(n), bool(rng.random() < 0.6)) for n in rng.integers(1, 5, 4)]
    tex = kw.pop("tex")
    w, h = kw.pop("width"), kw.pop("height")
    assert (kw["mode"], kw["estimator"], tex.shape) == (2, 1, (25, 6, 28))
    tr, orc = make_pair(tex, w, h, **kw)
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    tr.set_camera(eye, U, V, W)
    orc.set_camera(eye, U, V, W)
    first = 1
    for n, a in pattern:
        (tr.render_accumulate_async if a else tr.render_accumulate)(first, n)
        first += n
    mean, m2 = orc.render(first - 1)
    assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2)
    assert tr.counters() == orc.counters.as_dict()
    c = tr.counters()
    assert c["box_hits"] < c["paths"] and c["density_lookups"] == 897531
    tr.close()


def test_fast_rcp_and_sqrt_are_correctly_rounded_for_every_float_in_range():
    """rcp_moderate / sqrt_moderate (ct_device.hpp: the hardware's 1-ulp estimate + one fused Newton step, 3 and 4
    instructions instead of hipcc's 10-instruction IEEE sequences) must return the bits of 1.0f / x and sqrtf(x) -- the
    oracle computes those on the CPU -- for EVERY float of their stated range.  Checked exhaustively on the device."""
    tr = ds.CloudTracer(np.zeros((4, 4, 4), np.uint8), width=8, height=8)
    r = tr.debug_math_selftest(0)
    assert r["tested"] == 2 * 121 * (1 << 23) and r["mismatches"] == 0, r
    s = tr.debug_math_selftest(1)
    assert s["tested"] == 121 * (1 << 23) and s["mismatches"] == 0, s
    tr.close()


@pytest.mark.gpu
def test_handles_driven_from_several_host_threads_at_once(monkeypatch):
    """`cloudtrace collect --jobs K` keeps K renderer handles busy from K host threads; the library has no state outside
    a handle, so whatever the threads do at the same time -- enqueued frame batches of both estimators, point-radiance
    launches, scatter samples, descriptors, creating and destroying handles -- every result must be what the same
    calls give one after the other.  Invariants armed (a lost or doubled sample fails the call)."""
    import threading
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    volumes = [sphere_volume(36 + 4 * i, radius=0.3 + 0.02 * i, seed=60 + i) for i in range(4)]

    def job(i):
        out = {}
        for est in (0, 1):
            tr = ds.CloudTracer(volumes[i], width=96 + 8 * i, height=64, mode=0, cloud_size_m=15000.0, max_depth=300, estimator=est)
            tr.render_accumulate_async(1, 4)
            tr.render_accumulate_async(5, 3 + i)
            out[f"mean{est}"], out[f"m2{est}"], out[f"counters{est}"] = tr.mean(), tr.m2(), tr.counters()
            if est == 0:
                pos, d = tr.generate_scatter_samples(128, 7 * i)
                tasks = ds.make_point_tasks(np.repeat(pos, 3, axis=0), np.repeat(d, 3, axis=0), ids=np.repeat(np.arange(128), 3))
                for k in range(3):
                    tasks = tr.point_radiance_launch(tasks, 1 + 5 * k, 5)
                out["samples"], out["tasks"] = (pos, d), tasks
                out["descriptors"] = tr.collect_descriptors(pos[:16], d[:16])
            iv = tr.debug_invariants()
            assert iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
            tr.close()
        return out

    serial = [job(i) for i in range(4)]
    for _ in range(2):
        results, errors = [None] * 4, []

        def run(i):
            try:
                results[i] = job(i)
            except Exception as e:          # noqa: BLE001 -- reported below, with the thread's index
                errors.append((i, repr(e)))

        threads = [threading.Thread(target=run, args=(i,)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for i in range(4):
            for k, v in serial[i].items():
                got = results[i][k]
                if isinstance(v, dict):
                    assert got == v, (i, k)
                elif isinstance(v, tuple):
                    assert all(np.array_equal(a, b) for a, b in zip(got, v)), (i, k)
                else:
                    assert got.tobytes() == v.tobytes(), (i, k)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 0])
def test_delta_flights_parallel_to_an_axis_or_nearly_so(mode):
    """The DELTA kernel's DDA set-up takes a three-instruction reciprocal unless a lane of the wave has a direction
    component below 2^-60 (then the whole wave takes the compiler's IEEE sequence).  Cameras on the coordinate axes make
    primary rays with components that are exactly zero (the middle row and column of an even frame), cameras a hair off
    the axes make tiny ones; waves mix such lanes with ordinary ones.  Every result must be the oracle's either way."""
    tex = sphere_volume(40, radius=0.42, seed=77)
    w, h = 64, 48
    eyes = [(0.0, 0.0, 2.5), (2.5, 0.0, 0.0), (0.0, -2.5, 0.0), (1e-24, 0.0, 2.5), (0.0, 3e-30, -2.5), (2.5, 1e-21, 1e-26)]
    for eye in eyes:
        tr, orc = make_pair(tex, w, h, mode=mode, estimator=1, cloud_size_m=4000.0, max_depth=40)
        up = (0, 1, 0) if abs(eye[1]) < 1 else (0, 0, 1)
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), up, 30.0, w / h)
        tr.set_camera(eye, U, V, W)
        orc.set_camera(eye, U, V, W)
        tr.render_accumulate(1, 3)
        mean, m2 = orc.render(3)
        assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2), eye
        c, o = tr.counters(), orc.counters.as_dict()
        assert all(c[k] == o[k] for k in ("paths", "box_hits", "density_lookups", "inscatter_lookups", "scatter_events", "depth_capped")), (eye, c, o)
        tr.close()


@pytest.mark.parametrize("estimator", [0, 1])
@pytest.mark.parametrize("max_age", ["0", "1", "5"])
def test_short_batches_with_repeated_suspension_are_bit_exact(estimator, max_age, monkeypatch):
    """The reference renders 10 subframes per display update (Camera.cpp:189).  Batches that short are enqueued with a ring of
    scratch regions: a path may be handed from launch to launch several times (BatchArgs::max_age) and a batch is
    accumulated once max_age further launches have run.  Whatever the ring's length (CT_MAX_AGE: 0 = chosen from the batch's
    duration, 1 = round 2's two regions, 5 = six regions) the frame equals the waited-for one bit for bit, every sample is
    written exactly once (invariants armed: NaN-filled scratch, dealt + resumed == written + suspended), and the display
    update enqueued behind the batches (ct_tonemap_async) shows what ct_tonemap shows."""
    tex = ds.make_procedural_cloud(96)
    w = h = 160
    ref = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    ref.render_accumulate(1, 100)
    want = (ref.mean(), ref.m2(), ref.counters(), ref.tonemap(0.4)[0])
    ref.close()
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    monkeypatch.setenv("CT_MAX_AGE", max_age)
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    first = 1
    for n in [10] * 4 + [3] * 10 + [10] * 3:        # (a change of batch size re-lays the ring out: a flush in between)
        tr.render_accumulate_async(first, n)
        tr.tonemap_async(0.4)
        first += n
    tr.synchronize()
    tr.tonemap_async(0.4)
    tr.synchronize()
    screen = tr.download(_lib.CT_BUF_SCREEN)
    got = (tr.mean(), tr.m2(), tr.counters())
    iv = tr.debug_invariants()
    suspended = tr.debug_suspended()
    tr.close()
    assert iv["armed"] == 1 and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
    assert iv["resumed"] == iv["suspended"] == suspended > 0, (iv, suspended)
    assert got[2] == want[2]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(np.asarray(screen).reshape(h, w, 4), want[3])


@pytest.mark.parametrize("estimator", [0, 1])
@pytest.mark.parametrize("max_age", ["0", "3"])
def test_render_ahead_serves_the_display_cadence_bit_exactly(estimator, max_age, monkeypatch):
    """ct_set_render_ahead: enqueued calls of 10 subframes (Camera.cpp:189) are served by launches of 40, every call
    accumulating its own share.  Whenever something waits, the running mean is the one of exactly the subframes asked for --
    the waited-for handle's frame at that count, bit for bit -- although the estimator has run ahead; waited-for calls use up
    what was rendered ahead; a new pose drops it.  Invariants armed (every sample written exactly once)."""
    tex = ds.make_procedural_cloud(96)
    w = h = 160
    ref = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    want, first = {}, 1
    for upto in (30, 70, 130, 160):
        ref.render_accumulate(first, upto - first + 1)
        want[upto] = (ref.mean(), ref.m2(), ref.counters(), ref.tonemap(0.4)[0])
        first = upto + 1
    eye = (0.3, 2.2, 0.9)
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    ref.set_camera(eye, U, V, W)
    ref.reset()
    ref.render_accumulate(1, 20)
    want["pose2"] = (ref.mean(), ref.m2())
    ref.close()

    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    monkeypatch.setenv("CT_MAX_AGE", max_age)
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    tr.set_render_ahead(40)

    def updates(first, n):
        for k in range(n):
            tr.render_accumulate_async(first + 10 * k, 10)
            tr.tonemap_async(0.4)

    def same(upto):
        return np.array_equal(tr.mean(), want[upto][0]) and np.array_equal(tr.m2(), want[upto][1])

    updates(1, 3)                     # the cost-measuring launch of 10, then a launch of 40: subframes 11..50
    tr.synchronize()
    assert tr.rendered_subframes() == 50 and same(30)
    updates(31, 4)                    # 31..50 are there; 51..90 are launched by the third of these calls
    assert same(70) and tr.rendered_subframes() == 90
    tr.tonemap_async(0.4)
    tr.synchronize()
    assert np.array_equal(np.asarray(tr.download(_lib.CT_BUF_SCREEN)).reshape(h, w, 4), want[70][3])
    tr.render_accumulate(71, 60)      # a waited-for call: 20 subframes from the scratch, 40 rendered now
    assert same(130) and tr.rendered_subframes() == 130
    assert tr.counters() == want[130][2]            # (nothing rendered that was not asked for)
    updates(131, 3)
    assert same(160) and tr.rendered_subframes() == 170
    assert tr.counters()["paths"] == want[160][2]["paths"] // 160 * 170
    iv = tr.debug_invariants()
    assert iv["armed"] == 1 and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
    assert iv["resumed"] == iv["suspended"] > 0, iv
    tr.set_camera(eye, U, V, W)      # the ten subframes rendered ahead are of the old pose
    tr.reset()
    updates(1, 2)
    assert np.array_equal(tr.mean(), want["pose2"][0]) and np.array_equal(tr.m2(), want["pose2"][1])
    iv = tr.debug_invariants()
    tr.close()
    assert iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_render_ahead_under_random_call_sequences(seed, monkeypatch):
    """The state machine of render-ahead under calls in any order: enqueued and waited-for batches of random sizes (below and
    above the render-ahead), waits, display updates, changes of the render-ahead itself, resets.  A twin handle without
    render-ahead gets the same waited-for subframes; at every point where something waits the two running means are equal
    bit for bit, and the estimator has never been launched for less than was asked for."""
    rng = np.random.default_rng(seed)
    tex = ds.make_procedural_cloud(64)
    w = h = 96
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    est = seed & 1
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=est)
    twin = ds.CloudTracer(tex, width=w, height=h, estimator=est)
    tr.set_render_ahead(24)
    done = 0          # subframes asked of `tr`
    twin_done = 0

    def check():
        nonlocal twin_done
        if done > twin_done:
            twin.render_accumulate(twin_done + 1, done - twin_done)
            twin_done = done
        assert tr.rendered_subframes() >= done
        assert np.array_equal(tr.mean(), twin.mean()) and np.array_equal(tr.m2(), twin.m2()), done

    for _ in range(70):
        op = rng.integers(0, 10)
        if op <= 4:
            n = int(rng.integers(1, 12))
            tr.render_accumulate_async(done + 1, n)
            done += n
            if rng.integers(0, 2):
                tr.tonemap_async(0.4)
        elif op == 5:
            n = int(rng.integers(20, 60))             # more than the render-ahead: a launch of its own size
            tr.render_accumulate_async(done + 1, n)
            done += n
        elif op == 6:
            n = int(rng.integers(1, 40))
            tr.render_accumulate(done + 1, n)
            done += n
            check()
        elif op == 7:
            tr.synchronize()
            check()
        elif op == 8:
            tr.set_render_ahead(int(rng.choice([0, 8, 24, 40])))
            check()
        else:
            if rng.integers(0, 3) == 0:
                tr.reset()
                twin.reset()
                done = twin_done = 0
            else:
                check()
    tr.synchronize()
    check()
    iv = tr.debug_invariants()
    tr.close()
    twin.close()
    assert iv["armed"] == 1 and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv


@pytest.mark.parametrize("estimator", [0, 1])
def test_checkpoint_and_resume_continue_exactly(estimator, tmp_path):
    """(mean, M2, subframe count) is the whole state of a progressive render -- a sample's seed is (pixel, subframe id) -- so a
    fresh handle that is given the three (ct_upload, ct_set_subframes) continues bit for bit where the saved one was.  The
    reference's EXR dumps (Camera.cpp:211-214) hold neither variance nor count."""
    tex = ds.make_procedural_cloud(48)
    w, h = 72, 56
    a = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    a.render_accumulate(1, 23)
    a.save_state(tmp_path / "state.npz")
    a.render_accumulate(24, 17)
    want = (a.mean(), a.m2(), a.tonemap(0.4)[0])
    a.close()
    b = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    b.render_accumulate(1, 5)                      # (whatever the handle held before is replaced)
    assert b.load_state(tmp_path / "state.npz") == 23
    b.render_accumulate_async(24, 10)
    b.render_accumulate(34, 7)
    assert np.array_equal(b.mean(), want[0]) and np.array_equal(b.m2(), want[1]) and np.array_equal(b.tonemap(0.4)[0], want[2])
    with pytest.raises(_lib.CloudTraceError):
        b.upload(_lib.CT_BUF_SCREEN, np.zeros((h, w, 4), np.float32))
    b.close()


def test_checkpoint_of_a_frozen_image_and_resume_into_a_frozen_handle(tmp_path):
    """With ct_set_stop_when_converged the running mean freezes at the reference's stopping count N while the host keeps
    submitting: a checkpoint taken then must pair the buffers with N, not with the host's count, and loading a checkpoint into
    a handle whose image has frozen must release it -- or every later sample would be dropped silently.  Compared with a
    handle that never had a stopping rule and renders straight through."""
    tex = ds.make_procedural_cloud(64)
    size, mode = 1024, 2                      # (the "crossing" case of the test below: fails at 10 subframes, passes at 20)
    a = ds.CloudTracer(tex, width=size, height=size, mode=mode)
    a.set_stop_when_converged(10, 10)
    for k in range(5):
        a.render_accumulate_async(10 * k + 1, 10)
    a.synchronize()
    frozen_at = a.converged_at()[0]
    assert frozen_at == 20 and a.subframes == 50
    a.save_state(tmp_path / "frozen")          # (no suffix: np.savez adds one, load_state must find the same file)
    assert (tmp_path / "frozen.npz").exists()
    straight = ds.CloudTracer(tex, width=size, height=size, mode=mode)
    straight.render_accumulate(1, 20)
    assert np.array_equal(a.mean(), straight.mean()) and np.array_equal(a.m2(), straight.m2())
    straight.render_accumulate(21, 12)
    want = (straight.mean(), straight.m2())
    straight.close()
    # a fresh handle continues from the frozen count ...
    b = ds.CloudTracer(tex, width=size, height=size, mode=mode)
    assert b.load_state(tmp_path / "frozen") == 20
    b.render_accumulate(21, 12)
    assert np.array_equal(b.mean(), want[0]) and np.array_equal(b.m2(), want[1])
    b.close()
    # ... and so does the frozen handle itself once the checkpoint is loaded into it (the stopping rule switched off first:
    # left on, the image would pass the test again and stop again, which is the rule working, not samples getting lost)
    a.set_stop_when_converged(0, 10)
    assert a.load_state(tmp_path / "frozen.npz") == 20 and a.converged_at()[0] == 0
    a.render_accumulate(21, 12)
    assert np.array_equal(a.mean(), want[0]) and np.array_equal(a.m2(), want[1])
    # with the rule left on and a checkpoint loaded: frozen flag cleared by the load, samples accumulate until the next test
    a.set_stop_when_converged(10, 10)
    a.load_state(tmp_path / "frozen")
    a.render_accumulate(21, 5)                 # (no test falls inside: 25 is not a multiple of 10)
    assert a.converged_at()[0] == 0 and not np.array_equal(a.mean(), b_mean_at_20(tmp_path))
    a.close()


def b_mean_at_20(tmp_path):
    with np.load(tmp_path / "frozen.npz") as z:
        return z["mean"]


@pytest.mark.parametrize("estimator", [0, 1])
def test_a_handle_that_follows_a_moving_camera_equals_fresh_handles(estimator, monkeypatch):
    """Camera::rotate -> reset (Camera.cpp:93-98): one handle renders pose after pose.  What it carries from pose to pose -- the
    scratch, job lists, the tiles in which the last measured pose had its deepest paths (where the next pose's cost-measuring
    launch starts) -- changes schedules only: every pose's image equals the one a fresh handle renders, bit for bit, with the
    invariants armed.  Poses near each other (a drag), a jump across the cloud, and a pose that looks past it."""
    import math
    tex = ds.make_procedural_cloud(64)
    w, h = 120, 88
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    eyes = [(2.5 * math.cos(0.07 * k), -0.4 + 0.05 * k, 2.5 * math.sin(0.07 * k)) for k in range(4)]
    eyes += [(-2.2, 0.8, -1.1), (0.4, 2.6, 0.3), (2.5, -0.4, 0.0)]
    for i, eye in enumerate(eyes):
        look = (0, 0, 0) if i != 5 else (3.0, 2.6, 0.3)      # (the sixth pose looks past the box: no pixel hits it)
        U, V, W = ds.calculate_camera_variables(eye, look, (0, 1, 0), 30.0, w / h)
        tr.set_camera(eye, U, V, W)
        tr.reset()
        tr.render_accumulate_async(1, 10)
        tr.tonemap_async(0.4)
        tr.render_accumulate_async(11, 10)
        tr.render_accumulate(21, 14)
        fresh = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
        fresh.set_camera(eye, U, V, W)
        fresh.render_accumulate(1, 34)
        assert np.array_equal(tr.mean(), fresh.mean()) and np.array_equal(tr.m2(), fresh.m2()), i
        assert tr.counters() == fresh.counters(), i
        fresh.close()
    iv = tr.debug_invariants()
    tr.close()
    assert iv["armed"] == 1 and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv


def _pixels_outside_the_interval(mean, m2, n):
    """Camera::isConverged's count (Camera.cpp:244-262) in float32, for any subframe count."""
    N = np.float32(n)
    sigma = np.sqrt(m2[..., 0] / N)
    a = np.float32(1.96) * sigma / np.sqrt(N)
    r = a / (mean[..., 0] + np.float32(1.1920929e-07))
    return int((~((r < np.float32(0.02)) | (a < np.float32(1e-2)))).sum())


@pytest.mark.parametrize("case", ["crossing", "first_test", "never", "crossing_render_ahead", "first_test_render_ahead", "chunked"])
def test_stop_when_converged_freezes_the_reference_image(case, monkeypatch):
    """Camera::render tests isConverged() before every update of 10 subframes and stops at the first count that passes
    (Camera.cpp:179, 232-268).  ct_set_stop_when_converged takes that decision on the device behind every 10th subframe's
    accumulate kernel, so a host that only enqueues ends with the reference loop's image and count: compared with a twin that
    runs the reference's control flow with waited-for calls -- a count that fails and the next one that passes (single
    scatter, tests from 10 subframes on), the reference's first test at 100 passing, a frame that never converges, the same
    with render-ahead (accumulates cut at the multiples of 10), and a batch rendered in chunks (tested at its end)."""
    ahead = 40 if case.endswith("render_ahead") else 0
    kind = case.replace("_render_ahead", "")
    if kind == "crossing":
        vol, size, mode, first_test, updates = 64, 1024, 2, 10, 6
    elif kind == "first_test":
        vol, size, mode, first_test, updates = 64, 48, 0, 100, 14
    elif kind == "never":
        vol, size, mode, first_test, updates = 64, 96, 0, 100, 16
    else:
        vol, size, mode, first_test, updates = 64, 160, 2, 10, 5
        monkeypatch.setenv("CT_SCRATCH_MIB", "1")
    tex = ds.make_procedural_cloud(vol)
    twin = ds.CloudTracer(tex, width=size, height=size, mode=mode)
    n, stopped, tally = 0, 0, {}
    while n < updates * 10:
        if n >= first_test:
            tally[n] = _pixels_outside_the_interval(twin.mean(), twin.m2(), n)
            if n >= 100:
                assert twin.is_converged() == (tally[n] < 500, tally[n])
            if tally[n] < 500:
                stopped = n
                break
        twin.render_accumulate(n + 1, 10)
        n += 10
    want = (twin.mean(), twin.m2())
    twin.close()
    if kind == "crossing":
        assert stopped == 20 and tally[10] >= 500, tally          # (the case is what its name says)
    elif kind == "first_test":
        assert stopped == 100, tally
    elif kind == "never":
        assert stopped == 0 and min(tally.values()) >= 500, tally

    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    tr = ds.CloudTracer(tex, width=size, height=size, mode=mode)
    tr.set_stop_when_converged(10, first_test)
    if ahead:
        tr.set_render_ahead(ahead)
    for k in range(updates):
        tr.render_accumulate_async(10 * k + 1, 10)
        tr.tonemap_async(0.4)
        tr.converged_at()                                          # (never waits; whatever it says is allowed to be stale)
    tr.synchronize()
    frozen_at, tested_at, outside = tr.converged_at()
    got = (tr.mean(), tr.m2())
    iv = tr.debug_invariants()
    if kind == "chunked":
        # chunk by chunk a batch is a whole image only at its end: every update's end is tested here, so the outcome is the same
        assert stopped == frozen_at
    assert frozen_at == stopped, (frozen_at, tested_at, outside, tally)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    if stopped:
        assert tested_at == stopped and outside == tally[stopped]
        if stopped >= 100:
            assert tr.is_converged() == (True, tally[stopped])    # (tests the frozen image with its own count)
    else:
        assert tested_at == updates * 10 and outside == _pixels_outside_the_interval(got[0], got[1], tested_at)
    assert iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
    # a reset starts over
    tr.reset()
    assert tr.converged_at() == (0, 0, 0)
    tr.render_accumulate(1, 10)
    assert tr.mean().any()
    tr.close()


@pytest.mark.parametrize("estimator", [0, 1])
def test_batches_cut_by_pixel_groups_are_bit_exact(estimator, monkeypatch):
    """A batch whose samples do not fit one scratch region is rendered chunk by chunk of PIXEL GROUPS -- every launch all the
    subframes of a piece of the cost-sorted group order, its results in columns of its own, accumulated by its own kernel --
    with paths and job remainders passing from chunk to chunk.  1 MiB regions on a 160x160 frame: seven launches per
    batch.  Waited-for and enqueued calls, invariants armed, against the one-launch frame bit for bit."""
    tex = ds.make_procedural_cloud(96)
    w = h = 160
    ref = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    ref.render_accumulate(1, 70)
    want = (ref.mean(), ref.m2(), ref.counters())
    ref.close()
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    monkeypatch.setenv("CT_SCRATCH_MIB", "1")
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    tr.render_accumulate(1, 16)              # the cost-measuring launch, then ...
    tr.render_accumulate(17, 16)             # ... a waited-for batch in chunks
    tr.render_accumulate_async(33, 16)       # enqueued batches in chunks
    tr.render_accumulate_async(49, 16)
    tr.render_accumulate_async(65, 6)        # (another batch size: other chunks)
    got = (tr.mean(), tr.m2(), tr.counters())
    iv = tr.debug_invariants()
    tr.close()
    assert iv["armed"] == 1 and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
    assert iv["resumed"] == iv["suspended"] > 0
    assert got[2] == want[2]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])

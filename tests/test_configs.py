"""BASELINE.json's configurations at their stated sizes (run with -m gpu on an MI355X).

configs[2] (512^3 / 1024^2 / 1024 spp) lives in test_gpu_parity.py (test_full_size_properties_512_1024,
test_north_star_parity_after_1024_spp_on_a_window); configs[3] is the same job on 8 GPUs (its one-GPU parts:
test_merged_shards_tonemap_and_convergence_equal_the_single_gpu_job, test_pixel_tile_shards_sum_to_the_whole,
tests/test_distributed.py).  Here: configs[0], configs[1] and configs[4].  Everything goes through the C ABI and is
compared with the CPU oracle bit for bit -- the whole frame where the oracle can afford it, a window otherwise.
"""
import numpy as np
import pytest

import _oracle as O
import deepestscatter_amd as ds
from deepestscatter_amd import _lib

pytestmark = pytest.mark.gpu


def test_config0_128_perlin_256x256_16spp_single_scatter_whole_frame():
    """configs[0]: 128^3 procedural Perlin density, 256x256, 16 spp, single-scatter only
    (singleScatterSunRadiance, cloudRadianceMaterials.cu:120-148).  The one configuration whose FULL frame the
    oracle renders in seconds: every pixel's mean and M2, every counter, the shadow volume and the tonemapped
    screen must be identical."""
    tex = ds.make_procedural_cloud(128)
    w = h = 256
    tr = ds.CloudTracer(tex, width=w, height=h, mode=2)
    tr.render_accumulate(1, 10)            # the reference's cadence: 10 subframes per Camera::render (Camera.cpp:189)
    tr.render_accumulate_async(11, 6)
    mean, m2 = tr.mean(), tr.m2()
    orc = O.Oracle(tex, w, h, mode=2, fast=True)
    ref_mean, ref_m2 = orc.render(16)
    assert np.array_equal(tr.inscatter(), orc.inscatter)
    assert np.array_equal(mean, ref_mean) and np.array_equal(m2, ref_m2)
    assert tr.counters() == orc.counters.as_dict()
    assert tr.counters()["paths"] == w * h * 16 and (mean[..., 0] > 0).mean() > 0.05     # a cloud is in the picture
    screen, avg = tr.tonemap(0.4)
    ref_screen, ref_avg = O.reinhard(ref_mean, 0.4)
    assert np.array_equal(screen, ref_screen) and avg == ref_avg
    # the same job with the north star's estimator (its own oracle twin)
    td = ds.CloudTracer(tex, width=w, height=h, mode=2, estimator=1)
    td.render_accumulate(1, 16)
    od = O.Oracle(tex, w, h, mode=2, fast=True, estimator=1, inscatter=orc.inscatter)
    dm, dm2 = od.render(16)
    assert np.array_equal(td.mean(), dm) and np.array_equal(td.m2(), dm2) and td.counters() == od.counters.as_dict()
    td.close()
    tr.close()


@pytest.mark.parametrize("estimator", [0, 1])
def test_config1_256_512x512_256spp_on_a_window(estimator):
    """configs[1]: 256^3 cloud (the procedural one stands in for the Houdini .vdb: none ships with the reference),
    512x512, 256 spp, Mie multi-scatter + NEE.  The GPU renders the whole job as enqueued batches; the oracle renders
    a 32x32 window in the body of the cloud for all 256 subframes: mean and M2 bit-identical there, and the north
    star's relative-L2 criterion (<= 1e-3) holds trivially."""
    tex = ds.make_procedural_cloud(256)
    w = h = 512
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    first = 1
    for n in (100, 100, 56):
        tr.render_accumulate_async(first, n)
        first += n
    tr.synchronize()
    mean, m2, c = tr.mean(), tr.m2(), tr.counters()
    assert c["paths"] == w * h * 256 and np.isfinite(mean).all()
    orc = O.Oracle(tex, w, h, fast=True, estimator=estimator, inscatter=tr.inscatter())
    x0, y0 = 246, 262
    win = (x0, y0, x0 + 32, y0 + 32)
    ref_mean, ref_m2 = orc.render(256, window=win)
    got, ref = mean[y0:y0 + 32, x0:x0 + 32], ref_mean[y0:y0 + 32, x0:x0 + 32]
    assert ref[..., 0].mean() > 0.3                       # the window is inside the cloud
    err = float(np.linalg.norm(got.astype(np.float64) - ref) / np.linalg.norm(ref.astype(np.float64)))
    assert err <= 1e-3
    assert np.array_equal(got, ref)
    assert np.array_equal(m2[y0:y0 + 32, x0:x0 + 32], ref_m2[y0:y0 + 32, x0:x0 + 32])
    tr.close()


@pytest.fixture(scope="module")
def cloud_1024():
    return ds.make_procedural_cloud(1024)


def test_config4_shadow_volume_against_the_oracle(cloud_1024):
    """The 1024^3 shadow volume (where inscatter_kernel's clearance cap of 127 texels is reached and a march step
    advances two texels) against inScatter.cu:40-66 restated texel by texel, on >= 1e5 texels of every class."""
    from test_parity_gaps import check_shadow_volume
    tr = ds.CloudTracer(cloud_1024, width=64, height=64)
    got = tr.inscatter()
    tr.close()
    assert check_shadow_volume(cloud_1024, got, seed=1024) >= 100_000


@pytest.mark.parametrize("estimator,sparse", [(0, False), (1, False), (0, True)])
def test_config4_1024_2048x2048_window_and_no_lost_jobs(cloud_1024, estimator, sparse, monkeypatch):
    """configs[4] on one GPU: 1024^3 density (dense march bricks, and the sparse brick-compressed storage of
    CT_FLAG_SPARSE_BRICKS; per-XCD job queues on by default at this size, 32-texel majorant cells for DELTA), 2048x2048.  Enqueued batches with the invariants armed (NaN-filled scratch,
    path conservation: the configuration in which round 1's DELTA kernel dropped the jobs of seven of its eight
    queues) against synchronous batches -- mean, M2 and every counter of the WHOLE frame -- and against the oracle
    on a 12x12 window."""
    tex = cloud_1024
    w = h = 2048
    spp = 12
    ref = ds.CloudTracer(tex, width=w, height=h, estimator=estimator)
    ref.render_accumulate(1, spp)
    want = (ref.mean(), ref.m2(), ref.counters())
    ins = ref.inscatter()
    assert ref.debug_memory()["sparse"] == 0
    ref.close()
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=estimator, flags=_lib.CT_FLAG_SPARSE_BRICKS if sparse else 0)
    mem = tr.debug_memory()
    assert mem["sparse"] == (1 if sparse else 0)
    if sparse:
        assert mem["march_bricks_stored"] < 0.15 * mem["march_bricks_dense"]
    tr.render_accumulate_async(1, 7)
    tr.render_accumulate_async(8, spp - 7)
    got = (tr.mean(), tr.m2(), tr.counters())
    iv = tr.debug_invariants()
    tr.close()
    assert iv["armed"] == 1 and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
    assert iv["dealt"] == iv["written"] == want[2]["box_hits"] and iv["resumed"] == iv["suspended"] > 0, iv
    assert got[2] == want[2]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    orc = O.Oracle(tex, w, h, fast=True, estimator=estimator, inscatter=ins)
    if estimator == 1:
        assert orc.scene.maj_cell == 20                        # (round 4: the grid is cropped to the cloud; 32-texel cells before)
    x0, y0 = 1000, 1040
    win = (x0, y0, x0 + 12, y0 + 12)
    rm, rm2 = orc.render(spp, window=win)
    assert rm[y0:y0 + 12, x0:x0 + 12, 0].mean() > 0.2
    assert np.array_equal(got[0][y0:y0 + 12, x0:x0 + 12], rm[y0:y0 + 12, x0:x0 + 12])
    assert np.array_equal(got[1][y0:y0 + 12, x0:x0 + 12], rm2[y0:y0 + 12, x0:x0 + 12])

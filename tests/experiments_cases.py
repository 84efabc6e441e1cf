"""Parity cases of the measured-and-rejected experiments: the DELTA kernels that regroup paths by phase
(deepestscatter_amd/csrc/ct_exchange.hpp) and the march bricks behind HIP virtual memory (CT_FLAG_VMM_BRICKS).  Not collected
by the test run itself: tests/test_exchange.py runs this file in a child process on the EXPERIMENTS build of the library
(CT_LIBRARY=libcloudtrace_exp.so), the only build that holds them.

CT_EXCHANGE=1: block-wide exchange of paths between the 16 waves of a workgroup (slots + three rings in LDS);
CT_EXCHANGE=2: the same regrouping within a wave (private lists, no atomics).  Neither is the default -- both measured
slower than render_delta_kernel (profiles/r03c, r03d) -- but both must produce its results bit for bit:
a path's arithmetic does not depend on the lane, the wave or the order it runs in.
"""
import numpy as np
import pytest

import _oracle as O
import deepestscatter_amd as ds
from conftest import sphere_volume

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["1", "2"])
@pytest.mark.parametrize("n,size,mode", [(40, 56, 0), (48, 64, 1), (40, 64, 2)])
def test_exchange_kernels_equal_the_oracle_bit_for_bit(variant, n, size, mode, monkeypatch):
    monkeypatch.setenv("CT_EXCHANGE", variant)
    monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")      # NaN-filled scratch + samples dealt == results written
    tex = ds.make_procedural_cloud(n)
    tr = ds.CloudTracer(tex, width=size, height=size, mode=mode, estimator=1)
    tr.render_accumulate(1, 8)          # the cost-measuring launch of a pose keeps the per-lane kernel
    tr.render_accumulate(9, 8)          # exchange kernel
    tr.render_accumulate(17, 48)        # exchange kernel, long enough for the lists / rings to wrap
    spp = 64
    mean, m2, c = tr.mean(), tr.m2(), tr.counters()
    st, iv = tr.debug_stats(), tr.debug_invariants()
    orc = O.Oracle(tex, size, size, mode=mode, fast=True, estimator=1, inscatter=tr.inscatter())
    rm, rm2 = orc.render(spp)
    tr.close()
    assert st["watchdog"] == 0                          # no wave gave up on a bounded wait
    assert iv["armed"] == 1 and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0, iv
    assert c == orc.counters.as_dict()
    assert np.array_equal(mean, rm) and np.array_equal(m2, rm2)


@pytest.mark.parametrize("variant", ["1", "2"])
def test_exchange_kernels_on_the_benchmark_scene_equal_the_per_lane_kernel(variant, monkeypatch):
    """512^3 / 1024^2 (16-texel majorant cells, every block busy): 24 subframes, whole frame, against render_delta_kernel."""
    tex = ds.make_procedural_cloud(512)
    ref = ds.CloudTracer(tex, width=1024, height=1024, estimator=1)
    ref.render_accumulate(1, 24)
    want = (ref.mean(), ref.m2(), ref.counters())
    ref.close()
    monkeypatch.setenv("CT_EXCHANGE", variant)
    tr = ds.CloudTracer(tex, width=1024, height=1024, estimator=1)
    tr.render_accumulate(1, 8)
    tr.render_accumulate(9, 16)
    got = (tr.mean(), tr.m2(), tr.counters())
    st = tr.debug_stats()
    tr.close()
    assert st["watchdog"] == 0 and got[2] == want[2]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_vmm_backed_march_bricks_are_bit_exact(monkeypatch):
    """CT_FLAG_VMM_BRICKS / CT_SPARSE=2: the dense march-brick array as a reserved virtual range whose empty 2-MiB chunks share
    memory after their clearances have been rounded down to {0, 4, ..., 127} texels (brick rows padded to a power of two so
    that chunks hold whole rows).  The kernel is the dense one; only the exact free-space skip gets shorter where a clearance
    was rounded: radiance, M2 and the algorithm's counters equal the dense handle's and the oracle's, from several directions,
    on a small cloud in a big box (many empty chunks), a volume without a zero border and one smaller than a chunk."""
    rng = np.random.default_rng(99)
    # a long box (brick rows of 512 bricks after padding: a 2-MiB chunk holds 32 of them) with a blob in one corner: the rows
    # far from it and from the faces are equal after quantisation, and so are some whole chunks
    big = np.zeros((160, 320, 1400), np.uint8)
    big[8:40, 20:60, 30:90] = rng.integers(1, 256, (32, 40, 60)).astype(np.uint8)
    big[100:104, 250:254, 1200:1204] = 255
    cases = [(big, 7000.0, 2), (rng.integers(0, 256, (40, 48, 36)).astype(np.uint8), 80.0, 2), (sphere_volume(24, seed=3), 3000.0, 3)]
    for i, (tex, size, spp) in enumerate(cases):
        w, h = 48, 40
        monkeypatch.setenv("CT_SPARSE", "0")
        dense = ds.CloudTracer(tex, width=w, height=h, cloud_size_m=size)
        monkeypatch.setenv("CT_SPARSE", "2")
        monkeypatch.setenv("CT_DEBUG_INVARIANTS", "1")
        tr = ds.CloudTracer(tex, width=w, height=h, cloud_size_m=size)
        monkeypatch.delenv("CT_DEBUG_INVARIANTS")
        mem = tr.debug_memory()
        assert mem["sparse"] == 2 and dense.debug_memory()["sparse"] == 0
        if i == 0:
            assert mem["march_bricks_stored"] < mem["march_bricks_dense"]        # some empty chunks share memory
        assert np.array_equal(tr.inscatter(), dense.inscatter())
        for t in (dense, tr):
            t.render_accumulate(1, spp)
            t.render_accumulate_async(spp + 1, 2)
        assert np.array_equal(tr.mean(), dense.mean()) and np.array_equal(tr.m2(), dense.m2()) and tr.counters() == dense.counters(), i
        if i != 0:
            orc = O.Oracle(tex, w, h, fast=True, cloud_size_m=size, inscatter=dense.inscatter())
            mean, m2 = orc.render(spp + 2)
            assert np.array_equal(tr.mean(), mean) and np.array_equal(tr.m2(), m2) and tr.counters() == orc.counters.as_dict(), i
        for eye in ((0.3, 2.2, 0.4), (-2.3, 0.3, 0.2), (0.1, 0.05, -0.2)):
            U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
            for t in (dense, tr):
                t.set_camera(eye, U, V, W)
                t.render_subframe(7)
            assert np.array_equal(tr.frame(), dense.frame()), (i, eye)
        assert tr.debug_invariants()["violations"] == 0
        tr.close()
        dense.close()

"""The experimental DELTA kernels that regroup paths by phase (deepestscatter_amd/csrc/ct_exchange.hpp).  Not collected by the
test run itself: tests/test_exchange.py runs this file in a child process on the EXPERIMENTS build of the library
(CT_LIBRARY=libcloudtrace_exp.so), the only build that holds these kernels.

CT_EXCHANGE=1: block-wide exchange of paths between the 16 waves of a workgroup (slots + three rings in LDS);
CT_EXCHANGE=2: the same regrouping within a wave (private lists, no atomics).  Neither is the default -- both measured
slower than render_delta_kernel (profiles/r03c, r03d) -- but both must produce its results bit for bit:
a path's arithmetic does not depend on the lane, the wave or the order it runs in.
"""
import numpy as np
import pytest

import _oracle as O
import deepestscatter_amd as ds

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

#!/usr/bin/env python3
"""Soak run of the randomized differential test's generator: many random scenes and batch patterns, HIP path
vs oracle, bit for bit.  python tools/soak.py <seed> <cases>
(SOAK_PAD=<texels>: every volume gets a zero border of that width; SOAK_EST=<0|1>: one estimator for every case)"""
import os
import sys
from pathlib import Path
os.environ.setdefault("CT_DEBUG_INVARIANTS", "1")   # NaN-filled scratch + path conservation, checked by the library itself
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401  (first: see tests/conftest.py)
import numpy as np
import deepestscatter_amd as ds
from test_gpu_parity import _random_scene, make_pair

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
interior = delta_cases = 0   # DELTA cases, and those that ran the kernel without the box test and the clamp
for case in range(cases):
    kw, eye = _random_scene(rng)
    tex = kw.pop("tex"); w, h = kw.pop("width"), kw.pop("height")
    if os.environ.get("SOAK_PAD"):      # a zero border of that many texels: the cloud inside its volume (the DELTA estimator's interior kernel)
        tex = np.pad(tex, int(os.environ["SOAK_PAD"]))
    if os.environ.get("SOAK_EST"):
        kw["estimator"] = int(os.environ["SOAK_EST"])
    # per-XCD job queues in a third of the cases (read at ct_create)
    os.environ["CT_XCD_QUEUES"] = "1" if rng.random() < 0.33 else "0"
    tr, orc = make_pair(tex, w, h, **kw)
    if kw.get("estimator", 0) == 1:
        delta_cases += 1
        interior += 1 if tr.delta_grid()["interior"] else 0
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    tr.set_camera(eye, U, V, W); orc.set_camera(eye, U, V, W)
    first = 1
    pattern = [(int(n), bool(rng.random() < 0.6)) for n in rng.integers(1, 5, 4)]
    try:
        for n, a in pattern:
            (tr.render_accumulate_async if a else tr.render_accumulate)(first, n)
            first += n
        got_mean, got_m2, got_counters = tr.mean(), tr.m2(), tr.counters()
    except ds.CloudTraceError as e:
        bad += 1
        print(f"INVARIANT seed {seed} case {case}: {e}; dims {tex.shape[::-1]} {w}x{h} {kw} eye {eye} batches {pattern} "
              f"xcd {os.environ['CT_XCD_QUEUES']}", flush=True)
        tr.close()
        continue
    mean, m2 = orc.render(sum(n for n, _ in pattern))
    iv = tr.debug_invariants()
    assert iv["armed"] == 1 and iv["checks"] > 0, iv
    ok = np.array_equal(got_mean, mean) and np.array_equal(got_m2, m2) and got_counters == orc.counters.as_dict()
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed} case {case}: dims {tex.shape[::-1]} {w}x{h} {kw} eye {eye} batches {pattern}", flush=True)
        # what differs, and whether a second render of the same handle agrees with the first (a schedule-dependent result)
        dm, dv = got_mean != mean, got_m2 != m2
        print(f"   mean differs at {int(dm.any(axis=-1).sum())} pixels (max abs {float(np.abs(got_mean - mean).max()):.3e}), "
              f"m2 at {int(dv.any(axis=-1).sum())}; first pixels {np.argwhere(dm.any(axis=-1))[:4].tolist()}; counters hip {got_counters} "
              f"oracle {orc.counters.as_dict()}", flush=True)
        tr.reset()
        first2 = 1
        for n, a in pattern:
            (tr.render_accumulate_async if a else tr.render_accumulate)(first2, n)
            first2 += n
        print(f"   same handle, same batches again: mean equal to oracle {np.array_equal(tr.mean(), mean)}, to first run "
              f"{np.array_equal(tr.mean(), got_mean)}", flush=True)
    tr.close()
    if case % 50 == 49:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"done: {cases} cases, {bad} mismatches ({delta_cases} with the DELTA estimator, {interior} of them on its interior kernel)")

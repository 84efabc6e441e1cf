#!/usr/bin/env python3
"""Soak run for path continuation: random scenes at frame sizes where thousands of paths cross launch
boundaries; enqueued batches vs synchronous batches on the HIP path (mean, M2, counters bit for bit).
python tools/soak_continuation.py <seed> <cases>"""
import os
import sys
from pathlib import Path
os.environ.setdefault("CT_DEBUG_INVARIANTS", "1")   # NaN-filled scratch + path conservation, checked by the library itself
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
import numpy as np
import deepestscatter_amd as ds
from test_gpu_parity import _random_scene

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
total_suspended = 0
for case in range(cases):
    kw, eye = _random_scene(rng)
    tex = kw.pop("tex"); kw.pop("width"); kw.pop("height")
    w, h = int(rng.integers(96, 320)), int(rng.integers(64, 256))
    kw["estimator"] = int(rng.random() < 0.4)
    kw["cloud_size_m"] = float(rng.choice([7000.0, 20000.0, 40000.0]))
    kw["sample_step"] = 1.0 / 512
    kw["max_depth"] = int(rng.choice([50, 300, 2000]))
    os.environ["CT_XCD_QUEUES"] = "1" if rng.random() < 0.4 else "0"   # the enqueued handle: per-XCD queues in 40 % of the cases
    a = ds.CloudTracer(tex, width=w, height=h, **kw)
    os.environ["CT_XCD_QUEUES"] = "0"
    b = ds.CloudTracer(tex, width=w, height=h, **kw)
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    a.set_camera(eye, U, V, W); b.set_camera(eye, U, V, W)
    first = 1
    pattern = [int(n) for n in rng.integers(1, 9, int(rng.integers(2, 7)))]
    syncs = [bool(rng.random() < 0.2) for _ in pattern]
    try:
        for n, sy in zip(pattern, syncs):
            a.render_accumulate_async(first, n)
            b.render_accumulate(first, n)
            first += n
            if sy:
                a.synchronize()
        ok = np.array_equal(a.mean(), b.mean()) and np.array_equal(a.m2(), b.m2()) and a.counters() == b.counters()
        iv = a.debug_invariants()
        ok = ok and iv["violations"] == 0 and iv["samples_without_alpha_1"] == 0
    except ds.CloudTraceError as e:
        print(f"INVARIANT {e}", flush=True)
        ok = False
    total_suspended += a.debug_suspended()
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed} case {case}: dims {tex.shape[::-1]} {w}x{h} {kw} eye {eye} batches {pattern}", flush=True)
    a.close(); b.close()
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} mismatches, {total_suspended} paths suspended so far", flush=True)
print(f"done: {cases} cases, {bad} mismatches, {total_suspended} suspended paths")

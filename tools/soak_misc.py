#!/usr/bin/env python3
"""Soak run for the entry points beside the renderer: descriptors, scatter-sample generator, point-radiance
tasks (HIP path vs oracle) and pixel-tile shards with enqueued batches (sum of shards vs whole), on random
scenes.  python tools/soak_misc.py <seed> <cases>"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
import numpy as np
import deepestscatter_amd as ds
from deepestscatter_amd.cloudtrace import make_point_tasks
from test_gpu_parity import _random_scene, make_pair

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    kw, eye = _random_scene(rng)
    tex = kw.pop("tex"); w, h = kw.pop("width"), kw.pop("height")
    problems = []
    tr, orc = make_pair(tex, w, h, **kw)
    # descriptors at random points / directions
    pos = rng.uniform(-0.7, 0.7, (6, 3)).astype(np.float32)
    view = rng.normal(size=(6, 3)).astype(np.float32)
    if not np.array_equal(tr.collect_descriptors(pos, view), orc.collect_descriptors(pos, view)):
        problems.append("descriptors")
    if kw["estimator"] == 0 and tex.any():
        gp, gd = tr.generate_scatter_samples(10, batch_seed=case)
        op, od = orc.generate_scatter_samples(10, batch_seed=case)
        if not (np.array_equal(gp, op, equal_nan=True) and np.array_equal(gd, od, equal_nan=True)):
            problems.append("scatter samples")
        ok = ~np.isnan(gp[:, 0])
        if ok.any():
            t1 = make_point_tasks(gp[ok], gd[ok]); t2 = t1.copy()
            tr.point_radiance_launch(t1, 1, 3); orc.point_radiance_launch(t2, 1, 3)
            if t1.tobytes() != t2.tobytes():
                problems.append("point tasks")
    tr.close()
    # shards with enqueued batches
    count = int(rng.integers(2, 5))
    okw = {k: v for k, v in kw.items()}
    whole = ds.CloudTracer(tex, width=w, height=h, **okw)
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    whole.set_camera(eye, U, V, W)
    whole.render_accumulate(1, 5)
    merged = np.zeros_like(whole.mean())
    for i in range(count):
        sh = ds.CloudTracer(tex, width=w, height=h, shard_index=i, shard_count=count, **okw)
        sh.set_camera(eye, U, V, W)
        sh.render_accumulate_async(1, 2); sh.render_accumulate_async(3, 3)
        merged += sh.mean()
        sh.close()
    if not np.array_equal(merged, whole.mean()):
        problems.append("shards")
    whole.close()
    if problems:
        bad += 1
        print(f"MISMATCH seed {seed} case {case} {problems}: dims {tex.shape[::-1]} {w}x{h} {kw} eye {eye}", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} with mismatches", flush=True)
print(f"done: {cases} cases, {bad} with mismatches")

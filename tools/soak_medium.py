#!/usr/bin/env python3
"""Soak run at medium sizes: non-cubic procedural-like volumes of 48..160 texels per axis (both estimators; at these
sizes DELTA's majorant cells are 4 or 8 texels wide), frames of
200..640 pixels, a random oracle window; HIP path (enqueued batches) vs oracle bit for bit on the window,
plus counter identities on the whole frame.  python tools/soak_medium.py <seed> <cases>"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
import numpy as np
import deepestscatter_amd as ds
import _oracle as O

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    n = int(rng.choice([64, 96, 128, 160]))
    cube = ds.make_procedural_cloud(n, seed=int(rng.integers(1, 1 << 30)))
    nx, ny, nz = (int(rng.integers(48, n + 1)) for _ in range(3))
    z0, y0, x0 = ((n - nz) // 2, (n - ny) // 2, (n - nx) // 2)
    tex = np.ascontiguousarray(cube[z0:z0 + nz, y0:y0 + ny, x0:x0 + nx])
    w, h = int(rng.integers(200, 641)), int(rng.integers(200, 481))
    kw = dict(mode=int(rng.integers(0, 3)), estimator=int(rng.random() < 0.4), cloud_size_m=float(rng.choice([3000.0, 7000.0, 15000.0])),
              light_direction=tuple(float(v) for v in rng.normal(size=3)), max_depth=int(rng.choice([100, 2000])))
    eye = rng.normal(size=3); eye = tuple(float(v) for v in eye / np.linalg.norm(eye) * rng.uniform(1.2, 3.0))
    tr = ds.CloudTracer(tex, width=w, height=h, **kw)
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    tr.set_camera(eye, U, V, W)
    tr.render_accumulate_async(1, 3); tr.render_accumulate_async(4, 2)
    mean, m2 = tr.mean(), tr.m2()
    orc = O.Oracle(tex, w, h, fast=True, inscatter=tr.inscatter(), **kw)
    orc.set_camera(eye, U, V, W)
    wx, wy = int(rng.integers(0, w - 24)), int(rng.integers(0, h - 24))
    # aim the window at the cloud when possible: brightest 24x24 block of a coarse scan
    lum = mean[..., 0]
    ys, xs = np.unravel_index(np.argmax(lum), lum.shape)
    if rng.random() < 0.7:
        wx, wy = int(np.clip(xs - 12, 0, w - 24)), int(np.clip(ys - 12, 0, h - 24))
    win = (wx, wy, wx + 24, wy + 24)
    rm, rm2 = orc.render(5, window=win)
    c = tr.counters()
    ok = (np.array_equal(mean[wy:wy + 24, wx:wx + 24], rm[wy:wy + 24, wx:wx + 24]) and
          np.array_equal(m2[wy:wy + 24, wx:wx + 24], rm2[wy:wy + 24, wx:wx + 24]) and
          c["paths"] == 5 * w * h and c["scatter_events"] == c["inscatter_lookups"] and np.isfinite(mean).all())
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed} case {case}: dims {(nx, ny, nz)} {w}x{h} {kw} eye {eye} win {win}", flush=True)
    tr.close()
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"done: {cases} cases, {bad} mismatches")

#!/usr/bin/env python3
"""One-off diagnostic: DELTA at the continuation soak's sizes, one mode at a time, every case logged BEFORE it runs."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
import numpy as np
import deepestscatter_amd as ds
from test_gpu_parity import _random_scene

seed, cases, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]   # mode: sync | async
rng = np.random.default_rng(seed)
log = open(sys.argv[4], "w")
only = int(sys.argv[5]) if len(sys.argv) > 5 else -1
for case in range(cases):
    kw, eye = _random_scene(rng)
    tex = kw.pop("tex"); kw.pop("width"); kw.pop("height")
    w, h = int(rng.integers(96, 320)), int(rng.integers(64, 256))
    est = int(rng.random() < 0.4)
    kw["estimator"] = est
    kw["cloud_size_m"] = float(rng.choice([7000.0, 20000.0, 40000.0]))
    kw["sample_step"] = 1.0 / 512
    kw["max_depth"] = int(rng.choice([50, 300, 2000]))
    pattern = [int(n) for n in rng.integers(1, 9, int(rng.integers(2, 7)))]
    syncs = [bool(rng.random() < 0.2) for _ in pattern]
    if est != 1 or (only >= 0 and case != only):
        continue
    print(f"case {case}: dims {tex.shape[::-1]} {w}x{h} {kw} eye {eye} batches {pattern}", file=log, flush=True)
    t = ds.CloudTracer(tex, width=w, height=h, **kw)
    U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    t.set_camera(eye, U, V, W)
    first = 1
    for n, sy in zip(pattern, syncs):
        if mode == "async":
            t.render_accumulate_async(first, n)
            if sy:
                t.synchronize()
        else:
            t.render_accumulate(first, n)
        first += n
    m = t.mean()
    print(f"   ok finite={bool(np.isfinite(m).all())} suspended={t.debug_suspended()}", file=log, flush=True)
    t.close()
print("diag done", file=log, flush=True)

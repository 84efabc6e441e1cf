#!/usr/bin/env python3
"""Soak run for the shadow volume (inscatter_kernel vs the oracle's inScatter restatement): random volumes of random,
non-cubic sizes -- blobs, specks, dense noise without a border, slabs with holes, with and without the reference's zero
border, single non-empty boundary layers -- random sun directions (axis-parallel and in-plane ones among them), steps
and cloud sizes.  Every texel of the shadow volume must be equal.   python tools/soak_inscatter.py <seed> <cases>"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
import numpy as np
from test_gpu_parity import _random_scene, make_pair

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    kw, _ = _random_scene(rng)
    tex = kw.pop("tex"); kw.pop("width"); kw.pop("height")
    r = rng.random()
    if r < 0.15:                                   # sun along an axis, or in a coordinate plane
        light = np.zeros(3); light[int(rng.integers(0, 3))] = rng.choice([-1.0, 1.0])
        if rng.random() < 0.5:
            light[int(rng.integers(0, 3))] += rng.normal()
        kw["light_direction"] = tuple(float(v) for v in light)
    if rng.random() < 0.3:                         # exactly one boundary layer that is not empty
        face = int(rng.integers(0, 6))
        sl = [slice(None)] * 3
        sl[face // 2] = 0 if face % 2 == 0 else -1
        tex[tuple(sl)] = rng.integers(1, 256)
    kw["estimator"] = int(rng.random() < 0.2)
    tr, orc = make_pair(tex, 8, 8, **kw)
    got = tr.inscatter()
    if not np.array_equal(got, orc.inscatter):
        bad += 1
        d = np.argwhere(got != orc.inscatter)
        print(f"MISMATCH seed {seed} case {case}: dims {tex.shape} {kw} texels {len(d)} first {d[0]} got {got[tuple(d[0])]} want {orc.inscatter[tuple(d[0])]}", flush=True)
    tr.close()
    if (case + 1) % 250 == 0:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"done: {cases} cases, {bad} mismatches", flush=True)

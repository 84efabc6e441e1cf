#!/usr/bin/env python3
"""Replays tools/soak.py's generator up to one case and renders it in several ways (for a reported mismatch):
    python tools/repro_case.py <seed> <case> [repeats]
Prints the oracle's counters (twice: the oracle must agree with itself) and, for every variant -- per-XCD queues on/off,
invariants armed or not, the case's own batch pattern / all synchronous / all enqueued -- the HIP counters and whether
mean and M2 equal the oracle's."""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
import numpy as np
import deepestscatter_amd as ds
from test_gpu_parity import _random_scene, make_pair

seed, case = int(sys.argv[1]), int(sys.argv[2])
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rng = np.random.default_rng(seed)
for _ in range(case + 1):
    kw, eye = _random_scene(rng)
    xcd = rng.random() < 0.33
    pattern = [(int(n), bool(rng.random() < 0.6)) for n in rng.integers(1, 5, 4)]
tex = kw.pop("tex"); w, h = kw.pop("width"), kw.pop("height")
print(f"case {case} of seed {seed}: dims {tex.shape[::-1]} {w}x{h} {kw} eye {eye} batches {pattern} xcd {xcd}", flush=True)
U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
total = sum(n for n, _ in pattern)
want = None
for i in range(2):
    os.environ["CT_XCD_QUEUES"] = "0"
    tr, orc = make_pair(tex, w, h, **kw)
    tr.close()
    orc.set_camera(eye, U, V, W)
    mean, m2 = orc.render(total)
    c = orc.counters.as_dict()
    print(f"oracle run {i}: {c}", flush=True)
    assert want is None or (c == want[2] and np.array_equal(mean, want[0])), "the oracle disagrees with itself"
    want = (mean, m2, c)
for armed in ("1", "0"):
    for x in ("1", "0"):
        for name, pat in (("as reported", pattern), ("all synchronous", [(n, False) for n, _ in pattern]),
                          ("all enqueued", [(n, True) for n, _ in pattern])):
            for r in range(repeats):
                os.environ["CT_XCD_QUEUES"] = x
                os.environ["CT_DEBUG_INVARIANTS"] = armed
                tr = ds.CloudTracer(tex, width=w, height=h, **kw)
                tr.set_camera(eye, U, V, W)
                first = 1
                for n, a in pat:
                    (tr.render_accumulate_async if a else tr.render_accumulate)(first, n)
                    first += n
                got = (tr.mean(), tr.m2(), tr.counters())
                ok_img = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
                print(f"armed {armed} xcd {x} {name:16s} run {r}: image {'ok' if ok_img else 'DIFFERS'}, counters "
                      f"{'ok' if got[2] == want[2] else 'DIFFER ' + str({k: got[2][k] - want[2][k] for k in got[2] if got[2][k] != want[2][k]})}", flush=True)
                tr.close()

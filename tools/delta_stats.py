#!/usr/bin/env python3
"""Scheduler statistics of one DELTA launch on the benchmark scene (needs CT_STATS=1 in the environment)."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
assert os.environ.get("CT_STATS"), "run with CT_STATS=1"
import torch  # noqa: F401
import deepestscatter_amd as ds
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
est = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t = ds.CloudTracer(ds.make_procedural_cloud(512), width=1024, height=1024, estimator=est)
t.render_accumulate(1, 32)
s0, k0 = t.debug_stats(), t.counters()
if os.environ.get("CT_STATS_SYNC"):
    t.render_accumulate(33, S)                      # every path runs to its end inside the launch: the tail is counted
else:
    for i in range(3):                              # the way bench.py runs: enqueued launches that hand their paths on
        t.render_accumulate_async(33 + i * S, S)
    t.synchronize()
s1, k1 = t.debug_stats(), t.counters()
paths = k1["paths"] - k0["paths"]
for n in ("regen_phases", "regen_lanes", "march_phases", "march_lanes", "scatter_phases", "scatter_lanes", "fetched_steps",
          "fetched_zero_cells", "skipped_steps"):
    print(f"{n:28s} {(s1[n] - s0[n]) / paths:10.3f} per sample")
if est == 1:
    print(f"{'crossings of empty cells':28s} {(s1['raw'][9] - s0['raw'][9]) / paths:10.3f} per sample   (of skipped_steps)")
for n in ("density_lookups", "inscatter_lookups", "scatter_events"):
    print(f"{n:28s} {(k1[n] - k0[n]) / paths:10.3f} per sample")
import numpy as np
raw = np.zeros(64, np.uint64)
from deepestscatter_amd.cloudtrace import _p, check
check(t.L.ct_debug_stats(t.h, _p(raw)), t.h)
if est == 1:
    ht, hs = raw[16:24].astype(float), raw[24:32].astype(float)
    print("tracking visits by lanes taking part (1-8, 9-16, ..., 57-64), share of visits:", [round(v, 3) for v in ht / max(ht.sum(), 1)])
    print("scatter phases  by lanes taking part,                          share of phases:", [round(v, 3) for v in hs / max(hs.sum(), 1)])
    print("tracking visits made after the wave had found the job list empty:", int(raw[32]), "of", int(ht.sum()))

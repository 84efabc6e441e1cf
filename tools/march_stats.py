#!/usr/bin/env python3
"""What the MARCH kernel's fetches are made of on a benchmark scene (CT_STATS=1 build of the kernel): per sample,
the algorithm's lookups, the fetched steps, how many of those returned an all-zero footprint (and with which
clearance), the shadow-volume fetches and the reused ones.  python tools/march_stats.py <volume> <size> <spp>"""
import os, sys, json
from pathlib import Path
os.environ.setdefault("CT_STATS", "1")
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: F401
import deepestscatter_amd as ds
n, size, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = ds.CloudTracer(ds.make_procedural_cloud(n), width=size, height=size)
t.render_accumulate(1, 32)
s0, k0, f0 = t.debug_stats(), t.counters(), t.fetch_counters()
t.render_accumulate(33, spp)
s1, k1, f1 = t.debug_stats(), t.counters(), t.fetch_counters()
d = lambda a, b, k: b[k] - a[k]
paths = d(k0, k1, "paths")
out = {"volume": n, "size": size, "spp": spp, "memory": t.debug_memory(),
       "per_sample": {"density_lookups": d(k0, k1, "density_lookups") / paths, "fetched_steps": d(s0, s1, "fetched_steps") / paths,
                      "fetched_zero_footprints": d(s0, s1, "fetched_zero_cells") / paths,
                      "zero_footprints_with_clearance_0": d(s0, s1, "zero_cells_nonfree_brick") / paths,
                      "zero_footprints_with_clearance_1": d(s0, s1, "zero_cells_free_brick_d1") / paths,
                      "skipped_steps": d(s0, s1, "skipped_steps") / paths,
                      "inscatter_lookups": d(k0, k1, "inscatter_lookups") / paths, "inscatter_fetches": d(f0, f1, "inscatter_fetches") / paths,
                      "march_phase_lanes": d(s0, s1, "march_lanes") / max(d(s0, s1, "march_phases"), 1),
                      "scatter_phase_lanes": d(s0, s1, "scatter_lanes") / max(d(s0, s1, "scatter_phases"), 1)}}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""When do the waves of a short launch end?  (Diagnostics build with -DCT_STATS_FINE: CT_EXTRA_FLAGS=-DCT_STATS_FINE python -m deepestscatter_amd.build --force)"""
import os, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["CT_STATS"] = "1"
import deepestscatter_amd as ds
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 10
vol = int(sys.argv[2]) if len(sys.argv) > 2 else 512
size = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
shards = int(sys.argv[4]) if len(sys.argv) > 4 else 1
tex = ds.make_procedural_cloud(vol)
tr = ds.CloudTracer(tex, width=size, height=size, shard_index=0, shard_count=shards)
tr.render_accumulate(1, 32)
first = 33
for _ in range(8):
    tr.render_accumulate_async(first, spp); first += spp
tr.synchronize()
a = tr.debug_stats()
for _ in range(20):
    tr.render_accumulate_async(first, spp); first += spp
tr.synchronize()
b = tr.debug_stats()
d = {k: (b[k] - a[k] if isinstance(b[k], int) else [x - y for x, y in zip(b[k], a[k])]) for k in b if k != "max_scheduler_visits_of_a_wave"}
print("waves", d["waves"], "regen phases", d["regen_phases"], "lanes/regen %.1f" % (d["regen_lanes"] / max(d["regen_phases"], 1)),
      "march phases", d["march_phases"], "lanes %.1f" % (d["march_lanes"] / max(d["march_phases"], 1)),
      "scatter phases", d["scatter_phases"], "lanes %.1f" % (d["scatter_lanes"] / max(d["scatter_phases"], 1)))
print("wave end (0.25 ms bins from its start):", d["wave_end_hist_5ms"])
print("wave end minus drained (0.05 ms bins):", d["wave_end_minus_drained_hist_0p5ms"])
print("max visits", b["max_scheduler_visits_of_a_wave"])
rms, ams, n = tr.kernel_time()
print("kernel time", rms, ams, n)

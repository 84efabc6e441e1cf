"""Hit rate of the march kernel's per-lane NEE footprint cache on the benchmark scene (needs CT_STATS=1)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
assert os.environ.get("CT_STATS"), "run with CT_STATS=1"
import torch
import deepestscatter_amd as ds
t = ds.CloudTracer(ds.make_procedural_cloud(512), width=1024, height=1024)
t.render_accumulate(1, 32)
s0, k0 = t.debug_stats(), t.counters()
t.render_accumulate(33, 64)
s1, k1 = t.debug_stats(), t.counters()
print("reused", s1["nee_footprints_reused"] - s0["nee_footprints_reused"], "of", k1["inscatter_lookups"] - k0["inscatter_lookups"],
      (s1["nee_footprints_reused"] - s0["nee_footprints_reused"]) / (k1["inscatter_lookups"] - k0["inscatter_lookups"]))
import numpy as np
from deepestscatter_amd.cloudtrace import _p, check
ex = np.zeros(72, np.uint64)
check(t.L.ct_debug_stats_ex(t.h, _p(ex), 72), t.h)
print("since create: fetches that a SECOND entry (the footprint before the one replaced) would have held:", int(ex[70]), "a third:", int(ex[71]),
      "of", k1["inscatter_lookups"], "lookups,", s1["nee_footprints_reused"], "reused by the one entry")

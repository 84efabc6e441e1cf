import os, sys
sys.path.insert(0, "/root/repo")
import torch
import deepestscatter_amd as ds
t = ds.CloudTracer(ds.make_procedural_cloud(512), width=1024, height=1024)
t.render_accumulate(1, 32)
s0, k0 = t.debug_stats(), t.counters()
t.render_accumulate(33, 64)
s1, k1 = t.debug_stats(), t.counters()
print("reused", s1["nee_footprints_reused"] - s0["nee_footprints_reused"], "of", k1["inscatter_lookups"] - k0["inscatter_lookups"],
      (s1["nee_footprints_reused"] - s0["nee_footprints_reused"]) / (k1["inscatter_lookups"] - k0["inscatter_lookups"]))

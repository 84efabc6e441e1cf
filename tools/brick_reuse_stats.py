#!/usr/bin/env python3
"""Would a brick cache in LDS find anything?  (Round-2 review, item 5: measure first.)  Diagnostics build of the MARCH kernel:
of all march fetches, how many touch the 128-B brick line the lane's previous fetch touched, and how many touch a line that
another (lower) lane of the same wave fetches in the same instruction.

    python tools/brick_reuse_stats.py [volume=512] [size=1024] [spp=64]
"""
import os, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["CT_STATS"] = "1"
import deepestscatter_amd as ds
vol = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
tex = ds.make_procedural_cloud(vol)
tr = ds.CloudTracer(tex, width=size, height=size)
tr.render_accumulate(1, 32)
a = tr.debug_stats()
tr.render_accumulate(33, spp)
b = tr.debug_stats()
f = b["fetched_steps"] - a["fetched_steps"]
same = b["march_fetch_same_line_as_lanes_previous"] - a["march_fetch_same_line_as_lanes_previous"]
dup = b["march_fetch_line_shared_with_a_lower_lane"] - a["march_fetch_line_shared_with_a_lower_lane"]
print(json.dumps({"volume": vol, "size": size, "spp": spp, "march_fetches": f, "same_line_as_the_lanes_previous_fetch": same / f,
                  "line_also_fetched_by_a_lower_lane_of_the_wave": dup / f,
                  "march_lanes_per_phase": (b["march_lanes"] - a["march_lanes"]) / max(b["march_phases"] - a["march_phases"], 1)}))

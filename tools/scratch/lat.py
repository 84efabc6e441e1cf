import sys, time, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import deepestscatter_amd as ds
est = int(sys.argv[1]) if len(sys.argv) > 1 else 0
tex = ds.make_procedural_cloud(256)
tr = ds.CloudTracer(tex, ds.SceneParams(width=64, height=64, mode=1, estimator=est))
pos, dirs = tr.generate_scatter_samples(2048, 0)
tr.point_radiance_launch(ds.make_point_tasks(pos[:64], dirs[:64]), 1, 1)
# find deep tasks: one launch of all 2048 tasks x 1 frame each, per-task bounce count from separate launches of 64 tasks
deep = None
for i in range(0, 2048, 64):
    k0 = tr.counters(); r0 = tr.kernel_time()
    tr.point_radiance_launch(ds.make_point_tasks(pos[i:i+64], dirs[i:i+64]), 1, 1)
    k1 = tr.counters(); r1 = tr.kernel_time()
    b = (k1["inscatter_lookups"] - k0["inscatter_lookups"])
    if i < 640:
        print(i, "bounces", b, "kernel_ms %.3f" % (r1[0] - r0[0]), "us/bounce of longest (if 2000): %.2f" % ((r1[0]-r0[0]) * 1e3 / 2000))
for n in (1, 8, 64, 640, 6400):
    k0 = tr.counters(); r0 = tr.kernel_time()
    tr.point_radiance_launch(ds.make_point_tasks(pos[:64], dirs[:64]), 1, n)
    k1 = tr.counters(); r1 = tr.kernel_time()
    print("64 tasks x", n, "frames: kernel_ms %.3f" % (r1[0] - r0[0]), "bounces", k1["inscatter_lookups"] - k0["inscatter_lookups"])

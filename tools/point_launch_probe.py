#!/usr/bin/env python3
"""How a point-radiance launch (ct_point_radiance_launch, the device part of RadianceCollector::update) scales with its
size: time, bounces per experiment and lookups for 20480 threads x {25, 100, 400, 1600} frames on the procedural cloud.

    python tools/point_launch_probe.py [volume=256] [estimator=0]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import deepestscatter_amd as ds  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    est = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tex = ds.make_procedural_cloud(n)
    tr = ds.CloudTracer(tex, ds.SceneParams(width=64, height=64, mode=1, estimator=est))
    pos, dirs = tr.generate_scatter_samples(2048, 0)
    tasks = ds.make_point_tasks(np.repeat(pos, 10, axis=0), np.repeat(dirs, 10, axis=0), ids=np.repeat(np.arange(2048), 10))
    tr.point_radiance_launch(tasks.copy(), 1, 4)          # warm-up
    out = []
    for launches in (25, 100, 100, 400, 1600):
        k0 = tr.counters()
        r0 = tr.kernel_time()
        t0 = time.perf_counter()
        tr.point_radiance_launch(tasks.copy(), 1, launches)
        dt = time.perf_counter() - t0
        k1 = tr.counters()
        r1 = tr.kernel_time()
        paths = k1["paths"] - k0["paths"]
        rec = {"launches": launches, "experiments": int(paths), "wall_ms": dt * 1e3, "kernel_ms": r1[0] - r0[0],
               "bounces_per_experiment": (k1["inscatter_lookups"] - k0["inscatter_lookups"]) / paths,
               "lookups_per_experiment": (k1["density_lookups"] - k0["density_lookups"]) / paths,
               "Mexperiments_per_s": paths / dt / 1e6, "Gbounces_per_s": (k1["inscatter_lookups"] - k0["inscatter_lookups"]) / dt / 1e9}
        out.append(rec)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A display loop while the camera moves: every pose is new (Camera::rotate -> reset, Camera.cpp:93-98), and what counts is
how long the first update of a pose takes -- primary rays and pixel list, the cost-measuring launch, the job list, 10
subframes, the tonemap on the screen.    python tools/first_update_bench.py [--size 1024] [--poses 24]"""
import argparse, json, math, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--poses", type=int, default=24)
    ap.add_argument("--updates-per-pose", type=int, default=1)
    a = ap.parse_args()
    import deepestscatter_amd as ds
    W, H = a.size, a.height or a.size
    tex = ds.make_procedural_cloud(a.volume)
    tr = ds.CloudTracer(tex, width=W, height=H)
    tr.render_accumulate_async(1, 10); tr.tonemap(0.4)
    times = []
    for k in range(a.poses):
        phi = 0.05 * (k + 1)
        eye = (2.5 * math.cos(phi), -0.4, 2.5 * math.sin(phi))
        U, V, Wv = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, W / H)
        t0 = time.perf_counter()
        tr.set_camera(eye, U, V, Wv)
        tr.reset()
        for u in range(a.updates_per_pose):
            tr.render_accumulate_async(10 * u + 1, 10)
        tr.tonemap(0.4)                      # the screen on the host: the update is visible
        times.append((time.perf_counter() - t0) * 1e3)
    times.sort()
    print(json.dumps({"frame": [W, H], "poses": a.poses, "updates_per_pose": a.updates_per_pose,
                      "ms_per_pose_median": times[len(times) // 2], "ms_per_pose_min": times[0], "ms_per_pose_max": times[-1]}))

#!/usr/bin/env python3
"""A 1024-spp image FROM A NEW POSE (the bench's steps are steady state): the cost-measuring launch of the pose, the rest of the
job, the closing resume-only launch.  CT_TUNE_SUBFRAMES sets the length of the first.
    python tools/first_image_bench.py [--spp 1024]"""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--estimator", type=int, default=0)
    a = ap.parse_args()
    import deepestscatter_amd as ds
    tex = ds.make_procedural_cloud(a.volume)
    tr = ds.CloudTracer(tex, width=a.size, height=a.size, estimator=a.estimator)
    tr.render_accumulate_async(1, a.spp); tr.synchronize()          # scratch, job lists: not what is measured
    poses = [(2.5, -0.4, 0.0), (2.3, 0.6, 0.7), (-2.0, 0.5, 1.4), (2.5, -0.4, 0.0)]
    out = []
    for eye in poses:
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, 1.0)
        tr.set_camera(eye, U, V, W)
        tr.reset()
        t0 = time.perf_counter()
        tr.render_accumulate_async(1, a.spp)
        tr.synchronize()
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        tr.render_accumulate_async(a.spp + 1, a.spp)
        tr.synchronize()
        dt2 = time.perf_counter() - t1
        out.append({"eye": eye, "first_job_ms": dt * 1e3, "second_job_ms": dt2 * 1e3, "checksum": float(tr.mean().sum())})
    print(json.dumps({"tune_subframes": os.environ.get("CT_TUNE_SUBFRAMES", "32"), "spp": a.spp, "jobs": out,
                      "mean_first_job_ms": sum(o["first_job_ms"] for o in out) / len(out),
                      "mean_second_job_ms": sum(o["second_job_ms"] for o in out) / len(out)}))

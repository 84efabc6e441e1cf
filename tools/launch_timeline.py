#!/usr/bin/env python3
"""When do the waves of an estimator launch start and end?  (A diagnostics build -- CT_EXTRA_FLAGS=-DCT_DIAG_TIMELINE python -m
deepestscatter_amd.build --force -- with CT_TIMELINE=1: every wave stores both on the 100 MHz wall clock.)
Steady state of enqueued launches of --spp subframes; the last one's timeline: how many waves are resident over time.
    CT_TIMELINE=1 python tools/launch_timeline.py [--spp 10] [--size 1024]"""
import argparse, ctypes as C, json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=10)
    ap.add_argument("--updates", type=int, default=40)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024)
    a = ap.parse_args()
    os.environ["CT_TIMELINE"] = "1"
    import deepestscatter_amd as ds
    from deepestscatter_amd._lib import check
    tex = ds.make_procedural_cloud(a.volume)
    tr = ds.CloudTracer(tex, width=a.size, height=a.size)
    tr.render_accumulate(1, 30)
    first = 31
    for _ in range(a.updates):
        tr.render_accumulate_async(first, a.spp); first += a.spp
        tr.tonemap_async(0.4)
    waves = 6144
    out = np.zeros(4 * waves, np.uint64)
    check(tr.L.ct_debug_timeline(tr.h, out.ctypes.data_as(C.c_void_p), waves), tr.h)
    t = out.reshape(waves, 4).astype(np.int64)
    t = t[(t[:, 0] != 0) & (t[:, 1] != 0)]
    t0 = t[:, 0].min()
    s, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0      # microseconds
    total = e.max()
    grid = np.linspace(0, total, 41)
    resident = [(int(((s <= x) & (e > x)).sum())) for x in grid[:-1]]
    print(json.dumps({"spp_per_launch": a.spp, "waves": len(t), "launch_us": float(total),
                      "start_us_p50_p99_max": [float(np.percentile(s, 50)), float(np.percentile(s, 99)), float(s.max())],
                      "end_us_p01_p10_p50_p90": [float(np.percentile(e, 1)), float(np.percentile(e, 10)), float(np.percentile(e, 50)), float(np.percentile(e, 90))],
                      "mean_resident_waves": float(((e - s).sum()) / total), "resident_waves_over_time_40_bins": resident,
                      "learnt_us_p01_p50_p99": [float(np.percentile((t[:, 2] - t0) / 100.0, q)) for q in (1, 50, 99)],
                      "end_minus_learnt_us_p10_p50_p90_p99_max": [float(np.percentile((t[:, 1] - t[:, 2]) / 100.0, q)) for q in (10, 50, 90, 99, 100)],
                      "live_lanes_when_learnt_mean": float((t[:, 3] & 255).mean()), "old_lanes_when_learnt_mean": float(((t[:, 3] >> 8) & 255).mean()),
                      "waves_with_old_lanes": int((((t[:, 3] >> 8) & 255) > 0).sum()), "waves_with_job_unfinished": int((((t[:, 3] >> 16) & 1) > 0).sum())}))
    tr.close()

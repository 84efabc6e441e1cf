#!/usr/bin/env python3
"""Progressive rendering at the reference's cadence: Camera::render does 10 subframes, tonemaps, and returns to the display
loop (Camera.cpp:189-214).  Measures ms per update and Msamples/s for enqueued batches of --spp subframes with the display
update (ct_tonemap_async) behind each, against the same job as one long batch.

    python tools/progressive_bench.py [--spp 10] [--updates 100] [--estimator 0] [--volume 512] [--size 1024]
"""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np


def run_reference_loop(a, spp, updates):
    """Camera::render as the reference runs it, every call waited for: isConverged(), `spp` subframes, tonemap to the host."""
    import deepestscatter_amd as ds
    tr = ds.CloudTracer(run.tex, width=a.size, height=(a.height or a.size), estimator=a.estimator)
    tr.render_accumulate(1, 30)
    first = 31
    for _ in range(4):
        tr.render_accumulate(first, spp); first += spp
    t0 = time.perf_counter()
    for _ in range(updates):
        tr.is_converged()
        tr.render_accumulate(first, spp); first += spp
        tr.tonemap(0.4)
    dt = time.perf_counter() - t0
    tr.close()
    return {"loop": "reference (waited-for: is_converged, render_accumulate, tonemap)", "spp_per_update": spp, "updates": updates,
            "ms_per_update": dt / updates * 1e3, "Msamples_per_s": a.size * (a.height or a.size) * spp * updates / dt / 1e6}


def run(a, spp, updates, tonemap=True, ahead=0, stop=False):
    import deepestscatter_amd as ds
    tex = run.tex
    tr = ds.CloudTracer(tex, width=a.size, height=(a.height or a.size), estimator=a.estimator)
    tr.render_accumulate(1, 30)                       # cost-measuring launch of the pose (updates of 10 then end on multiples of 10)
    first = 31
    warm = max(4, 64 // spp)
    if ahead:
        # (whole launches in the warm-up and in the timed region: what is timed is what is counted)
        tr.set_render_ahead(ahead)
        per = max(1, ahead // spp)
        warm = 2 * per
        updates = max(per, updates // per * per)
    if stop:
        tr.set_stop_when_converged(10, 100)           # the reference's test, on the device behind every 10th subframe
    for _ in range(warm):                             # warm-up: scratch ring, job list for this batch size
        tr.render_accumulate_async(first, spp); first += spp
    tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(updates):
        tr.render_accumulate_async(first, spp); first += spp
        if tonemap:
            tr.tonemap_async(0.4)
        if stop:
            tr.converged_at()
    tr.synchronize()
    dt = time.perf_counter() - t0
    out = {"spp_per_update": spp, "updates": updates, "tonemap_every_update": tonemap, "ms_per_update": dt / updates * 1e3,
           "Msamples_per_s": a.size * (a.height or a.size) * spp * updates / dt / 1e6, "suspended_paths": tr.debug_suspended(),
           "checksum": float(tr.mean().astype(np.float64).sum()), "render_ahead": ahead,
           "stop_when_converged": (list(tr.converged_at()) if stop else None),
           "asked_subframes": first - 1, "rendered_subframes": tr.rendered_subframes()}
    tr.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, nargs="+", default=[10])
    ap.add_argument("--updates", type=int, default=100)
    ap.add_argument("--estimator", type=int, default=0)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024, help="frame width (and height, unless --height is given)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--ahead", type=int, nargs="*", default=[], help="also with ct_set_render_ahead(N) for every N given")
    ap.add_argument("--stop", action="store_true", help="also with ct_set_stop_when_converged(10, 100), and the reference's waited-for loop")
    ap.add_argument("--reference-spp", type=int, default=1000, help="the long batch the rate is compared with (0 = skip)")
    a = ap.parse_args()
    import deepestscatter_amd as ds
    run.tex = ds.make_procedural_cloud(a.volume)
    res = {"max_age_env": os.environ.get("CT_MAX_AGE", ""), "runs": []}
    for spp in a.spp:
        r = run(a, spp, max(2, a.updates * 10 // spp) if spp != 10 else a.updates)
        print(json.dumps(r), flush=True)
        res["runs"].append(r)
        for ahead in a.ahead:
            if ahead > spp:
                r = run(a, spp, a.updates, ahead=ahead)
                print(json.dumps(r), flush=True)
                res["runs"].append(r)
        if a.stop:
            for ahead in [0] + [x for x in a.ahead if x > spp]:
                r = run(a, spp, a.updates, ahead=ahead, stop=True)
                print(json.dumps(r), flush=True)
                res["runs"].append(r)
            r = run_reference_loop(a, spp, a.updates)
            print(json.dumps(r), flush=True)
            res["reference_loop"] = r
    if a.reference_spp:
        r = run(a, a.reference_spp, 2, tonemap=False)
        print("reference", json.dumps(r), flush=True)
        for x in res["runs"]:
            x["fraction_of_long_batch_rate"] = x["Msamples_per_s"] / r["Msamples_per_s"]
        res["reference"] = r
    print("progressive_summary", json.dumps(res))

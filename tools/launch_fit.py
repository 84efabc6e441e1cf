#!/usr/bin/env python3
"""What does an estimator launch cost besides its samples?  Enqueued steps of S subframes for several S, on the whole frame
and on one shard of an 8-way job; the per-launch durations come from the library's own events (CT_TRACE lines on stderr,
parsed by the caller: tools/gpu_launch_fit.sh) and from ct_kernel_time.

    CT_TRACE=1 python tools/launch_fit.py [--world 1 8] [--spp 128 256 512 1024] 2> trace.txt
"""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--spp", type=int, nargs="+", default=[128, 256, 512, 1024])
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024)
    a = ap.parse_args()
    import deepestscatter_amd as ds
    tex = ds.make_procedural_cloud(a.volume)
    for world in a.world:
        p = ds.SceneParams(width=a.size, height=a.size)
        p.shard_index, p.shard_count = 0, world
        tr = ds.CloudTracer(tex, p)
        tr.render_accumulate(1, 32)
        first = 33
        for S in a.spp:
            print(f"[fit] world {world} S {S} begin", file=sys.stderr, flush=True)
            tr.render_accumulate_async(first, S); first += S      # warm-up: layout, job list
            tr.synchronize()
            k0 = tr.kernel_time()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                tr.render_accumulate_async(first, S); first += S
            tr.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            k1 = tr.kernel_time()
            print(json.dumps({"world": world, "S": S, "steps": a.steps, "wall_ms_per_step": dt / a.steps,
                              "render_ms_per_step": (k1[0] - k0[0]) / a.steps, "accumulate_ms_per_step": (k1[1] - k0[1]) / a.steps}), flush=True)
            print(f"[fit] world {world} S {S} end", file=sys.stderr, flush=True)
        tr.close()

#!/usr/bin/env python3
"""What the ranks of an N-GPU bench run do, one after the other on ONE GPU and without torch.distributed: rank r
renders the tiles of shard r of `--world` for the step bench.py gives it -- the fixed 1024-subframe job by default
(strong scaling: `python bench.py --gpus N`), or 512 x world subframes with --weak.  Prints per-rank step times and
the figures the N-GPU run depends on: max/mean of the ranks' step times (the imbalance of the tile map) and the
speed-up that the slowest rank allows over the N=1 step (the 32 MiB reduce aside).

    python tools/shard_rank_rehearsal.py --world 8            # all 8 ranks
    python tools/shard_rank_rehearsal.py --world 8 --rank 3   # one rank
"""
import argparse
import json
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401  (before libcloudtrace, see tests/conftest.py)
import numpy as np
import deepestscatter_amd as ds

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=-1, help="-1 = every rank in turn")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--volume", type=int, default=512)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--weak", action="store_true")
ap.add_argument("--estimator", type=int, default=0)
ap.add_argument("--json", default="")
a = ap.parse_args()
tex = ds.make_procedural_cloud(a.volume)


def run(rank, world):
    S = 512 * world if a.weak else 1024
    p = ds.SceneParams(width=a.size, height=a.size, estimator=a.estimator)
    p.shard_index, p.shard_count = rank, world
    t = ds.CloudTracer(tex, params=p)
    t.render_accumulate_async(1, S)
    t.synchronize()
    k0, kt0 = t.counters(), t.kernel_time()
    t0 = time.perf_counter()
    for i in range(a.steps):
        t.render_accumulate_async(1 + S * (i + 1), S)
    t.synchronize()
    dt = time.perf_counter() - t0
    k1, kt1 = t.counters(), t.kernel_time()
    m = t.mean()
    own = m[..., 3] != 0
    rec = {"rank": rank, "world": world, "spp_per_step": S, "ms_per_step": dt / a.steps * 1e3,
           "estimator_ms_per_step": (kt1[0] - kt0[0]) / a.steps, "launches_per_step": (kt1[2] - kt0[2]) / a.steps,
           "accumulate_ms_per_step": (kt1[1] - kt0[1]) / a.steps, "own_pixels": int(own.sum()),
           "box_hits_per_subframe": (k1["box_hits"] - k0["box_hits"]) / (a.steps * S),
           "lookups_per_sample": (k1["density_lookups"] - k0["density_lookups"]) / max(k1["paths"] - k0["paths"], 1),
           "finite": bool(np.isfinite(m).all())}
    t.close()
    print(json.dumps(rec), flush=True)
    return rec


ranks = range(a.world) if a.rank < 0 else [a.rank]
recs = [run(r, a.world) for r in ranks]
out = {"ranks": recs}
if a.rank < 0 and a.world > 1:
    one = run(0, 1)
    ms = [r["ms_per_step"] for r in recs]
    out.update(n1=one, max_ms=max(ms), mean_ms=sum(ms) / len(ms), max_over_mean=max(ms) / (sum(ms) / len(ms)),
               speedup_allowed_by_slowest_rank=one["ms_per_step"] / max(ms))
    print(f"world {a.world}: rank step times max {max(ms):.1f} ms, mean {sum(ms) / len(ms):.1f} ms (max/mean {out['max_over_mean']:.3f}); "
          f"N=1 step {one['ms_per_step']:.1f} ms -> the slowest rank allows {out['speedup_allowed_by_slowest_rank']:.2f}x", flush=True)
if a.json:
    Path(a.json).write_text(json.dumps(out, indent=1))

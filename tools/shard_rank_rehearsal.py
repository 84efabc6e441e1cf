#!/usr/bin/env python3
"""What ONE rank of an N-GPU bench run does, on one GPU and without torch.distributed: the tiles of shard
`--rank` of `--world`, 512 x world subframes per step, steps enqueued.  Per-rank work is the N=1 step's, so
ms/step should match `python bench.py` (the all-reduce aside).

    python tools/shard_rank_rehearsal.py --world 8 --rank 3
"""
import argparse
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
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--volume", type=int, default=512)
ap.add_argument("--size", type=int, default=1024)
a = ap.parse_args()
S = 512 * a.world
tex = ds.make_procedural_cloud(a.volume)
p = ds.SceneParams(width=a.size, height=a.size)
p.shard_index, p.shard_count = a.rank, a.world
t = ds.CloudTracer(tex, params=p)
t.render_accumulate_async(1, S)
t.synchronize()
k0 = t.counters()
kt0 = t.kernel_time()
t0 = time.perf_counter()
for i in range(a.steps):
    t.render_accumulate_async(1 + S * (i + 1), S)
t.synchronize()
dt = time.perf_counter() - t0
k1 = t.counters()
kt1 = t.kernel_time()
m = t.mean()
own = m[..., 3] != 0
print(f"rank {a.rank}/{a.world}: {dt / a.steps * 1e3:.1f} ms/step of {S} subframes, "
      f"{(k1['paths'] - k0['paths']) / dt / 1e6:.0f} Msamples/s on this rank's tiles, own pixels {int(own.sum())}, "
      f"finite {bool(np.isfinite(m).all())}, mean radiance {float(m[..., 0][own].mean()):.5f}; per step: estimator "
      f"{(kt1[0] - kt0[0]) / a.steps:.1f} ms in {(kt1[2] - kt0[2]) / a.steps:.1f} launches, accumulate {(kt1[1] - kt0[1]) / a.steps:.2f} ms, "
      f"lookups/sample {(k1['density_lookups'] - k0['density_lookups']) / (k1['paths'] - k0['paths']):.1f}")

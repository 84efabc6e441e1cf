"""The N>1 path on CPU: two gloo ranks each hold only their pixel-tile shard of a radiance buffer
(produced by the oracle, since there is no GPU here) -- mean and Welford M2, stacked the way ShardedTracer
stages them -- and the frame reduce must rebuild the whole image and its M2 exactly.  Exercises deepestscatter_amd.distributed.frame_reduce, the tile->shard map and the
'foreign pixels are exactly zero' contract that makes SUM a merge."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import _oracle as O
    import deepestscatter_amd as ds
    from deepestscatter_amd.distributed import frame_gather, frame_reduce, shard_indices
    from conftest import sphere_volume

    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, spp = 40, 24, 3
    tex = sphere_volume(24, seed=3)
    orc = O.Oracle(tex, w, h, mode=0, threads=2)
    mean, m2 = orc.render(spp)
    mask = ds.shard_mask(w, h, rank, world)
    local = np.stack([mean, m2])          # the [2, H, W, 4] buffer ShardedTracer reduces with ONE collective
    local[:, ~mask] = 0                   # what a shard's handle holds: its tiles, zeros elsewhere
    t = torch.from_numpy(local)
    frame_reduce(t, 0)
    # the gather merge (every rank sends only its own tiles, 1/world of the bytes) must rebuild the same frame
    idx = [torch.from_numpy(i) for i in shard_indices(w, h, world)]
    assert sorted(int(v) for i in idx for v in i) == list(range(w * h))        # the shards partition the frame
    n_max = max(i.numel() for i in idx)
    g = torch.from_numpy(local.copy())
    packed = torch.zeros((2, n_max, 4), dtype=torch.float32)
    recv = [torch.zeros_like(packed) for _ in range(world)] if rank == 0 else None
    frame_gather(g, idx, packed, recv, 0)
    # every rank also checks the partition property with an all-reduce of the masks
    m = torch.from_numpy(mask.astype(np.int32))
    dist.all_reduce(m)
    assert int(m.min()) == 1 and int(m.max()) == 1
    if rank == 0:
        np.save(os.path.join(out_dir, "merged.npy"), t.numpy())
        np.save(os.path.join(out_dir, "gathered.npy"), g.numpy())
        np.save(os.path.join(out_dir, "whole.npy"), np.stack([mean, m2]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_frame_reduce_rebuilds_the_whole_frame(tmp_path, world):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    merged = np.load(tmp_path / "merged.npy")
    whole = np.load(tmp_path / "whole.npy")
    assert merged.shape == (2, 24, 40, 4)
    assert np.array_equal(merged, whole)            # mean AND Welford M2: the sum of disjoint shards is the whole frame
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), whole)       # ... and so is the gather of every shard's own tiles
    assert merged[0, ..., :3].max() > 0 and merged[1, ..., :3].max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("merge", ["reduce", "gather"])
def test_whole_bench_under_torchrun_with_two_ranks_on_one_gpu(merge):
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus 2), with the
    two ranks sharing the box's one GPU and gloo carrying the collectives (RCCL refuses two ranks per device): the torchrun
    path -- env parsing, sharding, staging, merge, max-over-ranks timing, the multi_gpu block of the JSON line -- cannot rot
    unseen between the rounds in which a multi-GPU node is available."""
    import json
    import subprocess
    port = 29700 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--single-device", "--volume", "64", "--width", "128", "--height", "96", "--spp-per-step", "24", "--no-cpu-baseline",
           "--merge", merge]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    m = out["multi_gpu"]
    assert out["rccl_ranks"] == m["rccl_ranks"] == m["world_size"] == 2 and m["backend"] == "gloo" and m["merge"] == merge
    assert len(m["ms_per_step_per_rank"]) == 2 and all(v > 0 for v in m["ms_per_step_per_rank"])
    assert len(m["merge_ms_per_step_per_rank"]) == 2

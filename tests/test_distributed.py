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
    from deepestscatter_amd.distributed import frame_reduce
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
    # every rank also checks the partition property with an all-reduce of the masks
    m = torch.from_numpy(mask.astype(np.int32))
    dist.all_reduce(m)
    assert int(m.min()) == 1 and int(m.max()) == 1
    if rank == 0:
        np.save(os.path.join(out_dir, "merged.npy"), t.numpy())
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
    assert merged[0, ..., :3].max() > 0 and merged[1, ..., :3].max() > 0

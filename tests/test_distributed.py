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


@pytest.mark.gpu
@pytest.mark.parametrize("merge", ["reduce", "gather"])
def test_sharded_tracer_merges_through_an_nccl_group_of_one_rank(merge):
    """The N > 1 merge on the real backend: torch.distributed "nccl" (= RCCL) initialised with ONE rank on the box's GPU,
    ShardedTracer staging and merging as it does for N ranks (reduce, and the gather of every rank's own tiles), enqueued and
    waited-for steps, tonemap of the merged frame.  gloo covers the multi-rank arithmetic (above); this covers RCCL, the
    device tensors and the stream ordering between the library's copies and torch's collective."""
    import subprocess
    port = 29900 + (os.getpid() % 90)
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "_nccl_one_rank.py"), merge, str(port)], capture_output=True, text=True,
                       timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and f"nccl one-rank merge ok: {merge}" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_eight_shards_at_full_size_merge_into_the_whole_frame():
    """BASELINE.json configs[3] at its own size: 512^3 / 1024^2 cut into EIGHT pixel-tile shards (2 subframes).  The eight
    handles' means and M2s sum to the single handle's frame bit for bit, the lookups and paths add up, and the same eight
    shards driven below the C ABI (ct_group_* with the device repeated eight times: merged by copies and an add kernel, since
    RCCL refuses two ranks per device) give that frame, its tonemap and its convergence count too."""
    import deepestscatter_amd as ds
    tex = ds.make_procedural_cloud(512)
    w = h = 1024
    whole = ds.CloudTracer(tex, width=w, height=h)
    whole.render_accumulate(1, 2)
    want = (whole.mean(), whole.m2(), whole.counters(), whole.tonemap(0.4), whole.is_converged())
    whole.close()
    mean, m2 = np.zeros_like(want[0]), np.zeros_like(want[1])
    total = {}
    covered = np.zeros((h, w), np.int32)
    for r in range(8):
        sh = ds.CloudTracer(tex, width=w, height=h, shard_index=r, shard_count=8)
        sh.render_accumulate(1, 2)
        m = sh.mean()
        mask = ds.shard_mask(w, h, r, 8)
        assert not m[~mask].any()                       # foreign pixels are exactly zero: SUM is a merge
        covered += mask
        mean += m
        m2 += sh.m2()
        for k, v in sh.counters().items():
            total[k] = total.get(k, 0) + v
        sh.close()
    assert covered.min() == 1 and covered.max() == 1
    assert np.array_equal(mean, want[0]) and np.array_equal(m2, want[1]) and total == want[2]
    g = ds.TracerGroup(tex, [0] * 8, width=w, height=h)
    g.render_accumulate(1, 2)
    assert np.array_equal(g.mean(), want[0]) and np.array_equal(g.m2(), want[1]) and g.counters() == want[2]
    screen, avg = g.tonemap(0.4)
    assert np.array_equal(screen, want[3][0]) and avg == want[3][1]
    assert g.is_converged() == want[4]
    g.close()


@pytest.mark.gpu
def test_bench_group_mode_and_torchrun_mode_produce_the_same_frame():
    """Two independent N-GPU implementations behind one bench line: `bench.py --gpus 2 --group` (ONE process, ct_group_* below
    the C ABI) and the torchrun path (one rank per shard over torch.distributed).  Both print the SHA-256 of the merged
    [mean | M2] frame; on this box's one GPU (shards sharing the device, gloo / the add kernel standing in for RCCL) the two
    hashes must equal each other and the N = 1 frame's -- the comparison the driver's 8-GPU node can repeat on RCCL."""
    import json
    import subprocess
    common = ["--steps", "2", "--warmup", "1", "--volume", "64", "--width", "128", "--height", "96", "--spp-per-step", "24", "--no-cpu-baseline",
              "--no-pmc-traffic", "--no-delta-leg", "--no-progressive-leg"]
    port = 29800 + (os.getpid() % 90)
    cmds = {
        "single": [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *common],
        "group": [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--group", "--single-device", *common],
        "torchrun": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", *common],
    }
    got = {}
    for name, cmd in cmds.items():
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
        assert r.returncode == 0, name + ": " + r.stdout[-1500:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
        assert len(lines) == 1, name
        got[name] = json.loads(lines[0])
        assert got[name]["subframes_in_the_frame"] == 72 and got[name]["value"] > 0
    assert got["group"]["n_gpus"] == 2 and got["group"]["multi_gpu"]["implementation"].startswith("ct_group")
    assert got["single"]["frame_sha256"] == got["group"]["frame_sha256"] == got["torchrun"]["frame_sha256"]


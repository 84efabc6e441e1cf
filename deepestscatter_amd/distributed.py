"""Multi-GPU: pixel-tile sharding + frame reduce.

One process per GPU (`torch.distributed`; backend "nccl" is RCCL on ROCm, "gloo" works on CPU for
tests).  Rank r owns the 8x8-pixel tiles (tx,ty) with (tx + 3*ty) % world == r; every rank keeps
full-frame, zero-initialised mean and M2 buffers and fills only its tiles, so the frame reduce is a
plain SUM of both and reproduces the single-GPU image (and its variance) bit for bit.  torch is plumbing here (device
memory for the staging tensor, the collective); all rendering happens in libcloudtrace.so.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib
from .cloudtrace import CloudTracer, SceneParams


def env_rank_world() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(width: int, height: int, world: int) -> list[np.ndarray]:
    """Flat pixel indices (y * width + x) of every shard's tiles, in ascending order: the tile-contiguous layout of the
    gather merge ("an all-gather over a tile-contiguous layout moves 1/G of the bytes")."""
    from .cloudtrace import shard_mask
    return [np.flatnonzero(shard_mask(width, height, r, world).reshape(-1)).astype(np.int64) for r in range(world)]


def frame_gather(local, indices: list, packed, recv: list | None, dst: int = 0):
    """The same merge with 1/world of the bytes: every rank packs the pixels of its own tiles (`indices[rank]`, flat pixel
    indices) from `local` [2, H, W, 4] into `packed` [2, n_max, 4], the packs are gathered on rank `dst`, which writes each
    into its place.  `recv`: world tensors like `packed` on rank dst, None elsewhere.  Exact: no arithmetic at all."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    # (a process group of ONE rank still goes through the backend: the single-GPU rehearsal of this path on RCCL)
    rank = dist.get_rank()
    flat = local.view(2, -1, 4)
    n = indices[rank].numel()
    packed[:, :n] = flat.index_select(1, indices[rank])
    staged = local.is_cuda and dist.get_backend() != "nccl"    # rehearsal backends: through the host
    src = packed.cpu() if staged else packed
    lst = None
    if rank == dst:
        lst = [r.cpu() for r in recv] if staged else recv
    dist.gather(src, lst, dst=dst)
    if rank == dst:
        for r, got in enumerate(lst):
            if r != dst:
                k = indices[r].numel()
                flat.index_copy_(1, indices[r], got[:, :k].to(local.device))
    return local


def frame_reduce(local, dst: int = 0):
    """SUM-reduce a per-rank radiance buffer (torch tensor, any device) to rank `dst`.  Tiles are
    disjoint and foreign pixels are exactly 0, so the sum is the merged frame."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if local.is_cuda and dist.get_backend() != "nccl":
            # rehearsal on a backend without device collectives (gloo): stage through the host
            host = local.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            local.copy_(host)
        else:
            dist.reduce(local, dst=dst, op=dist.ReduceOp.SUM)
    return local


class ShardedTracer:
    """A CloudTracer for this rank's shard plus the per-step frame reduce.

    `merged` is ONE [2, H, W, 4] tensor: the running mean and M2 (Welford) of this rank's tiles, zeros elsewhere; a
    step copies both into it and SUM-reduces it to rank 0 with one collective (tiles are disjoint, so the sum is an
    exact merge of both).  On rank 0 `tonemap()` and `is_converged()` then run on the merged frame
    (ct_tonemap_buffer / ct_is_converged_buffers): Reinhard's average luminance (reinhard.cu:44-55) and the count
    of unconverged pixels (Camera.cpp:232-268) are whole-frame quantities.

    The handle's stream is a torch stream of this object, so what the library enqueues (accumulate kernels, the
    copies into `merged`) and what torch enqueues (the RCCL reduce of `merged`) are ordered by plain stream order,
    with or without waiting in between."""

    def __init__(self, density: np.ndarray, params: SceneParams, rank: int, world: int, local_rank: int = 0,
                 stage_always: bool = False, merge: str = "reduce"):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        # "gather": every rank sends only its own tiles (1/world of the bytes); a single rank merges only when it is asked to
        # stage (stage_always: the one-GPU rehearsal, which then runs the chosen collective on a group of one)
        self.merge = merge if (world > 1 or stage_always) else "reduce"
        self._merge_events, self._merge_host_ms, self._merges = [], 0.0, 0
        self._merge_dev_ms, self._merge_dev_n = 0.0, 0
        params.shard_index, params.shard_count, params.device = rank, world, local_rank
        self.tracer = CloudTracer(density, params)
        self.stage = world > 1 or stage_always       # stage_always: single-rank rehearsal of the staged path
        self.merged = torch.zeros((2, params.height, params.width, 4), dtype=torch.float32, device="cuda")
        self._nbytes = self.merged[0].numel() * 4
        self._lagging = False
        self._merged_subframes = 0
        self.stream = None
        if self.merge == "gather":
            idx = shard_indices(params.width, params.height, world)
            n_max = max(len(i) for i in idx)
            self._idx = [torch.from_numpy(i).cuda() for i in idx]
            self._packed = torch.zeros((2, n_max, 4), dtype=torch.float32, device="cuda")
            self._recv = [torch.zeros_like(self._packed) for _ in range(world)] if rank == 0 else None
        if self.stage:
            torch.cuda.synchronize()                 # `merged` is zeroed before another stream touches it
            self.stream = torch.cuda.Stream()
            self.tracer.set_stream(self.stream.cuda_stream)

    @property
    def merged_mean(self):
        return self.merged[0]

    @property
    def merged_m2(self):
        return self.merged[1]

    def _collective(self):
        if self.merge == "gather":
            frame_gather(self.merged, self._idx, self._packed, self._recv, 0)
        else:
            frame_reduce(self.merged, 0)

    def _reduce(self):
        import torch.distributed as dist
        on_stream = not (dist.is_available() and dist.is_initialized()) or dist.get_backend() == "nccl"
        if on_stream:
            with self.torch.cuda.stream(self.stream):
                self._collective()
        else:
            # rehearsal backends stage through the host: wait for the copy first
            self.stream.synchronize()
            self._collective()

    def _stage(self, wait: bool):
        import time
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(self.stream)
        copy = self.tracer.copy_to_device if wait else self.tracer.copy_to_device_async
        copy(_lib.CT_BUF_MEAN, self.merged[0].data_ptr(), self._nbytes)
        copy(_lib.CT_BUF_M2, self.merged[1].data_ptr(), self._nbytes)
        self._reduce()
        e1.record(self.stream)
        self._merge_events.append((e0, e1))
        # (a long progressive run merges once per step: finished pairs are folded into a running sum, so neither the list nor
        # merge_ms() grows with the number of steps)
        while len(self._merge_events) > 8 and self._merge_events[0][1].query():
            a, b = self._merge_events.pop(0)
            self._merge_dev_ms += a.elapsed_time(b)
            self._merge_dev_n += 1
        self._merge_host_ms += (time.perf_counter() - t0) * 1e3
        self._merges += 1

    def merge_ms(self) -> float:
        """Device time per merge (the two copies into the staging buffer + the collective), averaged over the merges so far;
        with a rehearsal backend, whose collective runs on the host, the host time of the call instead."""
        if not self._merges:
            return 0.0
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
            return self._merge_host_ms / self._merges
        self.stream.synchronize()
        for a, b in self._merge_events:
            self._merge_dev_ms += a.elapsed_time(b)
            self._merge_dev_n += 1
        self._merge_events.clear()
        return self._merge_dev_ms / max(self._merge_dev_n, 1)

    def step(self, first_subframe: int, count: int):
        """Render + accumulate `count` subframes of this shard, then reduce [mean | M2] to rank 0.
        Returns the merged buffer (valid on rank 0)."""
        self.tracer.render_accumulate(first_subframe, count)
        self._merged_subframes = first_subframe + count - 1
        if self.stage:
            self._stage(True)
            return self.merged
        return None

    def step_async(self, first_subframe: int, count: int):
        """The same without waiting: the batch is enqueued (ct_render_accumulate_async).  Its estimator
        launch hands its surviving paths to the next one, so its accumulate kernel runs behind the NEXT
        batch's launch: the running mean that this step copies and reduces is the one after the previous
        batch.  `synchronize()` finishes the last batch and reduces once more; `merged` is complete on rank
        0 after it."""
        self.tracer.render_accumulate_async(first_subframe, count)
        self._merged_subframes = first_subframe + count - 1
        if self.stage:
            self._stage(False)
            self._lagging = True
            return self.merged
        return None

    def synchronize(self):
        self.tracer.synchronize()
        if self.stage and self._lagging:
            self._stage(True)
            self._lagging = False
        if self.stream is not None:
            self.stream.synchronize()

    # -- whole-frame quantities, on the merged frame (rank 0 of a multi-GPU job; the handle's own buffers otherwise) ----
    def tonemap(self, exposure: float = 0.4):
        """-> (uint8 [H,W,4] screen, average luminance) of the merged frame.  Rank 0 only when world > 1."""
        self.synchronize()
        if not self.stage:
            return self.tracer.tonemap(exposure)
        if self.rank != 0:
            raise RuntimeError("the merged frame lives on rank 0")
        return self.tracer.tonemap_buffer(self.merged[0].data_ptr(), exposure)

    def is_converged(self):
        """Camera::isConverged over the merged frame -> (converged, unconverged pixels).  Rank 0 only when world > 1."""
        self.synchronize()
        if not self.stage:
            return self.tracer.is_converged()
        if self.rank != 0:
            raise RuntimeError("the merged frame lives on rank 0")
        return self.tracer.is_converged_buffers(self.merged[0].data_ptr(), self.merged[1].data_ptr(), self._merged_subframes)

    def close(self):
        self.tracer.close()

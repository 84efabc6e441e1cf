"""Multi-GPU: pixel-tile sharding + frame reduce.

One process per GPU (`torch.distributed`; backend "nccl" is RCCL on ROCm, "gloo" works on CPU for
tests).  Rank r owns the 8x8-pixel tiles (tx,ty) with (tx + 3*ty) % world == r; every rank keeps
a full-frame, zero-initialised radiance buffer and fills only its tiles, so the frame reduce is a
plain SUM and reproduces the single-GPU image bit for bit.  torch is plumbing here (device
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


def frame_reduce(local, dst: int = 0):
    """SUM-reduce a per-rank radiance buffer (torch tensor, any device) to rank `dst`.  Tiles are
    disjoint and foreign pixels are exactly 0, so the sum is the merged frame."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
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

    The handle's stream is a torch stream of this object, so what the library enqueues (accumulate
    kernels, the copy of the running mean into `merged`) and what torch enqueues (the RCCL reduce of
    `merged`) are ordered by plain stream order, with or without waiting in between."""

    def __init__(self, density: np.ndarray, params: SceneParams, rank: int, world: int, local_rank: int = 0,
                 stage_always: bool = False):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        params.shard_index, params.shard_count, params.device = rank, world, local_rank
        self.tracer = CloudTracer(density, params)
        self.stage = world > 1 or stage_always       # stage_always: single-rank rehearsal of the staged path
        self.merged = torch.zeros((params.height, params.width, 4), dtype=torch.float32, device="cuda")
        self._nbytes = self.merged.numel() * 4
        self._lagging = False
        self.stream = None
        if self.stage:
            torch.cuda.synchronize()                 # `merged` is zeroed before another stream touches it
            self.stream = torch.cuda.Stream()
            self.tracer.set_stream(self.stream.cuda_stream)

    def _reduce(self):
        import torch.distributed as dist
        on_stream = not (dist.is_available() and dist.is_initialized()) or dist.get_backend() == "nccl"
        if on_stream:
            with self.torch.cuda.stream(self.stream):
                frame_reduce(self.merged, 0)
        else:
            # rehearsal backends stage through the host: wait for the copy first
            self.stream.synchronize()
            frame_reduce(self.merged, 0)

    def step(self, first_subframe: int, count: int):
        """Render + accumulate `count` subframes of this shard, then reduce the frame to rank 0.
        Returns the merged running mean (valid on rank 0)."""
        self.tracer.render_accumulate(first_subframe, count)
        if self.stage:
            self.tracer.copy_to_device(_lib.CT_BUF_MEAN, self.merged.data_ptr(), self._nbytes)
            self._reduce()
            return self.merged
        return None

    def step_async(self, first_subframe: int, count: int):
        """The same without waiting: the batch is enqueued (ct_render_accumulate_async).  Its estimator
        launch hands its surviving paths to the next one, so its accumulate kernel runs behind the NEXT
        batch's launch: the running mean that this step copies and reduces is the one after the previous
        batch.  `synchronize()` finishes the last batch and reduces once more; `merged` is complete on rank
        0 after it."""
        self.tracer.render_accumulate_async(first_subframe, count)
        if self.stage:
            self.tracer.copy_to_device_async(_lib.CT_BUF_MEAN, self.merged.data_ptr(), self._nbytes)
            self._reduce()
            self._lagging = True
            return self.merged
        return None

    def synchronize(self):
        self.tracer.synchronize()
        if self.stage and self._lagging:
            self.tracer.copy_to_device(_lib.CT_BUF_MEAN, self.merged.data_ptr(), self._nbytes)
            self._reduce()
            self._lagging = False
        if self.stream is not None:
            self.stream.synchronize()

    def close(self):
        self.tracer.close()

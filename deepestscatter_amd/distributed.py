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
    """A CloudTracer for this rank's shard plus the per-step frame reduce."""

    def __init__(self, density: np.ndarray, params: SceneParams, rank: int, world: int, local_rank: int = 0):
        import torch
        self.torch = torch
        self.rank, self.world = rank, world
        params.shard_index, params.shard_count, params.device = rank, world, local_rank
        self.tracer = CloudTracer(density, params)
        self.merged = torch.zeros((params.height, params.width, 4), dtype=torch.float32, device="cuda")
        self._nbytes = self.merged.numel() * 4
        self._shares_stream = False

    def step(self, first_subframe: int, count: int):
        """Render + accumulate `count` subframes of this shard, then reduce the frame to rank 0.
        Returns the merged running mean (valid on rank 0)."""
        self.tracer.render_accumulate(first_subframe, count)
        if self.world > 1:
            self.tracer.copy_to_device(_lib.CT_BUF_MEAN, self.merged.data_ptr(), self._nbytes)
            frame_reduce(self.merged, 0)
            return self.merged
        return None

    def step_async(self, first_subframe: int, count: int):
        """The same without waiting: the batch is enqueued (two may be in flight, see
        ct_render_accumulate_async); the copy of the running mean and the RCCL reduce are ordered behind
        its accumulate kernel on torch's current stream, which the handle shares.  `merged` is valid on
        rank 0 after `synchronize()`."""
        if self.world > 1 and not self._shares_stream:
            torch = self.torch
            if torch.distributed.get_backend() != "nccl":
                return self.step(first_subframe, count)     # host-staged rehearsal backends cannot be ordered on a stream
            self.tracer.set_stream(torch.cuda.current_stream().cuda_stream)
            self._shares_stream = True
        self.tracer.render_accumulate_async(first_subframe, count)
        if self.world > 1:
            self.tracer.copy_to_device_async(_lib.CT_BUF_MEAN, self.merged.data_ptr(), self._nbytes)
            frame_reduce(self.merged, 0)
            return self.merged
        return None

    def synchronize(self):
        self.tracer.synchronize()
        if self.world > 1:
            self.torch.cuda.synchronize()

    def close(self):
        self.tracer.close()

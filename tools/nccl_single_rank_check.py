#!/usr/bin/env python3
"""The multi-GPU step with the real RCCL backend on one rank (a second GPU is not available to the tests):
process group "nccl" of world size 1, ShardedTracer with the staged path forced, enqueued steps + the
collective on the shared torch stream; the staging tensor must hold the running mean and M2 afterwards."""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import torch.distributed as dist
import numpy as np
import deepestscatter_amd as ds
from deepestscatter_amd.distributed import ShardedTracer
from conftest import sphere_volume

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
tex = sphere_volume(36, radius=0.4, seed=5)
w, h = 160, 120
kw = dict(width=w, height=h, mode=0, cloud_size_m=20000.0, max_depth=300)
st = ShardedTracer(tex, ds.SceneParams(**kw), 0, 1, 0, stage_always=True)
ref = ds.CloudTracer(tex, **kw)
first = 1
for n in (3, 5, 2, 4):
    st.step_async(first, n)
    ref.render_accumulate(first, n)
    first += n
st.synchronize()
dist.barrier()
torch.cuda.synchronize()
ok = np.array_equal(st.merged_mean.cpu().numpy(), ref.mean()) and np.array_equal(st.merged_m2.cpu().numpy(), ref.m2())
st.step(first, 2); ref.render_accumulate(first, 2); st.synchronize()
ok = ok and np.array_equal(st.merged_mean.cpu().numpy(), ref.mean()) and np.array_equal(st.merged_m2.cpu().numpy(), ref.m2())
ok = ok and st.is_converged() == ref.is_converged()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
ok = ok and float(t.item()) == 1.5
st.close(); ref.close()
dist.destroy_process_group()
print("NCCL single-rank check:", "ok" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

"""Child process of tests/test_distributed.py: ShardedTracer's merge (reduce or gather) through a REAL nccl (= RCCL) process
group of one rank on the box's one GPU -- the collectives, the staging stream and the stream ordering between the library's
copies and torch's collective are the N > 1 code path; only the peer count differs.  TEST INFRASTRUCTURE.

    python tests/_nccl_one_rank.py reduce|gather <port>
"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    merge, port = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import deepestscatter_amd as ds
    from deepestscatter_amd.distributed import ShardedTracer
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    tex = ds.make_procedural_cloud(64)
    w, h = 160, 120
    st = ShardedTracer(tex, ds.SceneParams(width=w, height=h), 0, 1, 0, stage_always=True, merge=merge)
    assert st.merge == merge and st.stage
    first = 1
    for n in (6, 5, 9):
        st.step_async(first, n)        # enqueued: the merge of step k runs behind the launch of step k + 1
        first += n
    st.synchronize()
    merged = st.merged.cpu().numpy()
    assert np.array_equal(merged[0], st.tracer.mean()) and np.array_equal(merged[1], st.tracer.m2())
    st.step(first, 4)                  # ... and the waited-for form
    merged = st.merged.cpu().numpy()
    assert np.array_equal(merged[0], st.tracer.mean()) and np.array_equal(merged[1], st.tracer.m2())
    screen, avg = st.tonemap(0.4)
    s2, a2 = st.tracer.tonemap(0.4)
    assert np.array_equal(screen, s2) and avg == a2
    ones = torch.ones(1, device="cuda")
    dist.all_reduce(ones)
    assert float(ones.item()) == 1.0 and st.merge_ms() > 0.0
    # a fresh single handle renders the same image
    one = ds.CloudTracer(tex, width=w, height=h)
    one.render_accumulate(1, first + 3)
    assert np.array_equal(one.mean(), merged[0]) and np.array_equal(one.m2(), merged[1])
    one.close()
    st.close()
    dist.barrier()
    dist.destroy_process_group()
    print("nccl one-rank merge ok:", merge)


if __name__ == "__main__":
    main()

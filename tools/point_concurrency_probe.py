"""K renderer handles on one GPU, one host thread each, every thread calling ct_point_radiance_launch the way a
RadianceCollector does (20480 threads x 100 frames, 20 calls): aggregate ms per update and experiments/s.

    python tools/point_concurrency_probe.py K [estimator=0]

Under `rocprofv3 --kernel-trace` its trace feeds tools/kernel_overlap.py (how many estimator kernels are resident at once).
"""
import sys, time, threading
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import deepestscatter_amd as ds
K = int(sys.argv[1]); est = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tex = ds.make_procedural_cloud(256)
trs = [ds.CloudTracer(tex, ds.SceneParams(width=64, height=64, mode=1, estimator=est)) for _ in range(K)]
tasks = []
for i, tr in enumerate(trs):
    pos, dirs = tr.generate_scatter_samples(2048, i)
    t = ds.make_point_tasks(np.repeat(pos, 10, axis=0), np.repeat(dirs, 10, axis=0), ids=np.repeat(np.arange(2048), 10))
    tr.point_radiance_launch(t.copy(), 1, 100)
    tasks.append(t)
N = 20
def work(i):
    for k in range(N):
        trs[i].point_radiance_launch(tasks[i].copy(), 1 + 100 * k, 100)
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
[t.start() for t in th]; [t.join() for t in th]
dt = time.perf_counter() - t0
print("K=%d est=%d: %.2f ms per update (aggregate), %.1f Mexperiments/s" % (K, est, dt / (K * N) * 1e3, K * N * 2.048 / dt))

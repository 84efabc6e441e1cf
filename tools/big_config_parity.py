#!/usr/bin/env python3
"""BASELINE.json configs[4] (1024^3 volume, 2048^2 frame) against the oracle on a window, both estimators: at this
size the DELTA majorant cells are 32 texels wide, the per-XCD job queues are on and the batches are the 128 subframes
the 8 GiB scratch holds.  python tools/big_config_parity.py [spp]"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
import numpy as np
import deepestscatter_amd as ds
import _oracle as O

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t0 = time.time()
tex = ds.make_procedural_cloud(1024)
print(f"1024^3 cloud generated in {time.time() - t0:.1f} s", flush=True)
w = h = 2048
ins = None
bad = 0
for est in (0, 1):
    tr = ds.CloudTracer(tex, width=w, height=h, estimator=est)
    tr.render_accumulate_async(1, spp // 2)
    tr.render_accumulate_async(1 + spp // 2, spp - spp // 2)
    tr.synchronize()
    mean, m2 = tr.mean(), tr.m2()
    if ins is None:
        ins = tr.inscatter()
    tr.close()
    t0 = time.time()
    orc = O.Oracle(tex, w, h, fast=True, estimator=est, inscatter=ins)
    x0, y0 = 1000, 1040
    win = (x0, y0, x0 + 12, y0 + 12)
    rm, rm2 = orc.render(spp, window=win)
    ok = (np.array_equal(mean[y0:y0 + 12, x0:x0 + 12], rm[y0:y0 + 12, x0:x0 + 12]) and
          np.array_equal(m2[y0:y0 + 12, x0:x0 + 12], rm2[y0:y0 + 12, x0:x0 + 12]))
    bad += 0 if ok else 1
    print(f"estimator {('MARCH', 'DELTA')[est]}: window {win} x {spp} spp {'bit-identical' if ok else 'MISMATCH'}, "
          f"window mean radiance {float(rm[y0:y0 + 12, x0:x0 + 12, 0].mean()):.4f}, "
          f"{'majorant cells of %d texels, ' % orc.scene.maj_cell if est else ''}oracle {time.time() - t0:.1f} s", flush=True)
print("done:", "ok" if bad == 0 else f"{bad} mismatches")

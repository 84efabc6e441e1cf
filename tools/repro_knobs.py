#!/usr/bin/env python3
"""One reported soak case (tools/repro_case.py) rendered synchronously under a few settings, counters against the oracle's:
    python tools/repro_knobs.py <seed> <case>"""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401
import numpy as np
import deepestscatter_amd as ds
from test_gpu_parity import _random_scene, make_pair

seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for _ in range(case + 1):
    kw, eye = _random_scene(rng)
    rng.random()
    pattern = [(int(n), bool(rng.random() < 0.6)) for n in rng.integers(1, 5, 4)]
tex = kw.pop("tex"); w, h = kw.pop("width"), kw.pop("height")
U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
total = sum(n for n, _ in pattern)
for name, env, over in (("as is", {}, {}), ("no pre-walked prefix", {"CT_NO_ADVANCE": "1"}, {}), ("mode 0", {}, {"mode": 0}),
                        ("mode 1", {}, {"mode": 1}), ("MARCH", {}, {"estimator": 0}), ("eye outside", {}, {"_eye": (2.5, -0.4, 0.0)}),
                        ("one subframe", {}, {"_total": 1})):
    for k, v in env.items():
        os.environ[k] = v
    k2 = dict(kw)
    k2.update({k: v for k, v in over.items() if not k.startswith("_")})
    e = over.get("_eye", eye)
    n = over.get("_total", total)
    Ue, Ve, We = ds.calculate_camera_variables(e, (0, 0, 0), (0, 1, 0), 30.0, w / h)
    tr, orc = make_pair(tex, w, h, **k2)
    tr.set_camera(e, Ue, Ve, We); orc.set_camera(e, Ue, Ve, We)
    tr.render_accumulate(1, n)
    mean, m2 = orc.render(n)
    c, oc = tr.counters(), orc.counters.as_dict()
    print(f"{name:22s}: image {'ok' if np.array_equal(tr.mean(), mean) else 'DIFFERS'}, density_lookups hip {c['density_lookups']} oracle {oc['density_lookups']} "
          f"(diff {c['density_lookups'] - oc['density_lookups']}), others {'ok' if all(c[k] == oc[k] for k in c if k != 'density_lookups') else 'DIFFER'}; fetches {tr.fetch_counters()}", flush=True)
    tr.close()
    for k in env:
        del os.environ[k]

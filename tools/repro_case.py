import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np
import deepestscatter_amd as ds, _oracle as O
from test_gpu_parity import _random_scene, make_pair
rng = np.random.default_rng(20261003)
target = int(sys.argv[1])
for case in range(target+1):
    kw, eye = _random_scene(rng)
    tex = kw.pop("tex"); w, h = kw.pop("width"), kw.pop("height")
    ns=[int(n) for n in rng.integers(1, 5, 4)]; modes=[rng.random() < 0.6 for _ in ns]
    if case != target: continue
    print(case, tex.shape, w, h, kw, eye, ns, modes)
    for override in ({}, {"estimator":0}, {"mode":0}, {"max_depth":2000}, {"sample_step":1/512}):
        k2=dict(kw); k2.update(override)
        tr, orc = make_pair(tex, w, h, **k2)
        U, V, W = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, w / h)
        tr.set_camera(eye, U, V, W); orc.set_camera(eye, U, V, W)
        tr.render_accumulate(1, 3)
        mean, m2 = orc.render(3)
        d = (tr.mean()!=mean).any(axis=2)
        print(override, 'equal', np.array_equal(tr.mean(), mean), 'bad px', int(d.sum()), 'counters', tr.counters()==orc.counters.as_dict())
        if d.any():
            ys,xs=np.nonzero(d); print('  first bad', xs[0], ys[0], tr.mean()[ys[0],xs[0]], mean[ys[0],xs[0]])
        tr.close()

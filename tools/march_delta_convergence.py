import sys, numpy as np
sys.path.insert(0,'.')
import deepestscatter_amd as ds
tex = ds.make_procedural_cloud(96)
w=h=48; spp=2048
def run(**kw):
    tr = ds.CloudTracer(tex, width=w, height=h, **kw); tr.render_accumulate(1, spp)
    m = tr.mean()[...,0].astype(np.float64); c = tr.counters(); tr.close()
    return m.mean(), c['scatter_events']/c['paths'], c['density_lookups']/c['paths'], c['depth_capped']
for step in (1/128, 1/256, 1/512, 1/1024, 1/2048, 1/4096):
    print('MARCH step 1/%d' % round(1/step), run(sample_step=step))
print('DELTA', run(estimator=1))
for md in (500, 8000):
    print('max_depth', md, 'MARCH', run(max_depth=md), 'DELTA', run(estimator=1, max_depth=md))

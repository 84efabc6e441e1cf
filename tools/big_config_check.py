import time, numpy as np, sys
sys.path.insert(0,'.')
import deepestscatter_amd as ds
t=time.time(); tex=ds.make_procedural_cloud(1024); print('gen 1024^3', time.time()-t, (tex>0).mean())
b=(tex.reshape(128,8,128,8,128,8).max(axis=(1,3,5))>0).mean(); print('nonempty 8^3 bricks', b)
t=time.time(); tr=ds.CloudTracer(tex,width=2048,height=2048); print('create', time.time()-t)
t=time.time(); tr.render_accumulate(1,4); print('warm 4 spp', time.time()-t)
t=time.time(); tr.render_accumulate(5,32); dt=time.time()-t; print('32 spp', dt, 'Msamples/s', 2048*2048*32/dt/1e6)
c=tr.counters(); print(c, 'lookups/sample', (c['density_lookups']+c['inscatter_lookups'])/c['paths'])
m=tr.mean(); print('finite', np.isfinite(m).all(), 'mean radiance', m[...,0].mean(), 'max', m[...,0].max())
print(tr.kernel_time())
t=time.time(); tr.render_accumulate_async(37,64); tr.render_accumulate_async(101,64); tr.synchronize(); dt=time.time()-t; print('2 x 64 spp enqueued', dt, 'Msamples/s', 2048*2048*128/dt/1e6, 'suspended', tr.debug_suspended())
m2=tr.mean(); print('finite', np.isfinite(m2).all(), 'mean radiance', m2[...,0].mean())

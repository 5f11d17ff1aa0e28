import time, numpy as np, sys, os
sys.path.insert(0, '.')
import ray_tracer_amd as rt
w,h,spp=1920,1080,64
ubo = rt.make_ubo(w,h,spp,25); sph = rt.default_scene()
for thr in (0, 2000, 65536, 262144, 524288, 1048576):
    os.environ['RT_WF_DRAIN_THRESHOLD']=str(thr)
    with rt.Renderer(w,h,mode=rt.MODE_WAVEFRONT,flags=0) as r:
        r.set_params(ubo); r.set_spheres(sph)
        r.render(0,1,1); r.sync()
        for k in range(2):
            t0=time.perf_counter(); r.render(1+k,1,1); r.sync(); t1=time.perf_counter()
            st = r.stats()
            print('wavefront thr', thr, 'ms', st.last_render_ms, 'wall ms', (t1-t0)*1e3, 'rays', st.rays, 'launches', st.last_launches, flush=True)

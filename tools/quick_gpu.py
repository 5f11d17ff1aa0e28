import time, numpy as np, sys, os
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import ray_tracer_amd as rt
w,h,spp=1920,1080,8
ubo = rt.make_ubo(w,h,spp,25); sph = rt.grid_spheres_scene(10000)
for mode,name in ((rt.MODE_MEGAKERNEL,'mega'),(rt.MODE_WAVEFRONT,'wavefront')):
  for flags,fn in ((rt.FLAG_FORCE_BVH,'bvh'),(rt.FLAG_FORCE_BRUTE,'brute')):
    if fn=='brute' and name=='wavefront': continue
    with rt.Renderer(w,h,mode=mode,flags=flags) as r:
        r.set_params(ubo); r.set_spheres(sph)
        r.render(0,1,1); r.sync()
        st=r.stats(); print('build ms', st.last_build_ms, 'height', st.bvh_height, 'nodes', st.bvh_nodes)
        for k in range(2):
            r.render(1+k,1,1); r.sync()
            st1 = r.stats()
            print(name, fn, 'ms', st1.last_render_ms, 'rays/frame', st1.rays/(2+k), 'Grays/s', st1.rays/(2+k)/st1.last_render_ms/1e6, 'launches', st1.last_launches, flush=True)

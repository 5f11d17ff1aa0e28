import time, numpy as np, sys
sys.path.insert(0, '.')
import ray_tracer_amd as rt
w,h,spp=1920,1080,64
ubo = rt.make_ubo(w,h,spp,25); sph = rt.default_scene()
for mode,name in ((rt.MODE_MEGAKERNEL,'regen'),(rt.MODE_MEGAKERNEL_NESTED,'nested')):
  for flags,fn in ((0,'fast'),(rt.FLAG_STRICT_MATH,'strict')):
    with rt.Renderer(w,h,mode=mode,flags=flags) as r:
        r.set_params(ubo); r.set_spheres(sph)
        r.render(0,1,1); r.sync()
        st0 = r.stats()
        for k in range(3):
            r.render(1+k,1,1)
            st = r.stats()
            print(name, fn, 'ms', st.last_render_ms, 'rays', st.rays, flush=True)
        st = r.stats()
        rays_per_frame = st.rays/4
        print(name, fn, 'Grays/s', rays_per_frame/ (st.last_render_ms*1e-3)/1e9, flush=True)

import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, ray_tracer_amd as rt
w,h=3840,2160
sph=rt.grid_spheres_scene(10000)
for tag,ior in (("approx",None),("exact+ext",np.r_[np.zeros(9999,np.float32),np.float32(1.5)])):
    with rt.Renderer(w,h) as r:
        r.set_params(rt.make_ubo(w,h,32,25)); r.set_spheres(sph)
        if ior is not None: r.set_dielectric(ior)
        r.render(0,8,1); r.stats()
        t=[]
        for f in (1,2):
            r.render(8*f,8,1); t.append(r.stats().last_render_ms)
        print(tag, t, flush=True)

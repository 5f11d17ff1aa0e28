import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, ray_tracer_amd as rt, _oracle as o
for (w,h,vp) in ((800,608,(800,600)),(1920,1080,None)):
    ubo = rt.make_ubo(w,h,1,1,diverge=0.0, viewport=vp)
    sph = rt.default_scene()
    want = o.render(ubo, sph, w, h, threads=16)
    with rt.Renderer(w,h) as r:
        r.set_params(ubo); r.set_spheres(sph); r.render(0,1,1); ids = r.read_primary_ids()
    print(w,h,'primary-id mismatches vs oracle:', int((ids!=want['ids']).sum()))

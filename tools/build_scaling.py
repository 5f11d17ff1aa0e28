#!/usr/bin/env python3
"""LBVH build time and traversal rate versus sphere count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
w, h, spp = 1920, 1080, 4
ubo = rt.make_ubo(w, h, spp, 8)
for n in (1000, 10000, 100000, 1000000):
    sph = rt.grid_spheres_scene(n)
    with rt.Renderer(w, h, flags=rt.FLAG_FORCE_BVH) as r:
        r.set_params(ubo)
        r.set_spheres(sph)           # first build: includes allocations
        t0 = time.perf_counter(); r.set_spheres(sph); t1 = time.perf_counter()
        st = r.stats()
        r.render(0, 1, 1); r.sync(); r.render(1, 1, 1)
        s2 = r.stats()
        print(f"n {n:8d} build_ms(device) {st.last_build_ms:8.3f} set_spheres_wall_ms {(t1-t0)*1e3:8.2f} height {st.bvh_height:3d} "
              f"render_ms {s2.last_render_ms:8.3f} Mrays/s {s2.rays/2/s2.last_render_ms/1e3:9.1f}", flush=True)

#!/usr/bin/env python3
"""Frame times of the default scene (1920x1080, 64 spp) for the library RT_HIP_LIB names."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
ubo = rt.make_ubo(w, h, 64, 25)
with rt.Renderer(w, h) as r:
    r.set_params(ubo); r.set_spheres(rt.default_scene())
    r.render(0, 1, 1); r.stats()
    t = []
    for f in range(1, 41):
        r.render(f, 1, 1); t.append(r.stats().last_render_ms)
    late = sorted(t[16:])   # the tile order has settled by then (refreshed after launches 1, 2, 4, 8, 16)
    print(os.environ.get("RT_HIP_LIB", "current").split("/")[-2:][0], f"min {min(t):.3f} med(17..40) {late[len(late)//2]:.3f} mean(17..40) {sum(late)/len(late):.3f}", flush=True)

#!/usr/bin/env python3
"""Time of a context's first launches (default scene 1920x1080 64 spp; 10k spheres 8 spp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
w, h = 1920, 1080
for name, spp, sph in (("default", 64, rt.default_scene()), ("10k spheres", 8, rt.grid_spheres_scene(10000)),
                       ("default, second context", 64, rt.default_scene()), ("10k spheres, second context", 8, rt.grid_spheres_scene(10000))):
    with rt.Renderer(w, h) as r:
        r.set_params(rt.make_ubo(w, h, spp, 25)); r.set_spheres(sph)
        t = []
        for f in range(4):
            r.render(f, 1, 1); t.append(r.stats().last_render_ms)
        print(name, "launches 1-4:", " ".join(f"{x:.3f}" for x in t), "ms", flush=True)

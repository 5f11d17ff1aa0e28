#!/usr/bin/env python3
"""Plain against batched regen loop over image sizes (RT_REGEN_BATCH_TILES=0 forces the batched loop,
a huge value the plain one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
for w, h in ((1280, 720), (1920, 1080), (2560, 1440), (3200, 1800), (3840, 2160)):
    with rt.Renderer(w, h) as r:
        r.set_params(rt.make_ubo(w, h, 64, 25)); r.set_spheres(rt.default_scene())
        r.render(0, 1, 1); r.stats(); t = []
        for f in range(1, 8):
            r.render(f, 1, 1); t.append(r.stats().last_render_ms)
        print(os.environ.get("RT_REGEN_BATCH_TILES", "default"), f"{w}x{h} tiles {((w+15)//16)*((h+15)//16)} med {sorted(t)[3]:.3f} min {min(t):.3f}", flush=True)

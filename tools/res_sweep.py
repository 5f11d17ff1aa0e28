#!/usr/bin/env python3
"""Rays/s of the default scene against image size and frames per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
cases = [(960, 540, 64, 1), (1920, 1080, 64, 1), (1920, 1080, 64, 4), (1920, 1080, 16, 1), (1920, 1080, 256, 1),
         (3840, 2160, 64, 1), (7680, 4320, 64, 1)]
for w, h, spp, frames in cases:
    ubo = rt.make_ubo(w, h, spp, 25)
    with rt.Renderer(w, h) as r:
        r.set_params(ubo); r.set_spheres(rt.default_scene())
        r.render(50, 1, 1); r.sync(); s0 = r.stats()
        best = None
        for k in range(3):
            r.render(k * frames, frames, 1); r.sync(); s1 = r.stats()
            g = (s1.rays - s0.rays) / s1.last_render_ms / 1e6
            best = max(best or 0, g); s0 = s1
        print(f"{w}x{h} spp {spp} frames/launch {frames}: {best:.1f} Grays/s  last {s1.last_render_ms:.2f} ms", flush=True)

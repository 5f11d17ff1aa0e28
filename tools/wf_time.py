#!/usr/bin/env python3
"""Wavefront frame time, default scene 1920x1080 64 spp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
w, h = 1920, 1080
with rt.Renderer(w, h, mode=rt.MODE_WAVEFRONT) as r:
    r.set_params(rt.make_ubo(w, h, 64, 25)); r.set_spheres(rt.default_scene())
    r.render(0, 1, 1); r.stats(); t = []
    for f in range(1, 5):
        r.render(f, 1, 1); t.append(r.stats().last_render_ms)
    print(os.environ.get("RT_HIP_LIB", "current").split("/")[-2:][0], " ".join(f"{x:.2f}" for x in t), flush=True)

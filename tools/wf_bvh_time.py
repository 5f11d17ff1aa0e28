#!/usr/bin/env python3
"""Wavefront on the 10k-sphere scene, 1920x1080, 8 spp (per-kernel split under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
w, h = 1920, 1080
for mode, name in ((rt.MODE_WAVEFRONT, "wavefront"), (rt.MODE_MEGAKERNEL, "megakernel")):
    with rt.Renderer(w, h, mode=mode) as r:
        r.set_params(rt.make_ubo(w, h, 8, 25)); r.set_spheres(rt.grid_spheres_scene(10000))
        r.render(0, 1, 1); r.stats(); t = []
        for f in range(1, 4):
            r.render(f, 1, 1); t.append(r.stats().last_render_ms)
        print(name, " ".join(f"{x:.2f}" for x in t), "launches", r.stats().last_launches, flush=True)

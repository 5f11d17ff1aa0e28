#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 counter passes: N frames of the benchmark
configuration (default scene, 1920x1080, 64 spp) or of the 10k-sphere scene.

    rocprofv3 --pmc <counters> --output-format csv -d out -- python3 tools/prof_render.py [mode] [scene] [frames]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt  # noqa: E402

mode = {"megakernel": rt.MODE_MEGAKERNEL, "wavefront": rt.MODE_WAVEFRONT,
        "nested": rt.MODE_MEGAKERNEL_NESTED}[sys.argv[1] if len(sys.argv) > 1 else "megakernel"]
scene = sys.argv[2] if len(sys.argv) > 2 else "default"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if scene == "default":
    w, h, spp = 1920, 1080, 64
    sph = rt.default_scene()
else:
    w, h, spp = 1920, 1080, 8
    sph = rt.grid_spheres_scene(10000)
ubo = rt.make_ubo(w, h, spp, 25)
with rt.Renderer(w, h, mode=mode) as r:
    r.set_params(ubo)
    r.set_spheres(sph)
    for f in range(frames):
        r.render(f, 1, 1)
    r.sync()
    st = r.stats()
    print(f"frames {frames} rays {st.rays} last_ms {st.last_render_ms:.3f}")

#!/usr/bin/env python3
"""Small fixed workloads for rocprofv3 counter passes.

    rocprofv3 --pmc <counters> --output-format csv -d out -- python3 tools/prof_render.py [mode] [scene] [frames] [WxH]
scene: default (3 spheres, 1920x1080, 64 spp) | grid (10k spheres, 8 spp) | terrain (1M triangles, 8 spp, 8 bounces)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt  # noqa: E402

mode = {"megakernel": rt.MODE_MEGAKERNEL, "wavefront": rt.MODE_WAVEFRONT,
        "nested": rt.MODE_MEGAKERNEL_NESTED}[sys.argv[1] if len(sys.argv) > 1 else "megakernel"]
scene = sys.argv[2] if len(sys.argv) > 2 else "default"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
flags = rt.FLAG_COUNT_TRAVERSAL if os.environ.get("RT_COUNT") else 0
w, h, spp, bounces = 1920, 1080, 8, 25
if len(sys.argv) > 4:
    w, h = (int(v) for v in sys.argv[4].split("x"))
mesh = None
if scene == "default":
    spp, sph = 64, rt.default_scene()
elif scene == "grid":
    sph = rt.grid_spheres_scene(10000)
else:
    bounces, sph = 8, rt.default_scene()[:2]
    mesh = (*rt.terrain_mesh(709), rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15))
ubo = rt.make_ubo(w, h, spp, bounces)
with rt.Renderer(w, h, mode=mode, flags=flags) as r:
    r.set_params(ubo)
    r.set_spheres(sph)
    if mesh:
        r.set_mesh(*mesh)
    for f in range(frames):
        r.render(f, 1, 1)
    r.sync()
    st = r.stats()
    print(f"frames {frames} rays {st.rays} last_ms {st.last_render_ms:.3f} nodes/ray "
          f"{st.bvh_nodes_visited / max(st.rays, 1):.2f} prims/ray {st.bvh_spheres_tested / max(st.rays, 1):.2f}")

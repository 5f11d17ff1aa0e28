#!/usr/bin/env python3
"""Time of the first N tiles of the heaviest-first order launched alone (needs a library built with
-DRT_DEBUG_TILE_LIMIT, named by RT_HIP_LIB): N = 1 is the chain of the slowest pixel on an
otherwise idle device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
scene = os.environ.get("LONE_SCENE", "default")   # default | grid (10k spheres, 8 spp) | terrain (1M triangles, 8 spp, 8 bounces)
with rt.Renderer(w, h) as r:
    if scene == "default":
        r.set_params(rt.make_ubo(w, h, 64, 25)); r.set_spheres(rt.default_scene())
    elif scene == "grid":
        r.set_params(rt.make_ubo(w, h, 8, 25)); r.set_spheres(rt.grid_spheres_scene(10000))
    else:
        r.set_params(rt.make_ubo(w, h, 8, 8)); r.set_spheres(rt.default_scene()[:2])
        r.set_mesh(*rt.terrain_mesh(709), rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15))
    for f in range(20):   # the order settles; later refreshes come every 16th launch
        r.render(0, 1, 1)
    r.stats()
    for n in [int(x) for x in os.environ.get('LONE_TILES', '1,2,4,16,64,256,512,1024,1536,3072,6144,0').split(',')]:
        os.environ["RT_DEBUG_TILE_LIMIT"] = str(n)
        t = []
        for k in range(5):
            r.render(0, 1, 1); t.append(r.stats().last_render_ms)
        print(f"tiles {n or 'all'}: {min(t):.3f} ms", flush=True)

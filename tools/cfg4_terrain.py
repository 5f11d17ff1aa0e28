#!/usr/bin/env python3
"""BASELINE.json configs[3]: procedural 1M-triangle terrain, LBVH rebuilt every frame,
at most 8 bounces, one GPU.  Reports build and trace separately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
w, h, spp, frames = 1920, 1080, 8, 6
n = int(sys.argv[1]) if len(sys.argv) > 1 else 709
ubo = rt.make_ubo(w, h, spp, 8)
mat = rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15)
sph = rt.default_scene()[:2]
with rt.Renderer(w, h) as r:
    r.set_params(ubo); r.set_spheres(sph)
    rows = []
    for f in range(frames):
        v, i = rt.terrain_mesh(n, phase=0.15 * f)
        t0 = time.perf_counter(); r.set_mesh(v, i, mat); t1 = time.perf_counter()
        st0 = r.stats()
        r.render(f, 1, 1); r.sync()
        st = r.stats()
        rows.append((st.last_build_ms, (t1 - t0) * 1e3, st.last_render_ms, st.rays - st0.rays, st.bvh_height))
        if f == frames - 1:
            rt.write_png(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/gpurun_out/cfg4_terrain.png", r.read_rgba8()[::2, ::2].copy())
    for k, (b, wall, ms, rays, ht) in enumerate(rows):
        print(f"frame {k} tris {i.shape[0]} build_ms {b:.3f} set_mesh_wall_ms {wall:.1f} trace_ms {ms:.3f} "
              f"rays {rays} Mrays/s {rays / ms / 1e3:.0f} height {ht}", flush=True)

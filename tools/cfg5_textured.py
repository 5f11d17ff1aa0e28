#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: textured material, 16 progressive frames at
7680x4320.  The reference's own settings otherwise (50 spp, 25 bounces, default scene); the
brick texture is bound to the ground sphere.  With N GPUs each renders 16/N of the frames
(frame split) and the accumulation images are summed once; here one GPU renders its share
for N = 8 (2 frames) and, for the record, all 16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt

w, h = 7680, 4320
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ubo = rt.make_ubo(w, h, spp, 25)
sph = rt.default_scene()
tex = rt.brick_texture(1024, 1024)
for mode, name in ((rt.MODE_MEGAKERNEL, "megakernel"),):
    with rt.Renderer(w, h, mode=mode) as r:
        r.set_params(ubo)
        r.set_spheres(sph)
        for textured in (False, True):
            r.set_texture(tex if textured else None, 2, 1, rt.TEX_MAP_PLANAR_XZ, (0.25, 0.25))
            r.clear()
            r.render(100, 1, 1); r.sync()                      # warm-up
            r.clear()
            s0 = r.stats()
            t0 = time.perf_counter(); r.render(0, 2, 8); r.sync(); t1 = time.perf_counter()
            s1 = r.stats()
            print(f"{name} textured={textured} share-of-8 (frames 0,8): {s1.last_render_ms:.1f} ms, "
                  f"{(s1.rays - s0.rays) / s1.last_render_ms / 1e3:.0f} Mrays/s", flush=True)
        r.clear()
        s0 = r.stats()
        r.render(0, 16, 1); r.sync()
        s1 = r.stats()
        print(f"{name} textured=True all 16 frames: {s1.last_render_ms:.1f} ms, rays {s1.rays - s0.rays}, "
              f"{(s1.rays - s0.rays) / s1.last_render_ms / 1e3:.0f} Mrays/s", flush=True)
        img = r.read_rgba8()
        rt.write_png(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/gpurun_out/cfg5_textured.png", img[::8, ::8].copy())

#!/usr/bin/env python3
"""The wavefront pipeline on LBVH scenes with and without ray regrouping (RT_WF_REGROUP, read when a
context is created): frame time, and that the pictures are the same bits.

    python tools/wf_regroup.py [terrain|grid] [regroup 0|1] [drain threshold, default: the library's]
With no regroup argument both settings run in one process and are compared.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ray_tracer_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "terrain"
only = sys.argv[2] if len(sys.argv) > 2 else None
if len(sys.argv) > 3:
    os.environ["RT_WF_DRAIN_THRESHOLD"] = sys.argv[3]
w, h = 1920, 1080


def run(regroup, mode=rt.MODE_WAVEFRONT):
    os.environ["RT_WF_REGROUP"] = "1" if regroup else "0"
    with rt.Renderer(w, h, mode=mode) as r:
        if scene == "terrain":
            r.set_params(rt.make_ubo(w, h, 8, 8))
            r.set_spheres(rt.default_scene()[:2])
            v, i = rt.terrain_mesh(709, phase=0.0)
            r.set_mesh(v, i, rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15))
        else:
            r.set_params(rt.make_ubo(w, h, 8, 25))
            sph, ior = rt.grid_spheres_scene_glass(10000)
            r.set_spheres(sph)
            r.set_dielectric(ior)
        r.render(0, 1, 1)
        r.stats()
        t = []
        for f in range(1, 5):
            r.render(f, 1, 1)
            t.append(r.stats().last_render_ms)
        return sorted(t)[len(t) // 2], r.read_accum(), int(r.stats().rays), r.stats().last_launches


if only is not None:
    ms, _, rays, launches = run(only == "1")
    print(f"{scene} regroup={only}: {ms:.3f} ms per frame, {launches} launches")
else:
    mega, acc_m, rays_m, _ = run(False, rt.MODE_MEGAKERNEL)
    off, acc0, rays0, l0 = run(False)
    on, acc1, rays1, l1 = run(True)
    same = np.array_equal(np.nan_to_num(acc0).view(np.uint32), np.nan_to_num(acc1).view(np.uint32)) and rays0 == rays1
    print(f"{scene} {w}x{h}: megakernel {mega:.3f} ms; wavefront {off:.3f} ms ({l0} launches); "
          f"regrouped {on:.3f} ms ({l1} launches); same bits: {same}")

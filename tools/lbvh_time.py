#!/usr/bin/env python3
"""Frame times of the LBVH workloads for the library RT_HIP_LIB names (A/B runs of kernel variants).

    python tools/lbvh_time.py [grid|glass|terrain|grid4k|glass4k]...
grid / glass: 10k spheres (coated / glass third class) 1920x1080 8 spp; terrain: 1M triangles 1920x1080
8 spp 8 bounces; *4k: the 3840x2160 8 x 32 spp batch of BASELINE configs[2].  Prints the median of
5 launches and the two traversal tallies (meaningful in -DRT_SCHED_STATS builds: lanes served and
64 x times the step ran)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt

for scene in sys.argv[1:] or ["grid", "terrain"]:
    big = scene.endswith("4k")
    w, h = (3840, 2160) if big else (1920, 1080)
    spp, bounces, frames = (32, 25, 8) if big else (8, 25, 1)
    ior = mesh = None
    if scene.startswith("grid"):
        sph = rt.grid_spheres_scene(10000)
    elif scene.startswith("glass"):
        sph, ior = rt.grid_spheres_scene_glass(10000)
    else:
        bounces, sph = 8, rt.default_scene()[:2]
        mesh = (*rt.terrain_mesh(709), rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15))
    with rt.Renderer(w, h) as r:
        r.set_params(rt.make_ubo(w, h, spp, bounces))
        r.set_spheres(sph)
        if ior is not None:
            r.set_dielectric(ior)
        if mesh:
            r.set_mesh(*mesh)
        r.render(0, frames, 1)
        s0 = r.stats()
        t = []
        for f in range(1, 3 if big else 6):
            r.render(f * frames, frames, 1)
            t.append(r.stats().last_render_ms)
        s1 = r.stats()
        n = len(t)
        rays = (s1.rays - s0.rays) / n
        extra = ""
        if s1.bvh_spheres_tested:
            extra = f" lanes/step {64.0 * s1.bvh_nodes_visited / s1.bvh_spheres_tested:.1f} steps/ray {(s1.bvh_spheres_tested - s0.bvh_spheres_tested) / 64.0 / n / rays * 64:.2f}"
        lib = os.environ.get("RT_HIP_LIB", "current").split("/")[-2:][0]
        print(f"{lib} {scene}: med {sorted(t)[n // 2]:.3f} ms min {min(t):.3f} Grays/s {rays / sorted(t)[n // 2] / 1e6:.2f} "
              f"build {s1.last_build_ms:.3f} height {s1.bvh_height}{extra}", flush=True)

#!/usr/bin/env python3
"""Fast-math invariant of one case of tests/fuzz_parity.py, taken apart: python tools/fuzz_case_probe.py <seed> <case>
Renders the case with brute force and with the LBVH (megakernel, fast math), names the pixels whose
accumulation values or primary ids differ, and repeats the comparison without the glass table and
without the mesh."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fuzz_parity as fz
import ray_tracer_amd as rt

seed, case = int(sys.argv[1]), int(sys.argv[2])
c = fz.random_case(np.random.default_rng([seed, case]))

def render(search, glass=True, mesh=True, bounces=None, frames=None):
    ubo = c["ubo"].copy()
    if bounces is not None:
        ubo[41] = bounces
    with rt.Renderer(c["w"], c["h"], mode=rt.MODE_MEGAKERNEL, flags=search) as r:
        r.set_params(ubo)
        r.set_spheres(c["sph"])
        if glass and c["ior"] is not None:
            r.set_dielectric(c["ior"])
        if mesh and c["mesh"] is not None:
            r.set_mesh(*c["mesh"])
        for k in range(c["calls"]):
            r.render(c["first"] + k * c["frames"], c["frames"], 1)
        r.sync()
        acc = r.read_accum()
        return np.where(np.isnan(acc), np.float32(0), acc).view(np.uint32).copy(), r.read_primary_ids().copy(), int(r.stats().rays), r.stats().bvh_height

for glass, mesh in ((True, True), (False, True), (True, False), (False, False)):
    for bounces in (None, 1):
        a = render(rt.FLAG_FORCE_BRUTE, glass, mesh, bounces)
        b = render(rt.FLAG_FORCE_BVH, glass, mesh, bounces)
        d = (a[0] != b[0]).any(axis=-1)
        ys, xs = np.nonzero(d.reshape(c["h"], c["w"]))
        di = np.nonzero((a[1] != b[1]).reshape(-1))[0]
        print(f"glass {glass} mesh {mesh} bounces {bounces or int(c['ubo'][41])}: accum differs at {list(zip(xs.tolist(), ys.tolist()))[:8]} "
              f"ids differ at {[(int(i % c['w']), int(i // c['w']), int(a[1].reshape(-1)[i]), int(b[1].reshape(-1)[i])) for i in di[:8]]} "
              f"rays {a[2]} / {b[2]} height {b[3]}", flush=True)
print("spheres", c["sph"].shape[0], "triangles", 0 if c["mesh"] is None else c["mesh"][1].shape[0], "spp", int(c["ubo"][40]), "frames", c["frames"], "calls", c["calls"])

#!/usr/bin/env python3
"""LBVH traversal tallies for the 10k-sphere scene (counting build of the traversal)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
w, h, spp = 1920, 1080, 8
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ubo = rt.make_ubo(w, h, spp, 25)
sph = rt.grid_spheres_scene(n)
with rt.Renderer(w, h, flags=rt.FLAG_COUNT_TRAVERSAL) as r:
    r.set_params(ubo); r.set_spheres(sph)
    r.render(0, 1, 1); r.sync()
    st = r.stats()
    print(f"spheres {n} rays {st.rays} nodes/ray {st.bvh_nodes_visited / st.rays:.2f} "
          f"spheres/ray {st.bvh_spheres_tested / st.rays:.2f} height {st.bvh_height} "
          f"bytes/ray {(64 * st.bvh_nodes_visited + 16 * st.bvh_spheres_tested) / st.rays:.0f}")

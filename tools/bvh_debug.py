#!/usr/bin/env python3
"""Prints what the LBVH build reports (RT_BVH_DEBUG=1) for the terrain and the 10k-sphere scene."""
import os, sys
os.environ["RT_BVH_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracer_amd as rt
with rt.Renderer(256, 256) as r:
    r.set_params(rt.make_ubo(256, 256, 1, 1))
    r.set_spheres(rt.default_scene()[:2])
    v, i = rt.terrain_mesh(709, phase=0.0)
    r.set_mesh(v, i, rt.matte(0.5, 0.5, 0.5))
    print("terrain build ms", r.stats().last_build_ms)
    r.set_mesh(v, i, rt.matte(0.5, 0.5, 0.5))
    print("terrain build ms", r.stats().last_build_ms)
with rt.Renderer(256, 256) as r:
    r.set_params(rt.make_ubo(256, 256, 1, 1))
    r.set_spheres(rt.grid_spheres_scene(10000))
    print("grid build ms", r.stats().last_build_ms)

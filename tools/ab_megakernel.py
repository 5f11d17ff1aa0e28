#!/usr/bin/env python3
"""A/B timing of library variants on the benchmark frame, interleaved rounds in separate
processes per variant is avoided: each variant runs in its own process (one library per
process), rounds alternate.  Usage: ab_megakernel.py name=path.so ... [--mode megakernel]"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import ray_tracer_amd as rt
mode = {"megakernel": rt.MODE_MEGAKERNEL, "wavefront": rt.MODE_WAVEFRONT, "nested": rt.MODE_MEGAKERNEL_NESTED}[sys.argv[1]]
scene = sys.argv[2]
if scene == "default":
    w, h, spp, sph = 1920, 1080, 64, rt.default_scene()
else:
    w, h, spp, sph = 1920, 1080, 8, rt.grid_spheres_scene(10000)
ubo = rt.make_ubo(w, h, spp, 25)
with rt.Renderer(w, h, mode=mode) as r:
    r.set_params(ubo); r.set_spheres(sph)
    r.render(0, 1, 1); r.sync(); r.stats()
    ms = []
    for f in range(1, 8):
        r.render(f, 1, 1); ms.append(r.stats().last_render_ms)
    ms.sort()
    print(f"{ms[len(ms)//2]:.4f} {ms[0]:.4f}")
'''

def main():
    mode, scene = "megakernel", "default"
    variants = []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--mode": mode = args.pop(0)
        elif a == "--scene": scene = args.pop(0)
        else:
            name, path = a.split("=", 1)
            variants.append((name, os.path.abspath(path)))
    for rnd in range(3):
        for name, path in variants:
            env = dict(os.environ, RT_HIP_LIB=path)
            out = subprocess.run([sys.executable, "-c", CHILD, mode, scene], env=env, capture_output=True, text=True, timeout=300)
            print(f"round {rnd} {name:24s} median/min ms: {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}", flush=True)

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Shrinks a case of tests/fuzz_parity.py in which the LBVH search and the brute-force search
of the library disagree (strict math, nested megakernel), then logs the oracle's path segments of
the first differing pixel.

    python tests/fuzz_reduce.py <seed> <case>

Needs a GPU.  Finds the differing pixels, the first frame, sample count and bounce limit at which
they differ, drops triangles and spheres one at a time while the difference stays, prints what is
left, and saves it to gpurun_out/repro_case.npz.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _oracle as oracle  # noqa: E402
import fuzz_parity as fz  # noqa: E402
import ray_tracer_amd as rt  # noqa: E402


def render(c, search, ubo=None, first=None, frames=None):
    with rt.Renderer(c["w"], c["h"], mode=rt.MODE_MEGAKERNEL_NESTED, flags=rt.FLAG_STRICT_MATH | search) as r:
        r.set_params(c["ubo"] if ubo is None else ubo)
        r.set_spheres(c["sph"])
        if c["ior"] is not None:
            r.set_dielectric(c["ior"])
        if c["mesh"] is not None:
            r.set_mesh(*c["mesh"])
        r.render(c["first"] if first is None else first, c["frames"] if frames is None else frames, 1)
        r.sync()
        return r.read_accum().copy(), r.read_primary_ids().copy()


def differs(c, **kw):
    a, ia = render(c, rt.FLAG_FORCE_BRUTE, **kw)
    b, ib = render(c, rt.FLAG_FORCE_BVH, **kw)
    return (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1) | (ia != ib)


def main():
    seed, i = int(sys.argv[1]), int(sys.argv[2])
    c = fz.random_case(np.random.default_rng([seed, i]))
    c["tex"] = None
    d = differs(c)
    print("differing pixels (y, x)", np.argwhere(d).tolist(), flush=True)
    if not d.any():
        print("brute force and LBVH agree without the texture: not a search difference")
        return 0
    frame = c["first"]
    for f in range(c["frames"]):
        if differs(c, first=c["first"] + f, frames=1).any():
            frame = c["first"] + f
            break
    print("frame", frame, flush=True)
    for name, slot in (("bounce limit", 41), ("samples", 40)):
        for k in range(1, int(c["ubo"][slot]) + 1):
            u = c["ubo"].copy()
            u[slot] = k
            if differs(c, ubo=u, first=frame, frames=1).any():
                print("smallest", name, "that differs:", k, flush=True)
                break
    if c["mesh"] is not None:
        v, t, m = c["mesh"]
        t = t.copy()
        k = 0
        while k < len(t) and len(t) > 1:
            t2 = np.delete(t, k, axis=0)
            c2 = dict(c)
            c2["mesh"] = (v, t2, m)
            if differs(c2, first=frame, frames=1).any():
                t = t2
            else:
                k += 1
        c["mesh"] = (v, t, m)
        print("triangles left", t.tolist(), flush=True)
    k = len(c["sph"]) - 1
    while k >= 0 and len(c["sph"]) > 1:
        c2 = dict(c)
        c2["sph"] = np.delete(c["sph"], k, axis=0)
        c2["ior"] = None if c["ior"] is None else np.delete(c["ior"], k)
        if differs(c2, first=frame, frames=1).any():
            c = c2
        k -= 1
    np.set_printoptions(precision=9, linewidth=200)
    print("spheres left (centre, radius):")
    print(c["sph"][:, :4])
    print("ior", None if c["ior"] is None else c["ior"].tolist())
    if c["mesh"] is not None:
        print("triangle vertices:")
        for tri in c["mesh"][1]:
            print(c["mesh"][0][tri].tolist())
    d = differs(c, first=frame, frames=1)
    pix = np.argwhere(d)
    print("differing pixels of the reduced case", pix.tolist(), flush=True)
    a, ia = render(c, rt.FLAG_FORCE_BRUTE, first=frame, frames=1)
    b, ib = render(c, rt.FLAG_FORCE_BVH, first=frame, frames=1)
    for y, x in pix[:4]:
        print(f"pixel ({y},{x}): brute id {ia[y, x]} accum {a[y, x]}  lbvh id {ib[y, x]} accum {b[y, x]}")
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/repro_case.npz", ubo=c["ubo"], sph=c["sph"], ior=np.zeros(0) if c["ior"] is None else c["ior"],
             v=np.zeros((0, 3)) if c["mesh"] is None else c["mesh"][0],
             t=np.zeros((0, 3), dtype=np.uint32) if c["mesh"] is None else c["mesh"][1],
             m=np.zeros(12) if c["mesh"] is None else c["mesh"][2], frame=frame, w=c["w"], h=c["h"], pix=pix)
    # the oracle's segments of the first differing pixel
    if len(pix):
        y, x = pix[0]
        buf = np.zeros((4096, 8), dtype=np.float32)
        oracle.L.rto_debug_log.argtypes = [C.c_void_p, C.c_uint32]
        oracle.L.rto_debug_log_count.restype = C.c_uint32
        oracle.L.rto_debug_log(buf.ctypes.data_as(C.c_void_p), 4096)
        oracle.render(c["ubo"], c["sph"], c["w"], c["h"], first_frame=frame, n_frames=1, mesh=c["mesh"], ior=c["ior"],
                      pixels=np.array([y * c["w"] + x], dtype=np.uint32), threads=1)
        n = oracle.L.rto_debug_log_count()
        oracle.L.rto_debug_log(None, 0)
        for k in range(min(n, 64)):
            r = buf[k]
            print(f"  segment {k}: o {r[0:3]} d {r[3:6]} hit {int(r[6])} dst {r[7]:.9g}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

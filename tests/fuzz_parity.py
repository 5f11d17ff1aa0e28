#!/usr/bin/env python3
"""Randomised strict-mode parity runs: HIP kernels (through the C ABI) against the CPU oracle.

    python tests/fuzz_parity.py [cases] [seed] [first case]   (FUZZ_SECONDS=N: stop cleanly after N seconds; FUZZ_BIG=1: larger images, ordered mixed launches with sparse waves;
                                                       FUZZ_DEEP=1: thousands of primitives, brute force against LBVH)

Not collected by pytest (a long-running hunt for rare mismatches, run by hand on a GPU box).
Every case draws a scene (1 to 60 spheres of random size and material, some of them glass; sometimes
a small random triangle mesh, sometimes a texture), a camera, an image size that is not a multiple
of the tile, sample / bounce / frame counts, a kernel organisation and a search (brute force or
LBVH), renders it in strict-math mode, and demands the oracle's accumulation image, 8-bit image,
primary ids and segment count bit for bit.  The first mismatch stops the run with the case's seed.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _oracle as oracle  # noqa: E402
import ray_tracer_amd as rt  # noqa: E402


def random_case(rng):
    w = int(rng.integers(17, 150))
    h = int(rng.integers(17, 110))
    spp = int(rng.choice([1, 2, 3, 5, 8]))
    bounces = int(rng.choice([1, 2, 4, 7, 12]))
    frames = int(rng.choice([1, 1, 2, 4, 8]))
    if spp * frames >= 32 and rng.random() < 0.5:
        frames = 4   # keep most cases cheap, some with enough samples for the sparse-wave split
    ubo = rt.make_ubo(w, h, spp, bounces, pitch=float(rng.uniform(-0.5, 0.5)), yaw=float(rng.uniform(-3.1, 3.1)),
                      diverge=float(rng.choice([0.0, 1.0, 3.0])), defocus=float(rng.choice([0.0, 0.0, 2.0])),
                      sun_intensity=float(rng.choice([0.0, 2.0])))
    n = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 60, 150]))
    # a third of the cases sit on a lattice (integer centres, radii and vertices, no camera rotation,
    # no divergence): exact ties, rays along axes and through edges and vertices
    lattice = rng.random() < 0.33
    if lattice:
        ubo = rt.make_ubo(w, h, spp, bounces, diverge=0.0, defocus=float(rng.choice([0.0, 0.0, 16.0])),
                          sun_intensity=float(rng.choice([0.0, 2.0])))
    base = rt.default_scene()
    sph = np.zeros((n, 16), dtype=np.float32)
    for i in range(n):
        s = base[i % 3].copy()
        if i == 0 and rng.random() < 0.7:
            s[0:4] = [0.0, -1000.0 - rng.random(), 0.0, 1000.0]          # a ground sphere
        else:
            s[0:3] = rng.uniform(-6.0, 6.0, 3) + np.array([0.0, 1.0, -8.0])
            s[3] = rng.uniform(0.2, 2.0)
            if lattice:
                s[0:3] = np.rint(s[0:3])
                s[3] = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
        s[4:7] = rng.uniform(0.1, 1.0, 3)                                  # colour
        s[7] = float(rng.choice([0.0, 0.5, 1.0]))                          # smoothness
        s[8:11] = rng.uniform(0.0, 1.0, 3)                                 # emission colour
        s[11] = float(rng.choice([0.0, 0.0, 0.0, 3.0]))                    # emission strength
        s[12:15] = rng.uniform(0.5, 1.0, 3)                                # specular colour
        s[15] = float(rng.choice([0.0, 0.2, 1.0]))                         # specular probability
        sph[i] = s
    if n > 1 and rng.random() < 0.15:         # the camera inside a sphere
        sph[1, 0:4] = [0.0, 0.0, 0.0, float(rng.choice([0.5, 3.0]))] if lattice else [0.1, -0.2, 0.3, 2.5]
    if n > 2 and rng.random() < 0.1:          # extremes of size
        sph[2, 0:4] = [float(rng.uniform(-1e5, 1e5)), float(rng.uniform(-1e5, 1e5)), -3e5, 1e5]
        sph[n - 1, 3] = 1e-3
    if n > 3 and rng.random() < 0.3:          # coincident spheres (every hit of them is a tie), a point sphere
        sph[n - 1, 0:4] = sph[n - 2, 0:4]
        sph[n - 3, 3] = 0.0
    ior = None
    if rng.random() < 0.4:
        ior = np.where(rng.random(n) < 0.4, np.float32(1.5), np.float32(0.0)).astype(np.float32)
        ior[0] = 0.0
    mesh = None
    if rng.random() < 0.35:
        nv = int(rng.integers(3, 40))
        v = (rng.uniform(-5.0, 5.0, (nv, 3)) + np.array([0.0, 1.0, -8.0])).astype(np.float32)
        if lattice:
            v = np.rint(v).astype(np.float32)
        t = rng.integers(0, nv, (int(rng.choice([1, 5, 20, 59, 300])), 3)).astype(np.uint32)
        if len(t) > 4 and rng.random() < 0.5:  # the same triangle twice, and once more with its corners rotated
            t[-1] = t[0]
            t[-2] = t[1][[1, 2, 0]]
        mesh = (v, t, rt.matte(0.6, 0.5, 0.4, smoothness=float(rng.choice([0.0, 0.8])),
                               specular_probability=float(rng.choice([0.0, 0.3]))))
    tex = None
    if rng.random() < 0.3:
        tw, th = int(rng.integers(1, 20)), int(rng.integers(1, 20))
        n_all = n + (0 if mesh is None else len(mesh[1]))
        first_prim = int(rng.integers(0, n_all))          # spheres, triangles, or a range across both
        tex = dict(rgba8=rng.integers(0, 256, (th, tw, 4), dtype=np.uint8), first_prim=first_prim,
                   n_prims=int(rng.integers(1, n_all - first_prim + 1)), mapping=int(rng.integers(0, 2)),
                   scale=(float(rng.uniform(0.05, 2.0)), float(rng.uniform(0.05, 2.0))), srgb=bool(rng.integers(0, 2)))
    mode = int(rng.choice([rt.MODE_MEGAKERNEL, rt.MODE_MEGAKERNEL, rt.MODE_MEGAKERNEL_NESTED, rt.MODE_WAVEFRONT]))
    search = int(rng.choice([rt.FLAG_FORCE_BRUTE, rt.FLAG_FORCE_BVH, 0]))
    first = int(rng.integers(0, 1000))
    calls = int(rng.choice([1, 1, 2, 3]))   # several launches: the later ones run tile-ordered forms
    linear = bool(rng.random() < 0.2)
    # what changes between the launches of a case: nothing, the camera, a sphere, or the mesh's vertices
    edits = [str(rng.choice(["none", "none", "camera", "sphere", "mesh"])) for _ in range(calls)]
    edits[0] = "none"
    fast_check = bool(rng.random() < 0.5)
    edit_seed = int(rng.integers(0, 1 << 30))
    if os.environ.get("FUZZ_BIG"):   # larger images, enough samples for the ordered mixed launches (sparse waves)
        w, h = int(rng.integers(200, 420)), int(rng.integers(150, 300))
        ubo = rt.make_ubo(w, h, 8, int(rng.choice([4, 12])), pitch=float(rng.uniform(-0.3, 0.3)),
                          yaw=float(rng.uniform(-0.5, 0.5)), sun_intensity=float(rng.choice([0.0, 2.0])))
        frames, calls, mode = int(rng.choice([4, 8])), 3, rt.MODE_MEGAKERNEL
        edits = ["none"] * calls
        fast_check = False
    return dict(w=w, h=h, ubo=ubo, sph=sph, ior=ior, mesh=mesh, tex=tex, mode=mode, search=search,
                first=first, frames=frames, calls=calls, linear=linear, lattice=lattice, edits=edits, edit_seed=edit_seed, fast_check=fast_check)


def scenes_of(c):
    """(ubo, spheres, mesh) in force at each launch of the case."""
    rng = np.random.default_rng(c.get("edit_seed", 0))
    ubo, sph, mesh = c["ubo"].copy(), c["sph"].copy(), c["mesh"]
    out = []
    for k in range(c["calls"]):
        kind = c.get("edits", ["none"] * c["calls"])[k]
        if kind == "camera":
            spp, bounces = int(ubo[40]), int(ubo[41])
            new = rt.make_ubo(c["w"], c["h"], spp, bounces, pitch=float(rng.uniform(-0.4, 0.4)),
                              yaw=float(rng.uniform(-1.0, 1.0)), diverge=float(ubo[67]), defocus=float(ubo[66]),
                              sun_intensity=float(ubo[65]))
            ubo = new
        elif kind == "sphere":
            sph = sph.copy()
            j = int(rng.integers(0, sph.shape[0]))
            sph[j, 0:3] += rng.uniform(-1.0, 1.0, 3).astype(np.float32)
        elif kind == "mesh" and mesh is not None:
            v = (mesh[0] + rng.uniform(-0.3, 0.3, mesh[0].shape)).astype(np.float32)
            mesh = (v, mesh[1], mesh[2])
        out.append((ubo, sph, mesh, kind))
    return out


def run_case(c):
    flags = rt.FLAG_STRICT_MATH | c["search"] | (rt.FLAG_ACCUM_LINEAR if c.get("linear") else 0)
    amode = oracle.ACCUM_LINEAR if c.get("linear") else oracle.ACCUM_GAMMA
    want = None
    acc = img = None
    scenes = scenes_of(c)
    rays_want = 0
    for k in range(c["calls"]):
        ubo_k, sph_k, mesh_k, _ = scenes[k]
        want = oracle.render(ubo_k, sph_k, c["w"], c["h"], first_frame=c["first"] + k * c["frames"],
                             n_frames=c["frames"], mesh=mesh_k, texture=c["tex"], ior=c["ior"], accum=acc, rgba8=img,
                             accum_mode=amode)
        acc, img = want["accum"], want["rgba8"]
        rays_want += int(want["rays"])
    with rt.Renderer(c["w"], c["h"], mode=c["mode"], flags=flags) as r:
        r.set_params(c["ubo"])
        r.set_spheres(c["sph"])
        if c["ior"] is not None:
            r.set_dielectric(c["ior"])
        if c["mesh"] is not None:
            r.set_mesh(*c["mesh"])
        if c["tex"] is not None:
            t = c["tex"]
            r.set_texture(t["rgba8"], t["first_prim"], t["n_prims"], t["mapping"], t["scale"], t["srgb"])
        for k in range(c["calls"]):
            ubo_k, sph_k, mesh_k, kind = scenes[k]
            if kind == "camera":
                r.set_params(ubo_k)
            elif kind == "sphere":
                r.set_spheres(sph_k)
                if c["ior"] is not None:
                    r.set_dielectric(c["ior"])   # (the table belongs to the sphere set it was given for)
            elif kind == "mesh" and mesh_k is not None:
                r.set_mesh(*mesh_k)
            r.render(c["first"] + k * c["frames"], c["frames"], 1)
        r.sync()
        got = dict(accum=r.read_accum(), rgba8=r.read_rgba8(), ids=r.read_primary_ids(), rays=int(r.stats().rays))
    if c["mesh"] is None and c["tex"] is None and c["ior"] is None and c["calls"] == 1 and c["sph"].shape[0] <= 1024:
        # the entry point that allocates nothing must produce the same bits (it always searches by brute force)
        import torch
        dev = torch.device("cuda", 0)
        d_sph = torch.from_numpy(np.ascontiguousarray(c["sph"])).to(dev)
        d_acc = torch.zeros((c["h"], c["w"], 4), dtype=torch.float32, device=dev)
        d_img = torch.zeros((c["h"], c["w"], 4), dtype=torch.uint8, device=dev)
        d_ids = torch.full((c["h"], c["w"]), -1, dtype=torch.int32, device=dev)
        d_rays = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        rt.render_static(c["w"], c["h"], c["ubo"], d_sph.data_ptr(), c["sph"].shape[0], d_acc.data_ptr(),
                         d_img.data_ptr(), d_ids.data_ptr(), d_rays.data_ptr(), first_frame=c["first"],
                         n_frames=c["frames"], flags=rt.FLAG_STRICT_MATH | (rt.FLAG_ACCUM_LINEAR if c.get("linear") else 0))
        torch.cuda.synchronize()
        sa = d_acc.cpu().numpy().view(np.uint32)
        wa = np.asarray(want["accum"], dtype=np.float32).view(np.uint32)
        both_nan = np.isnan(d_acc.cpu().numpy()) & np.isnan(want["accum"])
        if not np.array_equal(np.where(both_nan, 0, sa), np.where(both_nan, 0, wa)):
            return ["rt_render_static: accum differs"]
        if not np.array_equal(d_ids.cpu().numpy(), want["ids"]) or int(d_rays.item()) != rays_want:
            return ["rt_render_static: ids or rays differ"]
    if c.get("fast_check"):
        # fast math has no oracle, but it has a property: the closest hit does not depend on the order
        # in which primitives are visited, and every organisation runs the same device functions per
        # path, so brute force and LBVH, megakernel, nested and wavefront must all produce the same bits
        ref = None
        for mode in (rt.MODE_MEGAKERNEL, rt.MODE_MEGAKERNEL_NESTED, rt.MODE_WAVEFRONT):
            for search in (rt.FLAG_FORCE_BRUTE, rt.FLAG_FORCE_BVH):
                with rt.Renderer(c["w"], c["h"], mode=mode, flags=search) as r:
                    r.set_params(c["ubo"])
                    r.set_spheres(c["sph"])
                    if c["ior"] is not None:
                        r.set_dielectric(c["ior"])
                    if c["mesh"] is not None:
                        r.set_mesh(*c["mesh"])
                    if c["tex"] is not None:
                        t = c["tex"]
                        r.set_texture(t["rgba8"], t["first_prim"], t["n_prims"], t["mapping"], t["scale"], t["srgb"])
                    for k in range(c["calls"]):
                        r.render(c["first"] + k * c["frames"], c["frames"], 1)
                    r.sync()
                    acc = r.read_accum()
                    cur = (np.where(np.isnan(acc), np.float32(0), acc).view(np.uint32).copy(), r.read_primary_ids().copy(),
                           int(r.stats().rays))
                if ref is None:
                    ref = cur
                elif not (np.array_equal(ref[0], cur[0]) and np.array_equal(ref[1], cur[1]) and ref[2] == cur[2]):
                    return [f"fast math: mode {mode} search {search} differs from megakernel / brute force "
                            f"({int((ref[0] != cur[0]).any(axis=-1).sum())} pixels, rays {ref[2]} vs {cur[2]})"]
    a = np.asarray(got["accum"], dtype=np.float32).view(np.uint32)
    b = np.asarray(want["accum"], dtype=np.float32).view(np.uint32)
    nan_ok = np.isnan(got["accum"]) & np.isnan(want["accum"])   # log(0) draws: NaN on both sides, payload aside
    problems = []
    if not np.array_equal(np.where(nan_ok, 0, a), np.where(nan_ok, 0, b)):
        problems.append(f"accum differs in {int(((a != b) & ~nan_ok).any(axis=-1).sum())} pixels")
    if not np.array_equal(got["rgba8"], want["rgba8"]):
        problems.append(f"rgba8 differs in {int((got['rgba8'] != want['rgba8']).any(axis=-1).sum())} pixels")
    if not np.array_equal(got["ids"], want["ids"]):
        problems.append(f"ids differ in {int((got['ids'] != want['ids']).sum())} pixels")
    if got["rays"] != rays_want:
        problems.append(f"rays {got['rays']} != {rays_want}")
    return problems


def deep_case(rng):
    """FUZZ_DEEP=1: thousands of primitives, too many for the oracle.  The library's brute-force
    search (which the other cases pin to the oracle) against its LBVH search, strict math."""
    w, h = int(rng.integers(40, 100)), int(rng.integers(40, 100))
    ubo = rt.make_ubo(w, h, int(rng.choice([1, 2])), int(rng.choice([2, 5])), yaw=float(rng.uniform(-0.3, 0.3)))
    n = int(rng.choice([300, 1000, 3000, 8000]))
    base = rt.default_scene()
    sph = np.repeat(base[:1], n, axis=0).copy()
    kind = str(rng.choice(["uniform", "clustered", "lattice", "line"]))
    if kind == "uniform":
        sph[:, 0:3] = rng.uniform(-20, 20, (n, 3)) + np.array([0, 0, -30.0])
    elif kind == "clustered":   # many centres in the same Morton cell, some identical
        centres = rng.uniform(-15, 15, (8, 3)) + np.array([0, 0, -30.0])
        sph[:, 0:3] = centres[rng.integers(0, 8, n)] + rng.normal(0, 0.05, (n, 3))
        sph[n // 2:, 0:3] = sph[: n - n // 2, 0:3]
    elif kind == "lattice":
        sph[:, 0:3] = np.rint(rng.uniform(-10, 10, (n, 3))) + np.array([0, 0, -25.0])
    else:                       # a line: the most unbalanced tree Morton codes can give
        sph[:, 0] = np.linspace(-30, 30, n); sph[:, 1] = 0.0; sph[:, 2] = -30.0
    sph[:, 3] = rng.uniform(0.05, 0.6, n)
    sph[:, 4:7] = rng.uniform(0.2, 1.0, (n, 3))
    sph[:, 7] = rng.choice([0.0, 1.0], n)
    sph[:, 15] = rng.choice([0.0, 0.5], n)
    if rng.random() < 0.5:
        sph[0, 0:4] = [0.0, -1030.0, 0.0, 1000.0]
    mesh = None
    if rng.random() < 0.5:
        nv = int(rng.choice([50, 500, 3000]))
        v = (rng.uniform(-15, 15, (nv, 3)) + np.array([0, 0, -30.0])).astype(np.float32)
        nt = int(rng.choice([500, 4000]))
        a = rng.integers(0, nv, nt)
        t = np.stack([a, (a + rng.integers(1, 4, nt)) % nv, (a + rng.integers(4, 9, nt)) % nv], axis=1).astype(np.uint32)
        mesh = (v, t, rt.matte(0.6, 0.5, 0.4, smoothness=0.5, specular_probability=0.3))
    return dict(w=w, h=h, ubo=ubo, sph=sph.astype(np.float32), mesh=mesh, first=int(rng.integers(0, 100)), kind=kind)


def run_deep(c):
    out = []
    for search in (rt.FLAG_FORCE_BRUTE, rt.FLAG_FORCE_BVH):
        with rt.Renderer(c["w"], c["h"], mode=rt.MODE_MEGAKERNEL, flags=rt.FLAG_STRICT_MATH | search) as r:
            r.set_params(c["ubo"])
            r.set_spheres(c["sph"])
            if c["mesh"] is not None:
                r.set_mesh(*c["mesh"])
            r.render(c["first"], 2, 1)
            r.sync()
            acc = r.read_accum()
            out.append((np.where(np.isnan(acc), np.float32(0), acc).view(np.uint32).copy(), r.read_primary_ids().copy(),
                        int(r.stats().rays), r.stats().bvh_height))
    a, b = out
    problems = []
    if not np.array_equal(a[0], b[0]):
        problems.append(f"accum differs in {int((a[0] != b[0]).any(axis=-1).sum())} pixels")
    if not np.array_equal(a[1], b[1]):
        problems.append(f"ids differ in {int((a[1] != b[1]).sum())} pixels")
    if a[2] != b[2]:
        problems.append(f"rays {a[2]} != {b[2]}")
    return problems, b[3]


def main():
    import time
    deadline = time.monotonic() + float(os.environ.get("FUZZ_SECONDS", "1e9"))   # a run ends cleanly at its time budget
    if os.environ.get("FUZZ_DEEP"):
        cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
        for i in range(cases):
            if time.monotonic() > deadline:
                print(f"time budget reached after {i} deep cases, no mismatch", flush=True)
                return 0
            c = deep_case(np.random.default_rng([seed, i, 77]))
            problems, height = run_deep(c)
            tag = (f"deep case {i} (seed {seed}): {c['w']}x{c['h']} spheres {c['sph'].shape[0]} {c['kind']} "
                   f"triangles {0 if c['mesh'] is None else c['mesh'][1].shape[0]} tree height {height}")
            if problems:
                print("MISMATCH", tag, problems, flush=True)
                return 1
            if i % 5 == 0:
                print("ok", tag, flush=True)
        print(f"{cases} deep cases, no mismatch", flush=True)
        return 0
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    start = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    for i in range(start, cases):
        if time.monotonic() > deadline:
            print(f"time budget reached after case {i - 1}, no mismatch", flush=True)
            return 0
        rng = np.random.default_rng([seed, i])
        c = random_case(rng)
        problems = run_case(c)
        tag = (f"case {i} (seed {seed}): {c['w']}x{c['h']} spheres {c['sph'].shape[0]} mode {c['mode']} search {c['search']} "
               f"spp {int(c['ubo'][40])} bounces {int(c['ubo'][41])} frames {c['frames']} x {c['calls']} "
               f"glass {c['ior'] is not None} mesh {c['mesh'] is not None} tex {c['tex'] is not None} "
               f"lattice {c['lattice']} linear {c['linear']} edits {c['edits']}")
        if problems:
            print("MISMATCH", tag, problems, flush=True)
            return 1
        if i % 20 == 0:
            print("ok", tag, flush=True)
    print(f"{cases} cases, no mismatch", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())

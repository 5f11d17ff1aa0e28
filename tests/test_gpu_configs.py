"""Every workload BASELINE.json names, at its stated size, on the GPU.

A full CPU render of these sizes would take minutes to hours, so each case combines
  * the oracle where it is affordable: the whole image when the scene is three spheres and the
    sample count small, otherwise a strided sample of the pixels through the oracle's
    pixel-list entry (pixels are independent, so a listed pixel has the bits it would have in a
    full render), bit for bit in strict-math mode;
  * size-independent properties on the whole frame: the kernel organisations trace identical
    paths, frame splits sum to the sequential image, counts and ranges are what the shader's
    loop structure implies.
"""
import numpy as np
import pytest

import ray_tracer_amd as rt

pytestmark = pytest.mark.gpu

ORACLE_THREADS = 16


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def _strided_pixels(w, h, step_x, step_y, phase=0):
    ys = np.arange(phase % step_y, h, step_y, dtype=np.uint32)
    xs = np.arange(phase % step_x, w, step_x, dtype=np.uint32)
    return (ys[:, None] * np.uint32(w) + xs[None, :]).reshape(-1)


# ---------------------------------------------------------------- configs[0]
@pytest.mark.parametrize("mode", [rt.MODE_MEGAKERNEL, rt.MODE_WAVEFRONT])
def test_config0_default_scene_720p_1spp_whole_image_bit_exact(oracle, mode):
    """configs[0]: the reference's scene, 1280x720, 1 spp, 25 bounces, frame 0 -- the case the
    reference itself would run on a CPU.  Strict math: every output equals the oracle's."""
    w, h = 1280, 720
    ubo = rt.make_ubo(w, h, 1, 25)
    sph = rt.default_scene()
    want = oracle.render(ubo, sph, w, h, threads=ORACLE_THREADS)
    with rt.Renderer(w, h, mode=mode, flags=rt.FLAG_STRICT_MATH) as r:
        r.set_params(ubo)
        r.set_spheres(sph)
        r.render(0, 1, 1)
        assert r.stats().rays == want["rays"]
        assert np.array_equal(r.read_primary_ids(), want["ids"])
        assert np.array_equal(_bits(r.read_accum()), _bits(want["accum"]))
        assert np.array_equal(r.read_rgba8(), want["rgba8"])
    # the fast-math launch that is benchmarked: same ids up to counted silhouette pixels
    with rt.Renderer(w, h, mode=mode) as r:
        r.set_params(ubo)
        r.set_spheres(sph)
        r.render(0, 1, 1)
        assert (r.read_primary_ids() != want["ids"]).mean() <= 1e-4
        assert abs(int(r.stats().rays) - want["rays"]) <= 2e-3 * want["rays"]


# ---------------------------------------------------------------- configs[2]
def test_config2_10k_spheres_4k_primary_ids(oracle):
    """configs[2]: 10,000 spheres (diffuse / metal / glass), LBVH, 3840x2160.  Jitter off, one
    segment per pixel: the LBVH search equals the brute-force search on ALL 8.3 M pixels (GPU
    against GPU, fast and strict), and equals the brute-force ORACLE on every third pixel of
    every third row (922k pixels x 10k spheres on the CPU)."""
    w, h = 3840, 2160
    ubo = rt.make_ubo(w, h, 1, 1, diverge=0.0)
    sph, ior = rt.grid_spheres_scene_glass(10000)
    out = {}
    for name, flags in (("bvh", rt.FLAG_FORCE_BVH), ("brute", rt.FLAG_FORCE_BRUTE),
                        ("bvh_strict", rt.FLAG_FORCE_BVH | rt.FLAG_STRICT_MATH)):
        with rt.Renderer(w, h, flags=flags) as r:
            r.set_params(ubo)
            r.set_spheres(sph)
            r.set_dielectric(ior)
            r.render(0, 1, 1)
            out[name] = r.read_primary_ids()
            assert r.stats().rays == w * h
            assert r.stats().bvh_nodes == (0 if name == "brute" else 9999)
    assert np.array_equal(out["bvh"], out["brute"]) and np.array_equal(out["bvh"], out["bvh_strict"])
    assert len(np.unique(out["bvh"])) > 3000                      # thousands of distinct spheres are visible
    pix = _strided_pixels(w, h, 3, 3)
    want = oracle.render(ubo, sph, w, h, ior=ior, pixels=pix, threads=ORACLE_THREADS)
    assert np.array_equal(out["bvh"].reshape(-1)[pix], want["ids"])


def test_config2_10k_spheres_4k_256spp_properties(oracle):
    """The same scene at the stated sample count: 256 spp as 8 progressive frames of 32 spp
    (the frame split that 1/2/4/8 GPUs share out).  Fast math, the benchmarked launch.
      * megakernel and wavefront trace identical paths: equal ray counts, bit-identical sums;
      * the frame-count channel is 8 everywhere, values finite and non-negative;
      * ranks 0 and 1 of a 2-GPU split (frames 0,2,4,6 and 1,3,5,7) sum to the sequential image
        up to the order of eight float additions;
      * the oracle agrees on a strided sample of pixels (frame 0, strict math, bit for bit)."""
    w, h, spp, frames, bounces = 3840, 2160, 32, 8, 25
    ubo = rt.make_ubo(w, h, spp, bounces)
    sph, ior = rt.grid_spheres_scene_glass(10000)

    def run(mode, flags, first, n, stride):
        with rt.Renderer(w, h, mode=mode, flags=flags) as r:
            r.set_params(ubo)
            r.set_spheres(sph)
            r.set_dielectric(ior)
            r.render(first, n, stride)
            return r.read_accum(), int(r.stats().rays), r.read_primary_ids()

    full, rays, ids = run(rt.MODE_MEGAKERNEL, 0, 0, frames, 1)
    # randomValue returns exactly 0 once in 2^32 draws and the shader takes log() of it
    # (ray_tracing.comp:123): about ten of this image's 4e10 draws poison their pixel with NaN,
    # as they would in the reference (whose rgba8 store turns NaN into 0).  Counted, not hidden.
    bad = ~np.isfinite(full).all(-1)
    print("pixels with a non-finite sum:", int(bad.sum()))
    assert bad.sum() <= 64
    assert (full[~bad] >= 0).all() and (full[..., 3] == float(frames)).all()
    assert w * h * spp * frames <= rays <= w * h * spp * frames * bounces
    even, rays_e, _ = run(rt.MODE_MEGAKERNEL, 0, 0, frames // 2, 2)
    odd, rays_o, _ = run(rt.MODE_MEGAKERNEL, 0, 1, frames // 2, 2)
    assert rays_e + rays_o == rays
    s = even + odd
    assert np.array_equal(s[..., 3], full[..., 3])
    ok = np.isfinite(s).all(-1) & ~bad
    assert np.max(np.abs(s[ok] - full[ok]) / np.maximum(full[ok], 1.0)) <= 8 * 1.2e-7
    del even, odd, s
    wf, rays_wf, ids_wf = run(rt.MODE_WAVEFRONT, 0, 0, 2, 4)      # frames 0 and 4 through the stages
    mk, rays_mk, ids_mk = run(rt.MODE_MEGAKERNEL, 0, 0, 2, 4)
    assert rays_wf == rays_mk and np.array_equal(ids_wf, ids_mk)
    assert np.array_equal(_bits(wf), _bits(mk))
    del wf, mk
    # oracle: frame 0 in strict math on 64 x 36 strided pixels (each 32 spp x up to 25 bounces x 10k spheres)
    pix = _strided_pixels(w, h, 60, 60, phase=30)
    want = oracle.render(ubo, sph, w, h, ior=ior, pixels=pix, threads=ORACLE_THREADS)
    strict, _, ids_s = run(rt.MODE_MEGAKERNEL, rt.FLAG_STRICT_MATH, 0, 1, 1)
    assert np.array_equal(ids_s.reshape(-1)[pix], want["ids"])
    assert np.array_equal(_bits(strict.reshape(-1, 4)[pix]), _bits(want["accum"]))


# ---------------------------------------------------------------- configs[3]
def test_config3_million_triangle_terrain_rebuilt_per_frame(oracle):
    """configs[3]: 1,002,528 triangles, the LBVH rebuilt for a new phase of the terrain every
    frame, at most 8 bounces, 1920x1080.  Each frame: strict math against the brute-force oracle
    on 4,096+ strided pixels (every path of those pixels tests the million triangles on the
    CPU), bit for bit; then the three kernel organisations against each other on the whole
    frame in fast math."""
    import torch
    w, h, spp, bounces = 1920, 1080, 2, 8
    ubo = rt.make_ubo(w, h, spp, bounces)
    sph = rt.default_scene()[:2]
    mat = rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15)
    pix = _strided_pixels(w, h, 30, 15, phase=7)
    assert pix.size >= 4096
    previous = None
    with rt.Renderer(w, h, flags=rt.FLAG_STRICT_MATH) as r:
        r.set_params(ubo)
        r.set_spheres(sph)
        for frame, phase in enumerate((0.0, 0.15)):
            v, i = rt.terrain_mesh(709, phase=phase)
            assert i.shape[0] == 1002528
            if frame == 0:
                r.set_mesh(v, i, mat)                                   # from host memory
            else:                                                       # from device memory
                dv = torch.from_numpy(v).cuda()
                di = torch.from_numpy(i.view(np.int32)).cuda()
                r.set_mesh_device(dv.data_ptr(), v.shape[0], di.data_ptr(), i.shape[0], mat)
            st = r.stats()
            assert st.bvh_nodes == 1002528 + 2 - 1 and 20 <= st.bvh_height <= 64
            r.clear()
            r.render(frame, 1, 1)
            ids, acc = r.read_primary_ids(), r.read_accum()
            want = oracle.render(ubo, sph, w, h, first_frame=frame, mesh=(v, i, mat), pixels=pix,
                                 threads=ORACLE_THREADS)
            assert (want["ids"] >= 2).mean() > 0.3                       # the terrain fills the lower half
            assert np.array_equal(ids.reshape(-1)[pix], want["ids"])
            assert np.array_equal(_bits(acc.reshape(-1, 4)[pix]), _bits(want["accum"]))
            if previous is not None:
                assert (ids != previous).mean() > 0.01                  # the rebuilt tree is of the new surface
            previous = ids
    # whole frame, fast math: megakernel (wave-scheduled), nested loops and wavefront stages agree
    v, i = rt.terrain_mesh(709, phase=0.3)
    ubo8 = rt.make_ubo(w, h, 8, bounces)
    res = []
    for mode in (rt.MODE_MEGAKERNEL, rt.MODE_MEGAKERNEL_NESTED, rt.MODE_WAVEFRONT):
        with rt.Renderer(w, h, mode=mode) as r:
            r.set_params(ubo8)
            r.set_spheres(sph)
            r.set_mesh(v, i, mat)
            r.render(2, 1, 1)
            res.append((r.read_accum(), r.read_primary_ids(), int(r.stats().rays)))
    acc, ids, rays = res[0]
    assert np.isfinite(acc).all() and (acc >= 0).all() and (acc[..., 3] == 1.0).all()
    assert w * h * 8 <= rays <= w * h * 8 * bounces
    for other in res[1:]:
        assert other[2] == rays and np.array_equal(other[1], ids)
        assert np.array_equal(_bits(other[0]), _bits(acc))


# ---------------------------------------------------------------- configs[4]
def test_config4_textured_8k_progressive_frame_pair(oracle):
    """configs[4]: textured material, 16 progressive frames at 7680x4320 split over 8 GPUs: rank
    r renders frames r and r + 8.  One rank's pair (r = 3), the reference's 50 spp and 25
    bounces, bricks on the ground sphere.
      * strict math against the oracle on every 16th pixel of every 16th row (129,600 pixels),
        sums and the 8-bit progressive image bit for bit;
      * fast math (the benchmarked launch): count channel 2, finite, non-negative; rendering
        the pair as two calls equals one call; PSNR against the strict image >= 40 dB."""
    w, h, spp, bounces, rank, world = 7680, 4320, 50, 25, 3, 8
    ubo = rt.make_ubo(w, h, spp, bounces)
    sph = rt.default_scene()
    tex = dict(rgba8=rt.brick_texture(1024, 1024), first_prim=2, n_prims=1, scale=(0.25, 0.25))
    pix = _strided_pixels(w, h, 16, 16, phase=5)
    want = oracle.render(ubo, sph, w, h, rank, 2, world, texture=tex, pixels=pix, threads=ORACLE_THREADS)

    def run(flags, calls):
        with rt.Renderer(w, h, flags=flags) as r:
            r.set_params(ubo)
            r.set_spheres(sph)
            r.set_texture(tex["rgba8"], 2, 1, rt.TEX_MAP_PLANAR_XZ, (0.25, 0.25))
            for first, n, stride in calls:
                r.render(first, n, stride)
            return r.read_accum(), r.read_rgba8(), int(r.stats().rays)

    acc_s, img_s, rays_s = run(rt.FLAG_STRICT_MATH, [(rank, 2, world)])
    assert np.array_equal(_bits(acc_s.reshape(-1, 4)[pix]), _bits(want["accum"]))
    assert np.array_equal(img_s.reshape(-1, 4)[pix], want["rgba8"])
    acc, img, rays = run(0, [(rank, 2, world)])
    bad = ~np.isfinite(acc).all(-1)                 # log(0) draws, see the configs[2] test
    print("pixels with a non-finite sum:", int(bad.sum()))
    assert bad.sum() <= 64
    assert (acc[~bad] >= 0).all() and (acc[..., 3] == 2.0).all()
    assert w * h * spp * 2 <= rays <= w * h * spp * 2 * bounces
    assert abs(rays - rays_s) <= 2e-3 * rays_s
    acc2, img2, rays2 = run(0, [(rank, 1, 1), (rank + world, 1, 1)])
    assert rays2 == rays and np.array_equal(_bits(acc2), _bits(acc)) and np.array_equal(img2, img)
    del acc2, img2
    diff = (acc[::4, ::4, :3].astype(np.float64) - acc_s[::4, ::4, :3])[np.isfinite(acc[::4, ::4]).all(-1) &
                                                                       np.isfinite(acc_s[::4, ::4]).all(-1)]
    mse = float(np.mean(diff ** 2)) / 4.0
    assert 10.0 * np.log10(1.0 / mse) >= 40.0

#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path: Mrays/s, counting every path segment.

Default workload = BASELINE.json configs[1], the configuration the metric is quoted on: the
reference's 3-sphere scene at 1920x1080, 64 samples per pixel per frame, at most 25 bounces,
camera viewport set to the image size.  One step = one frame batch per GPU: zero the float
accumulation image, render into it, (N > 1: sum the N accumulation images onto rank 0 with one
RCCL reduce over xGMI,) resolve to the 8-bit image.  With N GPUs rank r renders frame
step*N + r -- the sample-wise split whose frames carry the shader's own per-frame RNG streams
-- so per-GPU work is fixed (weak scaling) and every step yields an N*64-spp image on rank 0.
For N > 1 each rank keeps two accumulation images: the reduce of step i runs on RCCL's stream
while step i+1 is traced into the other image; all K renders, K reduces and K resolves lie
inside the timed region.

--config N selects another BASELINE.json workload (each prints its own roofline block):
  0  default scene 1280x720, 1 spp                       weak: one frame per GPU per step
  1  default scene 1920x1080, 64 spp (the default)       weak: one frame per GPU per step
  2  10,000 spheres (diffuse / metal / glass), LBVH, 3840x2160, 256 spp = 8 frames x 32 spp
     per step                                            strong: the 8 frames are split over the GPUs
  3  1,002,528-triangle terrain, LBVH rebuilt from device-resident vertices every step,
     1920x1080, 8 spp, 8 bounces                         one GPU (N > 1: independent replicas)
  4  default scene with a brick texture on the ground, 7680x4320, 16 progressive frames x 50
     spp per step                                        strong: the 16 frames are split over the GPUs

Prints ONE JSON line on rank 0.  Launch for N > 1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# peaks from the MI355X micro-architecture guide
PEAK_FP32_VALU_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PEAK_L2_GBS = 34500.0

WORKLOADS = {
    0: dict(name="configs[0]", w=1280, h=720, spp=1, bounces=25, frames=1, scaling="weak", scene="default",
            text="reference default scene (3 spheres, main.cpp:517-538), 1280x720, 1 spp/frame, maxBounce 25, "
                 "1 frame per GPU per step (BASELINE.json configs[0])"),
    1: dict(name="configs[1]", w=1920, h=1080, spp=64, bounces=25, frames=1, scaling="weak", scene="default",
            text="reference default scene (3 spheres, main.cpp:517-538), 1920x1080, 64 spp/frame, maxBounce 25, "
                 "1 frame per GPU per step (BASELINE.json configs[1])"),
    2: dict(name="configs[2]", w=3840, h=2160, spp=32, bounces=25, frames=8, scaling="strong", scene="grid",
            text="10,000 spheres (rt_grid_scene_glass, seed 1337: 60 % diffuse, 25 % metal, 15 % glass ior 1.5), "
                 "GPU LBVH, 3840x2160, 256 spp = 8 frames x 32 spp per step, maxBounce 25, frames split over "
                 "the GPUs (BASELINE.json configs[2])"),
    3: dict(name="configs[3]", w=1920, h=1080, spp=8, bounces=8, frames=1, scaling="weak", scene="terrain",
            text="1,002,528-triangle terrain (709x709 height field) + 2 spheres, vertices resident on the device, "
                 "LBVH rebuilt every step for a new phase of the surface, 1920x1080, 8 spp, maxBounce 8 "
                 "(BASELINE.json configs[3])"),
    4: dict(name="configs[4]", w=7680, h=4320, spp=50, bounces=25, frames=16, scaling="strong", scene="textured",
            text="reference default scene with a 1024x1024 brick texture on the ground sphere, 7680x4320, 16 "
                 "progressive frames x 50 spp per step, maxBounce 25, frames split over the GPUs "
                 "(BASELINE.json configs[4])"),
}


def algorithmic_flops(rays: float, samples: float, n_spheres: int) -> float:
    """FP32 flops the shader's arithmetic needs for this many path segments, counted from
    ray_tracing.comp with a = dot(d,d) hoisted and an FMA as 2 flops:
      per segment        5 + 20 * n_spheres      (ray_tracing.comp:152-156)
      per hit bounce     23 + 100                (hit point, normal: :159-165; scatter: :199-213)
      per escaping path  40                      (getEnvironmentLight, :239-250)
      per sample         30                      (primary ray, :86-93,224-237)
    hits = rays - samples (every path ends in exactly one miss; paths cut by the bounce
    limit are ~1e-4 of the total and are counted as misses)."""
    hits = max(rays - samples, 0.0)
    return rays * (5 + 20 * n_spheres) + hits * 123.0 + samples * 70.0


def cpu_baseline(wl, scene, threads: int):
    """The CPU restatement (oracle/, kind 'port') on a bounded sample of the same workload:
    every pixel of one frame while that is seconds of work (the 3-sphere scenes), otherwise a
    strided sample of the pixels through the oracle's pixel-list entry, sized for 10-30 s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import _oracle as oracle
    w, h = wl["w"], wl["h"]
    kw = dict(threads=threads)
    if scene.get("ior") is not None:
        kw["ior"] = scene["ior"]
    if scene.get("mesh") is not None:
        kw["mesh"] = scene["mesh"]
    if scene.get("tex") is not None:
        kw["texture"] = scene["tex"]
    ubo = scene["ubo"].copy()
    n_prims = scene["spheres"].shape[0] + (scene["mesh"][1].shape[0] if scene.get("mesh") is not None else 0)
    # segments the CPU manages in ~15 s: ~180 M segments x 3 spheres on 16 threads per second
    budget_tests = 2.5e9 * threads / 16.0 * 15.0
    per_pixel = wl["spp"] * 2.0 * max(n_prims, 3)
    n_pix = int(min(w * h, max(256, budget_tests / per_pixel)))
    if wl["name"] == "configs[4]":
        n_pix = min(n_pix, w * h // 16)
    if n_pix >= w * h:
        sample = f"all {w}x{h} pixels"
        pixels = None
    else:
        stride = max(1, (w * h) // n_pix)
        pixels = np.arange(stride // 2, w * h, stride, dtype=np.uint32)
        sample = f"{pixels.size} of {w * h} pixels (every {stride}th)"
        kw["pixels"] = pixels
    t0 = time.perf_counter()
    res = oracle.render(ubo, scene["spheres"], w, h, 0, 1, 1, **kw)
    dt = time.perf_counter() - t0
    return {
        "value": res["rays"] / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "nproc": os.cpu_count(),
        "kind": "port",
        "sample": f"same scene and settings, frame 0, {wl['spp']} spp, {sample}: {res['rays']} rays in {dt:.2f} s "
                  f"(brute force over {n_prims} primitives, as the shader's loop)",
    }


def source_fingerprint() -> str:
    """sha256 over the kernel sources (ray_tracer_amd/csrc/*.hip, *.h, in name order): what ties a
    committed counter pass to the code it was taken on, independent of commits that touch no kernel."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "ray_tracer_amd", "csrc")
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            with open(os.path.join(src, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def pmc_entry(config: int):
    """Counter figures of the benchmarked kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_latest.json; FETCH_SIZE doubled per the gfx950 correction, plus WRITE_SIZE).
    bench.py cannot collect hardware counters itself; {} if no profile is committed.  The entry
    carries the fingerprint of the kernel sources the passes were taken on; `traffic_stale` in the
    printed line says whether the sources have changed since."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        e = dict(d["configs"].get(str(config), {})) if "configs" in d else (dict(d) if config == 1 else {})
        if e:
            e["source_sha"] = d.get("source_sha")
            e["commit"] = d.get("commit")
            e["load_latency_cycles"] = d.get("load_latency_cycles")
            now = source_fingerprint()
            e["stale"] = None if d.get("source_sha") is None else (d.get("source_sha") != now)
            e["source_sha_now"] = now
        return e
    except Exception:
        return {}


def traffic_stamp(pmc) -> dict:
    """The fields that say where `traffic` and the counter ratios come from."""
    if not pmc:
        return {"traffic_source": None}
    out = {"traffic_source": pmc.get("source"), "traffic_commit": pmc.get("commit"),
           "traffic_source_sha": pmc.get("source_sha"), "source_sha_now": pmc.get("source_sha_now"),
           "traffic_stale": pmc.get("stale")}
    if pmc.get("stale"):
        out["warning"] = ("the kernel sources have changed since the committed counter passes were taken: "
                          "traffic and the counter ratios describe the earlier kernels (re-run tools/profile_configs.sh)")
    return out


def lbvh_algorithmic_build_bytes(n_prims: int, prim_bytes: int) -> float:
    """Bytes an LBVH build of n primitives has to move: read each primitive, write its box (24 B), write and
    then sort an 8-byte (code, index) pair through four 8-bit radix passes (read + write per pass), write
    n - 1 nodes of 64 bytes (both child boxes inline)."""
    return n_prims * (prim_bytes + 24.0 + 8.0 * (1 + 2 * 4)) + max(n_prims - 1, 0) * 64.0


def build_scene(rt, np, wl):
    w, h = wl["w"], wl["h"]
    scene = dict(ubo=rt.make_ubo(w, h, wl["spp"], wl["bounces"]), ior=None, mesh=None, tex=None)
    if wl["scene"] == "grid":
        scene["spheres"], scene["ior"] = rt.grid_spheres_scene_glass(10000)
    elif wl["scene"] == "terrain":
        scene["spheres"] = rt.default_scene()[:2]
        v, i = rt.terrain_mesh(709, phase=0.0)
        scene["mesh"] = (v, i, rt.matte(0.55, 0.5, 0.4, smoothness=0.7, specular_probability=0.15))
    elif wl["scene"] == "textured":
        scene["spheres"] = rt.default_scene()
        scene["tex"] = dict(rgba8=rt.brick_texture(1024, 1024), first_prim=2, n_prims=1, scale=(0.25, 0.25))
    else:
        scene["spheres"] = rt.default_scene()
    return scene


def apply_scene(r, rt, scene):
    r.set_params(scene["ubo"])
    r.set_spheres(scene["spheres"])
    if scene["ior"] is not None:
        r.set_dielectric(scene["ior"])
    if scene["mesh"] is not None:
        r.set_mesh(*scene["mesh"])
    if scene["tex"] is not None:
        t = scene["tex"]
        r.set_texture(t["rgba8"], t["first_prim"], t["n_prims"], rt.TEX_MAP_PLANAR_XZ, t["scale"])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 100; 10 for configs 2 and 4)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed steps (default: 30; 2 for configs 2 and 4)")
    ap.add_argument("--config", type=int, default=1, choices=sorted(WORKLOADS))
    ap.add_argument("--mode", choices=["megakernel", "wavefront", "nested"], default="megakernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="skip the untimed extras (cold first launch, wavefront, strict math, traversal tallies): "
                         "for profiler passes, where every launch of the process is averaged")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo with --rehearse-on-device lets N ranks share one GPU "
                         "to rehearse the N > 1 control flow (not a measurement)")
    ap.add_argument("--rehearse-on-device", type=int, default=-1,
                    help="put every rank on this device instead of LOCAL_RANK")
    ap.add_argument("--images-in-flight", type=int, default=0,
                    help="accumulation images per rank (default: 2 when N > 1, else 1)")
    args = ap.parse_args()
    wl = WORKLOADS[args.config]
    heavy = args.config in (2, 4)
    # The light workloads take about a millisecond per step, and the device needs ~20 ms of work to
    # reach its clocks (launch times of the 1920x1080 frame fall from 1.24 to 1.10 ms over the first
    # 18 launches of a process, whatever the launch order does): 30 untimed and 100 timed steps.
    if args.steps <= 0:
        args.steps = 10 if heavy else 100
    if args.warmup < 0:
        args.warmup = 2 if heavy else 30

    import numpy as np
    import torch
    import torch.distributed as dist

    import ray_tracer_amd as rt
    from ray_tracer_amd.distributed import finish_reduce, frame_partition, start_reduce

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.rehearse_on_device >= 0:
        local_rank = args.rehearse_on_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    mode = {"megakernel": rt.MODE_MEGAKERNEL, "wavefront": rt.MODE_WAVEFRONT,
            "nested": rt.MODE_MEGAKERNEL_NESTED}[args.mode]
    WIDTH, HEIGHT, SPP, BOUNCES = wl["w"], wl["h"], wl["spp"], wl["bounces"]
    scene = build_scene(rt, np, wl)
    ubo, spheres = scene["ubo"], scene["spheres"]
    n_spheres = spheres.shape[0]
    frames_per_step = wl["frames"] if wl["scaling"] == "strong" else world   # over all ranks

    # the accumulation images live in torch tensors so RCCL can reduce them in place
    # configs[3] on one GPU keeps two frames in flight, each in a context and on a stream of its
    # own: the LBVH rebuild of frame i+1 (nineteen small kernels and two host round trips) then runs
    # beside the trace of frame i instead of between two traces.  Every frame is still rebuilt
    # and traced inside the timed region.
    pipelined = world == 1 and ((wl["scene"] == "terrain" and args.images_in_flight != 1) or args.images_in_flight > 1)
    n_buf = args.images_in_flight if args.images_in_flight > 0 else (2 if world > 1 or pipelined else 1)
    stream = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(device=dev) if pipelined else stream for _ in range(n_buf)]
    accums, renderers = [], []
    for k in range(n_buf):
        a = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev)
        rr = rt.Renderer(WIDTH, HEIGHT, mode=mode, flags=0, device=local_rank,
                         stream=streams[k].cuda_stream, accum_buffer=a.data_ptr())
        apply_scene(rr, rt, scene)
        accums.append(a)
        renderers.append(rr)
    torch.cuda.synchronize(dev)
    in_flight = [False] * n_buf
    pending = [None] * n_buf

    # configs[3]: the terrain's vertices for eight phases of its motion, resident on the device
    # before the timed region; every step rebuilds the LBVH from one of them
    terrain = None
    if wl["scene"] == "terrain":
        idx = torch.from_numpy(scene["mesh"][1].view(np.int32)).to(dev)
        verts = [torch.from_numpy(rt.terrain_mesh(709, phase=0.15 * k)[0]).to(dev) for k in range(8)]
        terrain = dict(idx=idx, verts=verts, n_v=verts[0].shape[0], n_t=idx.shape[0], mat=scene["mesh"][2],
                       build_ms=0.0, builds=0)

    rehearsal = world > 1 and args.backend == "gloo"
    small = torch.device("cpu") if rehearsal else dev   # where the scalar all-reduces live

    def begin_sum(a):
        if not rehearsal:
            return start_reduce(a, dst=0)
        # rehearsal: gloo sums a host copy (its CUDA path is not what RCCL will run anyway)
        host = a.cpu()
        return (dist.reduce(host, dst=0, op=dist.ReduceOp.SUM, async_op=True), host, a)

    def end_sum(handle) -> None:
        if not rehearsal:
            finish_reduce(handle)
        elif handle is not None:
            work, host, a = handle
            work.wait()
            a.copy_(host)

    def retire(k: int) -> None:
        """finish the step that last used image k: its sum, then rank 0's 8-bit image"""
        if in_flight[k]:
            end_sum(pending[k])
            if rank == 0 and world > 1:   # one GPU: the render kernel wrote the 8-bit image itself
                renderers[k].resolve()
            in_flight[k], pending[k] = False, None

    def step(i: int) -> None:
        k = i % n_buf
        retire(k)
        with torch.cuda.stream(streams[k]):
            accums[k].zero_()
        if terrain is not None:
            v = terrain["verts"][i % len(terrain["verts"])]
            renderers[k].set_mesh_device(v.data_ptr(), terrain["n_v"], terrain["idx"].data_ptr(), terrain["n_t"],
                                         terrain["mat"])
        first, count, stride = frame_partition(i * frames_per_step, frames_per_step, rank, world)
        renderers[k].render(first, count, stride)
        pending[k] = begin_sum(accums[k])
        in_flight[k] = True
        if n_buf == 1:
            retire(k)

    def fence() -> None:
        for k in range(n_buf):
            retire(k)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def totals():
        st = [rr.stats() for rr in renderers]
        return sum(s.rays for s in st), sum(s.total_render_ms for s in st)

    cold_ms = None
    # outside the timed region: the other kernel organisation
    # (configs[1] asks for "megakernel vs wavefront"), the bit-exact strict-math launch, and for
    # LBVH scenes the traversal tallies that price a path segment in bytes.  They run BEFORE the
    # warm-up steps rather than after the timed ones: an idle device needs ~20 ms of work to reach
    # its clocks (the 1920x1080 frame takes 1.24 ms on the second launch of a process and 1.10 ms
    # from the twentieth on), and with these launches first the W warm-up and K timed steps all
    # see the device as a renderer that has been running for a moment sees it.
    other_ms = strict_ms = None
    bytes_per_segment = nodes_per_seg = prims_per_seg = None
    # (every rank of a multi-GPU run does them too, so that each device starts its steps warm)
    if args.mode == "megakernel" and not args.no_side_runs:
        def side(mode_, flags, frames=3):
            with rt.Renderer(WIDTH, HEIGHT, mode=mode_, flags=flags, device=local_rank) as x:
                apply_scene(x, rt, scene)
                x.render(0, wl["frames"], 1)
                x.stats()
                t = []
                for f in range(1, frames + 1):
                    x.render(f * wl["frames"], wl["frames"], 1)
                    t.append(x.stats().last_render_ms)
                return sorted(t)[len(t) // 2], x.stats()
        if args.config in (0, 1, 2, 3):
            other_ms, _ = side(rt.MODE_WAVEFRONT, 0, 3 if args.config != 2 else 1)
        strict_ms, _ = side(rt.MODE_MEGAKERNEL, rt.FLAG_STRICT_MATH, 3 if not heavy else 1)
        if wl["scene"] in ("grid", "terrain"):
            with rt.Renderer(WIDTH, HEIGHT, flags=rt.FLAG_COUNT_TRAVERSAL, device=local_rank) as x:
                probe = dict(scene)
                probe["ubo"] = rt.make_ubo(WIDTH, HEIGHT, 2, BOUNCES)
                apply_scene(x, rt, probe)
                x.render(0, 1, 1)
                s = x.stats()
                nodes_per_seg = s.bvh_nodes_visited / max(s.rays, 1)
                prims_per_seg = s.bvh_spheres_tested / max(s.rays, 1)
    # the LBVH build on a device that does nothing else (HIP events around the build's own kernels)
    build_block = None
    if wl["scene"] in ("grid", "terrain") and args.mode == "megakernel" and not args.no_side_runs:
        with rt.Renderer(WIDTH, HEIGHT, device=local_rank) as x:
            apply_scene(x, rt, scene)
            t = [x.stats().last_build_ms]
            for k in range(6):
                if terrain is not None:
                    v = terrain["verts"][k % len(terrain["verts"])]
                    x.set_mesh_device(v.data_ptr(), terrain["n_v"], terrain["idx"].data_ptr(), terrain["n_t"],
                                      terrain["mat"])
                else:
                    x.set_spheres(scene["spheres"])
                    if scene["ior"] is not None:
                        x.set_dielectric(scene["ior"])
                x.sync()
                t.append(x.stats().last_build_ms)
            t = sorted(t[1:])
            n_prims = n_spheres + (terrain["n_t"] if terrain is not None else 0)
            prim_bytes = 36 if terrain is not None else 16
            ms = t[len(t) // 2]
            alg = lbvh_algorithmic_build_bytes(n_prims, prim_bytes)
            build_block = {"ms": ms, "ms_min": t[0], "primitives": n_prims, "mprims_per_s": n_prims / ms / 1e3,
                           "algorithmic_bytes": alg, "gb_per_s": alg / ms / 1e6, "frac_of_hbm_peak": alg / ms / 1e6 / PEAK_HBM_GBS,
                           "height": x.stats().bvh_height,
                           "how": "median of 6 rebuilds on an otherwise idle device, HIP events around the build's kernels"}
    # the reduce on its own: one accumulation image summed onto rank 0, nothing else running
    reduce_block = None
    if world > 1:
        probe_img = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev)
        times = []
        for k in range(6):
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0r = time.perf_counter()
            end_sum(begin_sum(probe_img))
            torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - t0r)
        tr = torch.tensor([sorted(times[1:])[2]], dtype=torch.float64, device=small)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        nbytes = HEIGHT * WIDTH * 16
        peak = 7 * 153.0
        reduce_block = {"bytes_per_rank": nbytes, "ms": float(tr.item()) * 1e3,
                        "gb_per_s_into_root": (world - 1) * nbytes / float(tr.item()) / 1e9,
                        "peak_GBs": peak, "frac": (world - 1) * nbytes / float(tr.item()) / 1e9 / peak,
                        "how": "median of 5 stand-alone reduces of one W x H float4 image (max over ranks); the root "
                               "receives (N - 1) images over its 7 xGMI links of ~153 GB/s each; in the timed steps the "
                               "reduce of step i runs beside the trace of step i + 1"}
        del probe_img
    for i in range(args.warmup):
        step(i)
    fence()
    rays0, ms0 = totals()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    rays1, ms1 = totals()

    rays = float(rays1 - rays0)
    kernel_ms = float(ms1 - ms0)
    st_last = renderers[0].stats()

    if world == 1 and args.mode == "megakernel" and not args.no_side_runs:
        # first launch of a NEW context on a process whose kernels are loaded: no tile costs yet, so
        # a two-sample cost probe and the ordering pass precede the frame (all inside the figure)
        with rt.Renderer(WIDTH, HEIGHT, mode=mode, flags=0, device=local_rank) as x:
            apply_scene(x, rt, scene)
            x.render(0, wl["frames"], 1)
            cold_ms = x.stats().last_render_ms

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=small)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([rays], dtype=torch.float64, device=small)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_rays = float(t.item())
    else:
        total_rays = rays

    if rank == 0:
        steps = max(args.steps, 1)
        my_frames = frame_partition(0, frames_per_step, 0, world)[1]
        samples = float(WIDTH * HEIGHT * SPP * my_frames * steps)      # rank 0's
        launch_ms = kernel_ms / steps                                  # rank 0's render launches, HIP events
        pmc = pmc_entry(args.config) if args.mode == "megakernel" else {}
        cfg = {
            "workload": wl["text"],
            "mode": args.mode,
            "math": "fast (hardware transcendental, sqrt and rcp units in sampling and distances; "
                    "hit decisions and primary rays exact)",
            "parallelism": f"frame-split x{world}" + (" + RCCL reduce overlapped with the next step"
                                                      if world > 1 else "") +
                           (" [REHEARSAL: gloo on host copies, ranks share a device -- not a measurement]"
                            if rehearsal else ""),
            "ms_per_launch": launch_ms,
            "rays_per_step": total_rays / steps,
            "tile_order": "warm: every timed launch runs its tiles heaviest-first by the costs the previous "
                          "launch of the same view measured (progressive rendering repeats the view)",
            "cold_first_launch_ms": cold_ms,
            "strict_math_ms_per_launch": strict_ms,
            "wavefront_ms_per_launch": other_ms,
        }
        if args.config == 1:   # names kept from the first round's line
            cfg["ms_per_frame_1080p_64spp"] = launch_ms
            cfg["rays_per_frame"] = rays / steps
            cfg["wavefront_ms_per_frame_1080p_64spp"] = other_ms
            cfg["strict_ms_per_frame"] = strict_ms
        if terrain is not None:
            cfg["lbvh_build_ms_in_flight"] = st_last.last_build_ms   # beside another frame's trace; lbvh_build.ms is the build alone
            cfg["trace_ms"] = launch_ms
            cfg["lbvh_height"] = st_last.bvh_height
            cfg["frames_in_flight"] = n_buf
            cfg["note"] = ("value = segments / wall time of rebuild + trace per step; vertices and indices are "
                           "device-resident (rt_set_mesh_device), so no host-device copy is inside a step" +
                           ("; two frames in flight in two contexts on two streams: the rebuild of a frame and the "
                            "start of its trace run beside the end of the trace before it, so a step takes less than "
                            "one trace measured by its own events (trace_ms, which the roofline uses: conservative); "
                            "--images-in-flight 1 runs one frame after the other"
                            if pipelined else ""))
        if wl["scene"] in ("grid", "terrain"):
            # algorithmic bytes of a path segment through the LBVH: 64-byte nodes, 16-byte spheres
            # or 48-byte triangles, 64 bytes of material and geometry on a hit
            prim_bytes = 48.0 if wl["scene"] == "terrain" else 16.0
            hit_frac = max(rays - samples, 0.0) / max(rays, 1.0)
            bytes_per_segment = 64.0 * nodes_per_seg + prim_bytes * prims_per_seg + 64.0 * hit_frac \
                if nodes_per_seg is not None else None
            kernel = "k_mega_sched<fast>" if args.mode == "megakernel" else args.mode
            if bytes_per_segment is not None:
                alg_bytes = bytes_per_segment * rays / steps + 44.0 * WIDTH * HEIGHT * my_frames
                achieved = alg_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
            else:
                alg_bytes, achieved = None, None
            launch_s = launch_ms * 1e-3
            # three resources, each as a fraction of its own peak; the largest one governs
            hbm_bytes = pmc.get("hbm_bytes_per_launch")
            hbm_frac = hbm_bytes / launch_s / 1e9 / PEAK_HBM_GBS if hbm_bytes and launch_s > 0 else None
            l2_bytes = pmc.get("l1_to_l2_bytes_per_launch")
            l2_frac = l2_bytes / launch_s / 1e9 / PEAK_L2_GBS if l2_bytes and launch_s > 0 else None
            lat = pmc.get("load_latency_cycles") or {}
            lat_frac = lat_ms = lat_avg = None
            if nodes_per_seg is not None and lat and pmc.get("l1_hit_rate") is not None and \
                    pmc.get("waves_per_simd") and pmc.get("valu_lane_utilisation"):
                l1, l2h = pmc["l1_hit_rate"], pmc.get("l2_hit_rate", 1.0)
                lat_avg = l1 * lat["l1_16KB"] + (1 - l1) * (l2h * lat["l2_1MB"] + (1 - l2h) * lat["mall_64MB"])
                fetches = (rays / steps) * (nodes_per_seg + prims_per_seg)
                lanes = 1024 * pmc["waves_per_simd"] * 64 * pmc["valu_lane_utilisation"]
                lat_ms = fetches * lat_avg / lanes / 2.4e9 * 1e3
                lat_frac = lat_ms / launch_ms if launch_ms > 0 else None
            cands = [(f, n) for f, n in ((hbm_frac, "hbm"), (l2_frac, "l2"), (lat_frac, "latency")) if f is not None]
            governing = max(cands)[1] if cands else "hbm"
            frac = dict((n, f) for f, n in cands).get(governing)
            roof = {
                "bound": governing,
                "kernel": kernel,
                "frac": frac,
                "achieved": {"hbm": (hbm_bytes or 0) / launch_s / 1e9 if launch_s > 0 else None,
                             "l2": (l2_bytes or 0) / launch_s / 1e9 if launch_s > 0 else None,
                             "latency": lat_ms}.get(governing),
                "peak": {"hbm": PEAK_HBM_GBS, "l2": PEAK_L2_GBS, "latency": launch_ms}.get(governing),
                "unit": {"hbm": "GB/s", "l2": "GB/s", "latency": "ms (model of dependent fetches / measured launch)"}.get(governing),
                "traffic": hbm_bytes,
                **traffic_stamp(pmc),
                "launch_ms": launch_ms,
                "frac_hbm": hbm_frac,
                "frac_l2": l2_frac,
                "frac_latency": lat_frac,
                "l1_to_l2_bytes_per_launch": l2_bytes,
                "latency_model": {
                    "ms": lat_ms, "mean_load_latency_cycles": lat_avg, "load_latency_cycles": lat or None,
                    "what": "segments x (nodes + primitives fetched per segment, tallied live) x mean dependent-load "
                            "latency (idle-chip pointer chase, tools/micro/load_latency.hip, weighted by the counter "
                            "passes' L1 / L2 hit rates) / (1024 SIMDs x resident waves x 64 lanes x share of lanes active) "
                            "at 2.4 GHz: the time the launch needs if every active lane keeps exactly one fetch in flight"},
                # the rate at which the search consumes node and primitive bytes, served by L1 / L2: NOT a share of HBM
                "achieved_cached_gb_per_s": achieved,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_per_segment": bytes_per_segment,
                "nodes_per_segment": nodes_per_seg,
                "primitives_tested_per_segment": prims_per_seg,
                "valu_instructions_per_simd_cycle": pmc.get("valu_instructions_per_simd_cycle"),
                "valu_issue_frac_of_hw_rate": pmc.get("valu_issue_frac_of_hw_rate"),
                "valu_lane_utilisation": pmc.get("valu_lane_utilisation"),
                "waves_per_simd": pmc.get("waves_per_simd"),
                "l1_hit_rate": pmc.get("l1_hit_rate"),
                "l2_hit_rate": pmc.get("l2_hit_rate"),
            }
        else:
            flops = algorithmic_flops(rays, samples, n_spheres) / steps
            achieved = flops / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0
            roof = {
                # Branchy FP32 vector work with no dense contraction: the kernel issues no MFMA, and
                # the bound that applies is FP32 vector issue (the same 157.3 TFLOP/s as the f32 MFMA peak).
                "bound": "valu",
                "kernel": {"megakernel": "k_mega_regen<fast,brute>", "nested": "k_mega_nested<fast,brute>",
                           "wavefront": "wavefront stages"}[args.mode],
                "achieved": achieved,
                "peak": PEAK_FP32_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                "traffic": pmc.get("hbm_bytes_per_launch"),
                **traffic_stamp(pmc),
                # from the same committed counter passes: wave-level VALU instructions per SIMD-cycle over the
                # hardware rate (a wave64 VALU instruction occupies a SIMD-32 for 2 cycles: 0.5 per cycle), over
                # what a SIMD was measured to issue for independent adds (0.40, tools/micro/issue_cost.hip), and
                # the share of the 64 lanes active per instruction
                "valu_instructions_per_simd_cycle": pmc.get("valu_instructions_per_simd_cycle"),
                "valu_issue_frac_of_hw_rate": pmc.get("valu_issue_frac_of_hw_rate"),
                "valu_issue_frac_of_measured_ceiling": pmc.get("valu_issue_frac"),
                "valu_lane_utilisation": pmc.get("valu_lane_utilisation"),
                "launch_ms": launch_ms,
                "algorithmic_flops_per_launch": flops,
                "note": "FP32 vector issue bound (no MFMA in a path tracer); HBM traffic is ~50 B/pixel/frame",
            }
        out = {
            "metric": "Mrays/sec (all bounces)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": wl["scaling"],
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": cfg,
            "roofline": roof,
        }
        if build_block is not None:
            out["lbvh_build"] = build_block
        if reduce_block is not None:
            out["reduce"] = reduce_block
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(wl, scene, threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    for rr in renderers:
        rr.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

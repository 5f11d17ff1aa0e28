#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (every path segment) of the path-tracing hot path.

Workload (BASELINE.json configs[1]): the reference's default 3-sphere scene at
1920x1080, 64 samples per pixel per frame, at most 25 bounces, camera viewport set to
the image size.  One step = one frame batch of the hot path per GPU: zero the float
accumulation image, render one 64-spp frame into it, (N > 1: sum the N accumulation
images onto rank 0 with one RCCL reduce over xGMI,) resolve to the 8-bit image.  With
N GPUs rank r renders frame index step*N + r -- the sample-wise split whose frames
carry the shader's own per-frame RNG streams -- so per-GPU work is fixed (weak
scaling) and every step yields an N*64-spp image on rank 0.  For N > 1 each rank keeps
two accumulation images: the reduce of step i runs on RCCL's stream while step i+1 is
traced into the other image, and rank 0 resolves step i when its sum has landed; all
K renders, K reduces and K resolves lie inside the timed region.

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

WIDTH, HEIGHT, SPP, BOUNCES = 1920, 1080, 64, 25


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


def cpu_baseline(ubo, spheres, threads: int):
    """The CPU restatement (oracle/, kind 'port') on a bounded sample of the same workload:
    one full frame of the benchmark workload (all 64 spp), timed once."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as oracle
    ubo = ubo.copy()
    spp = SPP
    ubo[40] = float(spp)
    t0 = time.perf_counter()
    res = oracle.render(ubo, spheres, WIDTH, HEIGHT, 0, 1, 1, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": res["rays"] / dt / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"same scene {WIDTH}x{HEIGHT}, {spp} spp (of {SPP}), maxBounce {BOUNCES}, "
                  f"frame 0: {res['rays']} rays in {dt:.2f} s",
    }


def pmc_traffic_bytes():
    """HBM bytes per launch of the benchmarked kernel from the committed rocprofv3 --pmc
    passes (profiles/: FETCH_SIZE doubled per the gfx950 correction, plus WRITE_SIZE).
    bench.py cannot collect hardware counters itself; None if no profile is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return float(d["hbm_bytes_per_launch"]), d.get("source", path), d
    except Exception:
        return None, None, {}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["megakernel", "wavefront", "nested"], default="megakernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo with --rehearse-on-device lets N ranks share one GPU "
                         "to rehearse the N > 1 control flow (not a measurement)")
    ap.add_argument("--rehearse-on-device", type=int, default=-1,
                    help="put every rank on this device instead of LOCAL_RANK")
    ap.add_argument("--images-in-flight", type=int, default=0,
                    help="accumulation images per rank (default: 2 when N > 1, else 1)")
    args = ap.parse_args()

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
    ubo = rt.make_ubo(WIDTH, HEIGHT, SPP, BOUNCES)
    spheres = rt.default_scene()
    n_spheres = spheres.shape[0]

    # the accumulation images live in torch tensors so RCCL can reduce them in place
    n_buf = args.images_in_flight if args.images_in_flight > 0 else (2 if world > 1 else 1)
    stream = torch.cuda.current_stream(dev)
    accums, renderers = [], []
    for _ in range(n_buf):
        a = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev)
        rr = rt.Renderer(WIDTH, HEIGHT, mode=mode, flags=0, device=local_rank,
                         stream=stream.cuda_stream, accum_buffer=a.data_ptr())
        rr.set_params(ubo)
        rr.set_spheres(spheres)
        accums.append(a)
        renderers.append(rr)
    in_flight = [False] * n_buf
    pending = [None] * n_buf

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
        accums[k].zero_()
        first, count, stride = frame_partition(i * world, world, rank, world)
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

    # configs[1] asks for "megakernel vs wavefront": time a few frames of the other
    # organisation outside the timed region (rank 0 of a single-GPU run only)
    other_ms = None
    if world == 1 and args.mode == "megakernel":
        with rt.Renderer(WIDTH, HEIGHT, mode=rt.MODE_WAVEFRONT, device=local_rank) as wf:
            wf.set_params(ubo)
            wf.set_spheres(spheres)
            wf.render(0, 1, 1)
            wf.stats()
            t = []
            for f in range(1, 4):
                wf.render(f, 1, 1)
                t.append(wf.stats().last_render_ms)
            other_ms = sorted(t)[1]
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
        samples = float(WIDTH * HEIGHT * SPP * args.steps)  # per GPU
        launch_ms = kernel_ms / max(args.steps, 1)          # rank 0's render launches, HIP events
        flops = algorithmic_flops(rays, samples, n_spheres) / max(args.steps, 1)
        achieved = flops / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0
        traffic, traffic_src, pmc = pmc_traffic_bytes() if args.mode == "megakernel" else (None, None, {})
        out = {
            "metric": "Mrays/sec (all bounces)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"reference default scene (3 spheres, main.cpp:517-538), {WIDTH}x{HEIGHT}, "
                            f"{SPP} spp/frame, maxBounce {BOUNCES}, 1 frame per GPU per step "
                            "(BASELINE.json configs[1])",
                "mode": args.mode,
                "math": "fast (hardware transcendental, sqrt and rcp units in sampling and distances; "
                        "hit decisions and primary rays exact)",
                "parallelism": f"frame-split x{world}" + (" + RCCL reduce overlapped with the next frame"
                                                          if world > 1 else "") +
                               (" [REHEARSAL: gloo on host copies, ranks share a device -- not a measurement]"
                                if rehearsal else ""),
                "ms_per_frame_1080p_64spp": launch_ms,
                "rays_per_frame": rays / max(args.steps, 1),
                "wavefront_ms_per_frame_1080p_64spp": other_ms,
            },
            "roofline": {
                # the contract's two bounds are "hbm" and "mfma": this is the compute one.  The kernel
                # issues no MFMA (there is no dense contraction in a path tracer); its arithmetic is
                # FP32 vector work, priced against the f32 peak, which is the same 157.3 TFLOP/s
                "bound": "mfma",
                "compute_unit": "valu",
                "kernel": {"megakernel": "k_mega_regen<fast,brute>", "nested": "k_mega_nested<fast,brute>",
                           "wavefront": "wavefront stages"}[args.mode],
                "achieved": achieved,
                "peak": PEAK_FP32_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                # from the same committed counter passes: wave-level VALU instructions per SIMD-cycle
                # over what a SIMD was measured to issue, and share of the 64 lanes active per instruction
                "valu_issue_frac": pmc.get("valu_issue_frac"),
                "valu_lane_utilisation": pmc.get("valu_lane_utilisation"),
                "launch_ms": launch_ms,
                "algorithmic_flops_per_launch": flops,
                "note": "branchy FP32 vector work, no dense contraction: priced against the FP32 "
                        "vector peak (= the f32 MFMA peak); HBM traffic is ~50 B/pixel/frame",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(ubo, spheres, threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    for rr in renderers:
        rr.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

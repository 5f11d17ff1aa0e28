#!/usr/bin/env python3
"""One rank of a multi-process GPU render (started by tests/test_gpu_multi.py through
`python -m torch.distributed.run`): the HIP kernels trace this rank's share of the frames of a
10k-sphere-style scene (spheres of three material classes incl. glass, LBVH), the accumulation
images are summed onto rank 0, rank 0 resolves and writes what it holds to an .npz file.

With at least WORLD_SIZE devices each rank takes device LOCAL_RANK and the sum is RCCL's; with fewer
(the one-GPU box) all ranks share device 0 and the sum goes through gloo on host copies.

    mp_gpu_worker.py OUT.npz WIDTH HEIGHT SPP BOUNCES FRAMES N_SPHERES STRICT
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    out, w, h, spp, bounces, frames, n_spheres, strict = sys.argv[1], *map(int, sys.argv[2:9])
    import numpy as np
    import torch
    import torch.distributed as dist

    import ray_tracer_amd as rt
    from ray_tracer_amd.distributed import render_split

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    shared = torch.cuda.device_count() < world
    device = 0 if shared else local
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if shared:
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=dev)
    sph, ior = rt.grid_spheres_scene_glass(n_spheres)
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    flags = rt.FLAG_STRICT_MATH if strict else 0
    with rt.Renderer(w, h, flags=flags, device=device, stream=torch.cuda.current_stream(dev).cuda_stream,
                     accum_buffer=accum.data_ptr()) as r:
        r.set_params(rt.make_ubo(w, h, spp, bounces))
        r.set_spheres(sph)
        r.set_dielectric(ior)
        render_split(r, accum, 0, frames, rank, world, stage_on_host=shared)
        rays = torch.tensor([r.stats().rays], dtype=torch.int64, device="cpu" if shared else dev)
        dist.all_reduce(rays)
        if rank == 0:
            np.savez(out, accum=accum.cpu().numpy(), rgba8=r.read_rgba8(), rays=np.int64(rays.item()),
                     backend=np.array("gloo-host" if shared else "nccl"), bvh_nodes=np.int64(r.stats().bvh_nodes))
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

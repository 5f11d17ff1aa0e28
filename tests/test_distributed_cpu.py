"""The N > 1 path on CPU: two gloo ranks render their share of the frames (here with the
CPU oracle standing in for the device), reduce the accumulation images, and rank 0's sum
equals the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ray_tracer_amd.distributed import (finish_reduce, frame_partition, reduce_accumulation,
                                        start_reduce)

HERE = os.path.dirname(os.path.abspath(__file__))


def test_frame_partition_covers_every_frame_once():
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 5, 8, 16, 17):
            seen = []
            for rank in range(world):
                first, count, stride = frame_partition(10, n, rank, world)
                assert stride == world
                seen += [first + k * stride for k in range(count)]
            assert sorted(seen) == list(range(10, 10 + n))
    with pytest.raises(ValueError):
        frame_partition(0, 4, 2, 2)


def test_reduce_helpers_are_no_ops_without_a_process_group():
    t = torch.ones(4)
    assert reduce_accumulation(t) is t
    assert start_reduce(t) is None
    finish_reduce(None)
    assert torch.equal(t, torch.ones(4))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    sys.path.insert(0, HERE)
    import _oracle as oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, frames = 48, 32, 5
    cam = oracle.camera_create(90.0, 1.0, 0.01, 10000.0)
    cam.viewport_width, cam.viewport_height = float(w), float(h)
    ubo = oracle.default_ubo(cam)
    ubo[40], ubo[41] = 2, 6
    first, count, stride = frame_partition(0, frames, rank, world)
    res = oracle.render(ubo, oracle.default_spheres(), w, h, first, count, stride, threads=2)
    accum = torch.from_numpy(res["accum"])
    rays = torch.tensor([res["rays"]], dtype=torch.int64)
    # the split form, two images in flight as bench.py keeps them
    twice = accum * 2.0
    pending = [start_reduce(accum, dst=0), start_reduce(twice, dst=0)]
    assert all(p is not None for p in pending)
    for p in pending:
        finish_reduce(p)
    dist.all_reduce(rays)
    if rank == 0:
        assert torch.equal(twice, accum * 2.0)
        np.savez(out_path, accum=accum.numpy(), rays=rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_split_matches_single_process(oracle, tmp_path):
    out = str(tmp_path / "sum.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    w, h, frames = 48, 32, 5
    cam = oracle.camera_create(90.0, 1.0, 0.01, 10000.0)
    cam.viewport_width, cam.viewport_height = float(w), float(h)
    ubo = oracle.default_ubo(cam)
    ubo[40], ubo[41] = 2, 6
    want = oracle.render(ubo, oracle.default_spheres(), w, h, 0, frames, 1)
    assert int(got["rays"][0]) == want["rays"]
    assert np.array_equal(got["accum"][..., 3], want["accum"][..., 3])
    # same five per-frame colours, summed in a different order
    assert np.max(np.abs(got["accum"] - want["accum"])) <= 5 * 2.0 ** -23 * 5
    assert np.array_equal(oracle.resolve_rgba8(got["accum"]), oracle.resolve_rgba8(want["accum"])) or \
        np.max(np.abs(oracle.resolve_rgba8(got["accum"]).astype(int) -
                      oracle.resolve_rgba8(want["accum"]).astype(int))) <= 1

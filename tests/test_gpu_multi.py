"""The N > 1 path with the HIP kernels, on however many devices the box has.

On a one-GPU box: two (and eight) processes share device 0, each traces its share of the frames with
the kernels, the sum goes through gloo on host copies.  On a multi-GPU box the same tests also run with
one device per rank and RCCL (2, 4, 8 ranks up to the device count), `rt_headless --gpus N` sums
with ncclReduce between N devices of one process, and `bench.py --gpus N` prints its line.
The partition is the shader's own: frames are independent sample batches (ray_tracing.comp:80-82)
averaged with equal weight (:106-107).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import ray_tracer_amd as rt

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W, H, SPP, BOUNCES, FRAMES, SPHERES = 160, 96, 4, 8, 8, 2000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _device_count():
    import torch
    return torch.cuda.device_count()


def _torchrun(n, script_args, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _single(strict):
    sph, ior = rt.grid_spheres_scene_glass(SPHERES)
    with rt.Renderer(W, H, flags=rt.FLAG_STRICT_MATH if strict else 0) as r:
        r.set_params(rt.make_ubo(W, H, SPP, BOUNCES))
        r.set_spheres(sph)
        r.set_dielectric(ior)
        r.render(0, FRAMES, 1)
        acc = r.read_accum()
        r.resolve()
        return acc, r.read_rgba8(), int(r.stats().rays), int(r.stats().bvh_nodes)


def _ranks():
    """2 and 8 ranks always (sharing device 0 through gloo when the box has fewer devices: the smallest
    and the largest run, one frame per rank at 8); 4 when there are that many devices."""
    n = _device_count()
    return [2, 8] + ([4] if 4 <= n else [])


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("strict", [1, 0])
def test_ranks_render_their_frames_and_rank0_holds_the_image(oracle, tmp_path, world, strict):
    if world not in _ranks():
        pytest.skip(f"{world} ranks need {world} devices, the box has {_device_count()}")
    out = str(tmp_path / "image.npz")
    res = _torchrun(world, [os.path.join(HERE, "mp_gpu_worker.py"), out, str(W), str(H), str(SPP), str(BOUNCES),
                            str(FRAMES), str(SPHERES), str(strict)])
    assert res.returncode == 0, res.stderr[-3000:]
    got = np.load(out)
    assert str(got["backend"]) == ("nccl" if _device_count() >= world else "gloo-host")
    acc, img, rays, nodes = _single(strict)
    assert int(got["rays"]) == rays and int(got["bvh_nodes"]) == nodes == SPHERES - 1 - (SPHERES - 1 - nodes)
    # the same eight per-frame colours per pixel, summed in another order
    assert np.array_equal(got["accum"][..., 3], acc[..., 3]) and (acc[..., 3] == FRAMES).all()
    finite = np.isfinite(acc).all(-1)
    assert np.max(np.abs(got["accum"] - acc)[finite]) <= FRAMES * 2.0 ** -23 * np.max(acc[finite])
    assert np.max(np.abs(got["rgba8"].astype(int) - img.astype(int))) <= 1
    if strict:
        # and against the oracle, brute force over every sphere, on a sample of the pixels
        sph, ior = rt.grid_spheres_scene_glass(SPHERES)
        pix = np.arange(7, W * H, 41, dtype=np.uint32)
        want = oracle.render(rt.make_ubo(W, H, SPP, BOUNCES), sph, W, H, 0, FRAMES, 1, ior=ior, pixels=pix)
        mine = got["accum"].reshape(-1, 4)[pix]
        assert np.max(np.abs(mine - want["accum"])) <= FRAMES * 2.0 ** -23 * np.max(want["accum"])
        assert np.max(np.abs(got["rgba8"].reshape(-1, 4)[pix].astype(int) -
                             oracle.resolve_rgba8(want["accum"].reshape(1, -1, 4)).reshape(-1, 4).astype(int))) <= 1


def _headless(args, timeout=300):
    exe = os.path.join(os.path.dirname(rt.LIB_PATH), "rt_headless")
    if not os.path.exists(exe):
        pytest.skip("rt_headless is not built (make -C ray_tracer_amd/csrc)")
    return subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("gpus", [2, 4, 8])
def test_headless_driver_on_n_devices_equals_one_device(tmp_path, gpus):
    """rt_headless --gpus N (a thread per device, then one ncclReduce, in one process): image(N) == image(1)
    up to the order of the float sums.  The one-device reference takes the same path with a communicator
    of one rank (--reduce), so both pictures are resolved from a float sum of the same eight frames;
    test_headless_driver_multi_device_path pins that path to the library on any box."""
    if gpus > _device_count():
        pytest.skip(f"--gpus {gpus} needs {gpus} devices, the box has {_device_count()}")
    common = ["--width", "176", "--height", "112", "--samples", "3", "--bounces", "6", "--frames", "8",
              "--scene", "glass:500", "--reduce"]
    one, many = str(tmp_path / "one.ppm"), str(tmp_path / "many.ppm")
    a = _headless(common + ["--gpus", "1", "--out", one])
    b = _headless(common + ["--gpus", str(gpus), "--out", many])
    assert a.returncode == 0, a.stderr
    assert b.returncode == 0, b.stderr
    ia, ib = json.loads(a.stdout.strip().splitlines()[-1]), json.loads(b.stdout.strip().splitlines()[-1])
    assert ia["rays"] == ib["rays"] and ib["gpus"] == gpus
    assert ia["reduce_bytes_per_device"] == 0 and ib["reduce_bytes_per_device"] == 176 * 112 * 16
    head = b"P6\n176 112\n255\n"
    pa = np.frombuffer(open(one, "rb").read()[len(head):], dtype=np.uint8).astype(int)
    pb = np.frombuffer(open(many, "rb").read()[len(head):], dtype=np.uint8).astype(int)
    assert np.abs(pa - pb).max() <= 1 and float((pa != pb).mean()) < 0.01


def test_headless_driver_reports_a_failing_rank_instead_of_hanging():
    """A rank whose render fails must end the run with its message (exit 1) while the others have not
    yet entered the collective -- not leave them waiting in it (--fail-rank: fault injection)."""
    n = min(_device_count(), 8)
    res = _headless(["--width", "64", "--height", "48", "--frames", "4", "--gpus", str(n), "--reduce",
                     "--fail-rank", str(n - 1)], timeout=120)
    assert res.returncode == 1 and "injected failure" in res.stderr, (res.returncode, res.stderr)
    assert f"device {n - 1}" in res.stderr


@pytest.mark.parametrize("gpus", [2, 4, 8])
def test_bench_line_on_n_devices(gpus):
    """bench.py --gpus N as the driver launches it (RCCL).  On a one-GPU box the control flow of N = 2 and
    N = 8 is rehearsed with all ranks on device 0 and gloo on host copies -- not a measurement."""
    n_dev = _device_count()
    args = [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--config", "0",
            "--no-side-runs"]
    if gpus > n_dev:
        if gpus == 4:
            pytest.skip(f"--gpus {gpus} needs {gpus} devices, the box has {n_dev}")
        args += ["--backend", "gloo", "--rehearse-on-device", "0"]   # (2 and 8 ranks: the driver's smallest and largest run)
    res = _torchrun(gpus, args)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == gpus and line["unit"] == "Mrays/s" and line["value"] > 0
    assert line["cpu_baseline"] is None
    # weak scaling: one 1280x720 1-spp frame per rank per step, ~1.5 segments per sample
    assert 1.2 * 1280 * 720 * gpus <= line["config"]["rays_per_step"] <= 3 * 1280 * 720 * gpus
    if gpus <= n_dev:
        red = line["reduce"]
        assert red["bytes_per_rank"] == 1280 * 720 * 16 and red["peak_GBs"] == 7 * 153.0

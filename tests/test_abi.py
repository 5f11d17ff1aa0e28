"""The C-ABI library: loads, exports every declared symbol, and its host half (camera,
default scene, uniform block) agrees with the CPU oracle bit for bit.  No GPU needed."""
import ctypes as C
import os

import numpy as np
import pytest

import ray_tracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    syms = rt.exported_symbols()
    assert len(syms) >= 30
    missing = [n for n, ok in syms if not ok]
    assert not missing, missing
    for must in ("rt_create", "rt_destroy", "rt_set_params", "rt_set_spheres", "rt_render",
                 "rt_sync", "rt_read_rgba8", "rt_read_accum_f32", "rt_read_primary_ids",
                 "rt_get_stats", "rt_last_error", "rt_camera_create", "rt_default_ubo"):
        assert (must, True) in syms


def test_contract_layout_compiles_as_c(tmp_path):
    """rt_contract.h carries the size/offset static asserts (272-byte UBO, 64-byte sphere);
    compiling it as plain C checks them with a second compiler."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    src = tmp_path / "t.c"
    src.write_text('#include "rt_hip.h"\n#include "rt_host.h"\nint main(void){return 0;}\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_struct_sizes_in_the_binding():
    assert C.sizeof(rt.Camera) == 36
    assert C.sizeof(rt.Stats) == 64
    assert rt.default_ubo().nbytes == 272 and rt.default_scene().nbytes == 192


def test_default_ubo_and_scene_match_oracle(oracle):
    assert np.array_equal(rt.default_ubo().view(np.uint32), oracle.default_ubo().view(np.uint32))
    assert np.array_equal(rt.default_scene().view(np.uint32), oracle.default_spheres().view(np.uint32))
    s = rt.default_scene()
    assert tuple(s[2, :4]) == (0.0, -101.0, -5.0, 100.0) and s[0, 15] == 0.5 and s[1, 7] == 0.5


def test_camera_matches_oracle_bit_for_bit(oracle):
    rng = np.random.default_rng(5)
    for _ in range(50):
        pitch, yaw = (rng.random(2) * 6 - 3).astype(np.float32)
        a = rt.Camera.create(90.0, 1.0, 0.01, 10000.0)
        b = oracle.camera_create(90.0, 1.0, 0.01, 10000.0)
        a.pitch = b.pitch = float(pitch)
        a.yaw = b.yaw = float(yaw)
        a.viewport_width = b.viewport_width = 1920.0
        a.viewport_height = b.viewport_height = 1080.0
        assert np.array_equal(a.view().view(np.uint32), oracle.camera_view(b).view(np.uint32))
        assert np.array_equal(a.projection().view(np.uint32),
                              oracle.camera_projection(b).view(np.uint32))
        assert np.array_equal(a.up().view(np.uint32), oracle.camera_up(b).view(np.uint32))
        a.rotate(0.37, -0.21)
        oracle.camera_rotate(b, 0.37, -0.21)
        assert (a.pitch, a.yaw) == (b.pitch, b.yaw)


def test_camera_defaults_and_speeds():
    cam = rt.Camera.default()
    assert (cam.fov, cam.near_plane, cam.far_plane) == (45.0, np.float32(0.1), 1000.0)
    assert (cam.viewport_width, cam.viewport_height, cam.distance) == (800.0, 600.0, 10.0)
    assert rt.lib.rt_camera_rotation_speed(C.byref(cam)) == np.float32(0.8)
    assert rt.lib.rt_camera_zoom_speed(C.byref(cam)) == np.float32(4.0)   # (10*0.2)^2
    cam.distance = 1000.0
    assert rt.lib.rt_camera_zoom_speed(C.byref(cam)) == 100.0             # capped
    pan = (C.c_float * 2)()
    rt.lib.rt_camera_pan_speed(C.byref(cam), pan)
    x = np.float32(0.8)
    assert np.isclose(pan[0], 0.0366 * x * x - 0.1778 * x + 0.3021, rtol=1e-6)
    # up vector flips sign of the yaw response once the camera is upside down
    cam = rt.Camera.create(90.0, 1.0, 0.01, 10000.0)
    cam.pitch = 3.0
    assert cam.up()[1] < 0
    cam.rotate(1.0, 0.0)
    assert cam.yaw == np.float32(0.8)


def test_make_ubo_sets_viewport_to_image():
    ubo = rt.make_ubo(1920, 1080, 64, 25)
    assert tuple(ubo[32:34]) == (1920.0, 1080.0) and tuple(ubo[40:42]) == (64.0, 25.0)
    assert np.isclose(ubo[16], 1080.0 / 1920.0)


def test_grid_scene_is_deterministic_and_well_formed():
    a = rt.grid_spheres_scene(500)
    b = rt.grid_spheres_scene(500)
    assert np.array_equal(a, b) and a.shape == (500, 16)
    assert tuple(a[0, :4]) == (0.0, -1002.0, -60.0, 1000.0)
    assert (a[1:, 3] >= 0.2).all() and (a[1:, 3] <= 0.5).all()
    # every small sphere rests on the ground sphere
    d = np.linalg.norm(a[1:, :3].astype(np.float64) - a[0, :3].astype(np.float64), axis=1)
    assert np.allclose(d, 1000.0 + a[1:, 3], atol=2e-3)


def test_write_ppm(tmp_path):
    img = np.zeros((2, 3, 4), dtype=np.uint8)
    img[0, :, 0] = 255
    img[1, :, 2] = 128
    p = str(tmp_path / "x.ppm")
    rt.write_ppm(p, img, flip_y=True)
    data = open(p, "rb").read()
    assert data.startswith(b"P6\n3 2\n255\n")
    body = np.frombuffer(data[len(b"P6\n3 2\n255\n"):], dtype=np.uint8).reshape(2, 3, 3)
    assert (body[0, :, 2] == 128).all() and (body[1, :, 0] == 255).all()  # buffer row 1 on top


def test_no_gpu_fails_loudly():
    """Without a device the product raises; it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError):
        rt.Renderer(16, 16)


def test_write_png_round_trips(tmp_path):
    """The PNG writer uses stored deflate blocks; decode it with zlib and compare."""
    import struct
    import zlib
    rng = np.random.default_rng(0)
    img = (rng.random((37, 300, 4)) * 255).astype(np.uint8)   # > 65535 raw bytes: several blocks
    p = str(tmp_path / "x.png")
    rt.write_png(p, img, flip_y=True)
    data = open(p, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(tag + body) == crc
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    assert ihdr == (300, 37, 8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(37, 1 + 300 * 3)
    assert (raw[:, 0] == 0).all()
    assert np.array_equal(raw[:, 1:].reshape(37, 300, 3), img[::-1, :, :3])


def test_public_headers_are_plain_c(tmp_path):
    """include/*.h compiled as strict C99 together with the call sequence a C host makes."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-c",
                          os.path.join(root, "tests", "native", "abi_c99.c"), "-o", str(tmp_path / "abi.o")],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr

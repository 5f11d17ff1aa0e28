"""Host-side readers of the data formats either side of the path: Wavefront OBJ meshes and
baseline JPEG textures (no GPU needed).

JPEG parity is "within a level or two" by nature: decoders differ in inverse-DCT arithmetic
and chroma interpolation.  The fixtures under tests/golden/ were encoded and decoded with
Pillow (libjpeg); tolerance: every channel within 3 levels, mean absolute error < 0.5.
"""
import os

import numpy as np
import pytest

import ray_tracer_amd as rt

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["420", "444_restart", "grey"])
def test_jpeg_decoder_matches_libjpeg_within_tolerance(name):
    want = np.load(os.path.join(GOLDEN, "picture_decoded.npz"))[name]
    data = open(os.path.join(GOLDEN, f"picture_{name}.jpg"), "rb").read()
    got = rt.decode_jpeg(data)
    assert got.shape == (61, 93, 4) and (got[..., 3] == 255).all()
    err = np.abs(got[..., :3].astype(int) - want.astype(int))
    assert err.max() <= 3 and err.mean() < 0.5
    flipped = rt.decode_jpeg(data, flip_y=True)          # stb's flip-on-load
    assert np.array_equal(flipped, got[::-1])


def test_jpeg_decoder_rejects_what_it_cannot_read():
    data = open(os.path.join(GOLDEN, "picture_420.jpg"), "rb").read()
    progressive = data.replace(b"\xff\xc0", b"\xff\xc2", 1)      # SOF0 -> SOF2
    for bad in (b"", b"not a jpeg", data[:200], progressive):
        with pytest.raises(rt.RtError):
            rt.decode_jpeg(bad)


def test_reference_texture_asset_decodes():
    """The reference's only texture (1024x1024 baseline 4:2:0), read where it lies; skipped
    where the reference tree is absent."""
    path = "/root/reference/assets/textures/brickwall.jpg"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    Image = pytest.importorskip("PIL.Image")
    got = rt.decode_jpeg(open(path, "rb").read())
    want = np.asarray(Image.open(path).convert("RGB"))
    assert got.shape == (1024, 1024, 4)
    err = np.abs(got[..., :3].astype(int) - want.astype(int))
    assert err.max() <= 3 and err.mean() < 0.5


OBJ = """# a unit cube: quads, mixed corner syntax, one face with relative indices
o cube
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vt 0 0
vn 0 0 1
f 1 2 3 4
f 5/1 6/1 7/1 8/1
f 1//1 2//1 6//1 5//1
f 2/1/1 3/1/1 7/1/1 6/1/1
f -6 -5 -1 -2
f 4 1 5 8
s off
usemtl none
"""


def test_obj_reader(tmp_path):
    p = tmp_path / "cube.obj"
    p.write_text(OBJ)
    v, i = rt.load_obj(str(p))
    assert v.shape == (8, 3) and v.dtype == np.float32
    assert i.shape == (12, 3) and i.dtype == np.uint32
    assert np.array_equal(v[6], [1, 1, 1])
    assert np.array_equal(i[0], [0, 1, 2]) and np.array_equal(i[1], [0, 2, 3])      # fan of the quad
    assert np.array_equal(i[8], [2, 3, 7]) and np.array_equal(i[9], [2, 7, 6])      # -6 -5 -1 -2
    # every cube face is covered twice by area: total area 6
    a, b, c = v[i[:, 0]], v[i[:, 1]], v[i[:, 2]]
    assert abs(0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum() - 6.0) < 1e-6


def test_obj_reader_errors(tmp_path):
    with pytest.raises(rt.RtError):
        rt.load_obj(str(tmp_path / "missing.obj"))
    for text in ("v 1 2\n", "v 0 0 0\nv 1 0 0\nf 1 2\n", "v 0 0 0\nf 1 2 3\n", "v 0 0 0\nf 0 1 1\n"):
        p = tmp_path / "bad.obj"
        p.write_text(text)
        with pytest.raises(rt.RtError):
            rt.load_obj(str(p))
    p = tmp_path / "empty.obj"
    p.write_text("# nothing\n")
    v, i = rt.load_obj(str(p))
    assert v.shape == (0, 3) and i.shape == (0, 3)


def test_readers_survive_corrupt_input_under_sanitizers(tmp_path):
    """tests/native/ingest_fuzz.cpp built with AddressSanitizer and UBSan (host code only):
    the fixtures decode, thousands of truncated / corrupted JPEGs and random OBJ texts are
    decoded or rejected without a single out-of-bounds access or undefined operation."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ingest_fuzz")
    build = subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
         "-o", exe, os.path.join(root, "tests", "native", "ingest_fuzz.cpp"),
         os.path.join(root, "ray_tracer_amd", "csrc", "rt_ingest.cpp"),
         os.path.join(root, "ray_tracer_amd", "csrc", "rt_host.cpp")],
        capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, GOLDEN, str(tmp_path / "fuzz.obj"), "800"], capture_output=True, text=True,
                         timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "rejected" in run.stdout

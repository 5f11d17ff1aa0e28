"""ctypes binding of librt_hip.so (include/rt_hip.h, include/rt_host.h)."""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# RT_HIP_LIB selects another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")

MODE_MEGAKERNEL = 0
MODE_WAVEFRONT = 1
MODE_MEGAKERNEL_NESTED = 2

FLAG_STRICT_MATH = 1 << 0
FLAG_ACCUM_LINEAR = 1 << 1
FLAG_FORCE_BRUTE = 1 << 2
FLAG_FORCE_BVH = 1 << 3
FLAG_COUNT_TRAVERSAL = 1 << 4


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt_hip error {code}: {message}")
        self.code = code


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C ray_tracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    # One HIP runtime per process.  PyTorch wheels carry a private libamdhip64 (same
    # SONAME as /opt/rocm's); a process that loads both copies loses the GPU in whichever
    # initialises second.  Importing torch first makes librt_hip.so bind to the copy torch
    # uses, so torch tensors, RCCL and these kernels share one runtime and one address space.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    return C.CDLL(LIB_PATH)


lib = _load()


class Camera(C.Structure):
    """rt_camera == the reference's Camera struct (src/camera.h:7-13)."""
    _fields_ = [(n, C.c_float) for n in (
        "fov", "near_plane", "far_plane", "pitch", "yaw", "aspect_ratio", "distance",
        "viewport_width", "viewport_height")]

    @classmethod
    def default(cls) -> "Camera":
        cam = cls()
        lib.rt_camera_init(C.byref(cam))
        return cam

    @classmethod
    def create(cls, fov: float, aspect: float, near: float, far: float) -> "Camera":
        cam = cls()
        lib.rt_camera_create(C.c_float(fov), C.c_float(aspect), C.c_float(near), C.c_float(far),
                             C.byref(cam))
        return cam

    def rotate(self, dx: float, dy: float) -> None:
        lib.rt_camera_rotate(C.byref(self), C.c_float(dx), C.c_float(dy))

    def _vec(self, fn, n: int) -> np.ndarray:
        out = (C.c_float * n)()
        fn(C.byref(self), out)
        return np.array(out, dtype=np.float32)

    def projection(self) -> np.ndarray:
        return self._vec(lib.rt_camera_projection_matrix, 16)

    def view(self) -> np.ndarray:
        return self._vec(lib.rt_camera_view_matrix, 16)

    def orientation(self) -> np.ndarray:
        return self._vec(lib.rt_camera_orientation, 4)

    def up(self) -> np.ndarray:
        return self._vec(lib.rt_camera_up, 3)

    def right(self) -> np.ndarray:
        return self._vec(lib.rt_camera_right, 3)

    def forward(self) -> np.ndarray:
        return self._vec(lib.rt_camera_forward, 3)


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("width", C.c_uint32),
                ("height", C.c_uint32), ("mode", C.c_uint32), ("flags", C.c_uint32),
                ("stream", C.c_void_p), ("accum_buffer", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("frames", C.c_uint64), ("last_render_ms", C.c_float),
                ("total_render_ms", C.c_float), ("last_launches", C.c_uint32),
                ("bvh_nodes", C.c_uint32), ("last_build_ms", C.c_float), ("bvh_height", C.c_uint32),
                ("bvh_nodes_visited", C.c_uint64), ("bvh_spheres_tested", C.c_uint64),
                ("bvh_fallback", C.c_uint32), ("reserved", C.c_uint32)]


lib.rt_last_error.restype = C.c_char_p
lib.rt_version.restype = C.c_char_p
lib.rt_camera_rotation_speed.restype = C.c_float
lib.rt_camera_zoom_speed.restype = C.c_float
lib.rt_default_scene.restype = C.c_uint32
lib.rt_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
lib.rt_destroy.argtypes = [C.c_void_p]
lib.rt_set_params.argtypes = [C.c_void_p, C.c_void_p]
lib.rt_set_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
lib.rt_set_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
lib.rt_set_texture.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]


def _optional(name: str, argtypes) -> None:
    """Entry points a library selected with RT_HIP_LIB may predate (A/B runs against older
    builds): calling one that is missing raises AttributeError at the call, not at import."""
    try:
        getattr(lib, name).argtypes = argtypes
    except AttributeError:
        pass


_optional("rt_set_mesh_device", [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p])
_optional("rt_set_dielectric", [C.c_void_p, C.c_void_p, C.c_uint32])
lib.rt_brick_texture.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
lib.rt_brick_texture.restype = None
lib.rt_ingest_error.restype = C.c_char_p
lib.rt_free.argtypes = [C.c_void_p]
lib.rt_free.restype = None
lib.rt_load_obj.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
lib.rt_decode_jpeg.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p),
                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
lib.rt_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
lib.rt_sync.argtypes = [C.c_void_p]
lib.rt_clear.argtypes = [C.c_void_p]
lib.rt_resolve.argtypes = [C.c_void_p]
lib.rt_read_rgba8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.rt_read_accum_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.rt_read_primary_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.rt_accum_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
lib.rt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
lib.rt_debug_eval.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
lib.rt_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
lib.rt_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]


def _check(rc: int) -> None:
    if rc != 0:
        raise RtError(rc, (lib.rt_last_error() or b"").decode())


def default_ubo(cam: Optional[Camera] = None) -> np.ndarray:
    """The 272-byte uniform block as the reference's main() fills it (68 float32)."""
    ubo = np.zeros(68, dtype=np.float32)
    lib.rt_default_ubo(C.byref(cam) if cam is not None else None, ubo.ctypes.data_as(C.c_void_p))
    return ubo


def ubo_set_camera(cam: Camera, ubo: np.ndarray) -> None:
    assert ubo.dtype == np.float32 and ubo.size == 68 and ubo.flags.c_contiguous
    lib.rt_ubo_set_camera(C.byref(cam), ubo.ctypes.data_as(C.c_void_p))


def default_scene() -> np.ndarray:
    """The reference's three spheres as (3, 16) float32 = 64-byte records."""
    s = np.zeros((3, 16), dtype=np.float32)
    n = lib.rt_default_scene(s.ctypes.data_as(C.c_void_p))
    assert n == 3
    return s


def write_ppm(path: str, rgba8: np.ndarray, flip_y: bool = True) -> None:
    h, w = rgba8.shape[:2]
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if lib.rt_write_ppm(path.encode(), img.ctypes.data_as(C.c_void_p), w, h, int(flip_y)) != 0:
        raise OSError(f"rt_write_ppm({path}) failed")


def write_png(path: str, rgba8: np.ndarray, flip_y: bool = True) -> None:
    h, w = rgba8.shape[:2]
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if lib.rt_write_png(path.encode(), img.ctypes.data_as(C.c_void_p), w, h, int(flip_y)) != 0:
        raise OSError(f"rt_write_png({path}) failed")


def debug_eval(op: int, a: np.ndarray, b: Optional[np.ndarray] = None, device: int = 0) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float32)
        assert b.shape == a.shape
        bp = b.ctypes.data_as(C.c_void_p)
    _check(lib.rt_debug_eval(device, op, a.ctypes.data_as(C.c_void_p), bp,
                             out.ctypes.data_as(C.c_void_p), a.size))
    return out


TEX_SRGB = 1
TEX_MAP_PLANAR_XZ = 0
TEX_MAP_OCTAHEDRAL = 1


class _TextureDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("flags", C.c_uint32), ("mapping", C.c_uint32), ("scale_u", C.c_float),
                ("scale_v", C.c_float), ("first_prim", C.c_uint32), ("n_prims", C.c_uint32)]


def brick_texture(width: int = 256, height: int = 256) -> np.ndarray:
    """The library's procedural brick wall, (height, width, 4) uint8 sRGB."""
    out = np.empty((height, width, 4), dtype=np.uint8)
    lib.rt_brick_texture(width, height, out.ctypes.data_as(C.c_void_p))
    return out


def load_obj(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """(vertices (n,3) float32, indices (m,3) uint32) of a Wavefront OBJ file."""
    v, i = C.c_void_p(), C.c_void_p()
    nv, nt = C.c_uint32(), C.c_uint32()
    if lib.rt_load_obj(os.fsencode(path), C.byref(v), C.byref(nv), C.byref(i), C.byref(nt)) != 0:
        raise RtError(-1, lib.rt_ingest_error().decode())
    try:
        vertices = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_float)), (nv.value * 3,)).copy() \
            if nv.value else np.zeros(0, np.float32)
        indices = np.ctypeslib.as_array(C.cast(i, C.POINTER(C.c_uint32)), (nt.value * 3,)).copy() \
            if nt.value else np.zeros(0, np.uint32)
    finally:
        lib.rt_free(v)
        lib.rt_free(i)
    return vertices.reshape(-1, 3), indices.reshape(-1, 3)


def decode_jpeg(data: bytes, flip_y: bool = False) -> np.ndarray:
    """(h, w, 4) uint8 from the bytes of a baseline JPEG file."""
    out, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
    data = bytes(data)
    buf = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
    if lib.rt_decode_jpeg(buf, len(data), int(flip_y), C.byref(out), C.byref(w), C.byref(h)) != 0:
        raise RtError(-1, lib.rt_ingest_error().decode())
    try:
        img = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()
    finally:
        lib.rt_free(out)
    return img


class Renderer:
    """One rt_ctx: an image of width x height on one GPU."""

    def __init__(self, width: int, height: int, mode: int = MODE_MEGAKERNEL, flags: int = 0,
                 device: int = 0, stream: int = 0, accum_buffer: int = 0):
        self.width, self.height = int(width), int(height)
        cfg = _Config(C.sizeof(_Config), device, self.width, self.height, mode, flags,
                      stream or None, accum_buffer or None)
        self._ctx = C.c_void_p()
        _check(lib.rt_create(C.byref(cfg), C.byref(self._ctx)))

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_params(self, ubo: np.ndarray) -> None:
        ubo = np.ascontiguousarray(ubo, dtype=np.float32)
        assert ubo.nbytes == 272
        _check(lib.rt_set_params(self._ctx, ubo.ctypes.data_as(C.c_void_p)))

    def set_spheres(self, spheres: np.ndarray) -> None:
        spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 16)
        _check(lib.rt_set_spheres(self._ctx, spheres.ctypes.data_as(C.c_void_p), spheres.shape[0]))

    def set_mesh(self, vertices: np.ndarray, indices: np.ndarray, material: np.ndarray) -> None:
        """vertices (n,3) float32, indices (m,3) uint32, material (12,) float32 = colour,
        emissionColour, specularColour."""
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        m = np.ascontiguousarray(material, dtype=np.float32).reshape(12)
        _check(lib.rt_set_mesh(self._ctx, v.ctypes.data_as(C.c_void_p), v.shape[0],
                               i.ctypes.data_as(C.c_void_p), i.shape[0], m.ctypes.data_as(C.c_void_p)))

    def set_mesh_device(self, d_vertices: int, n_vertices: int, d_indices: int, n_triangles: int,
                        material: np.ndarray) -> None:
        """The same from device memory: d_vertices / d_indices are device addresses of
        n_vertices * 3 float32 and n_triangles * 3 uint32."""
        m = np.ascontiguousarray(material, dtype=np.float32).reshape(12)
        _check(lib.rt_set_mesh_device(self._ctx, d_vertices, n_vertices, d_indices, n_triangles,
                                      m.ctypes.data_as(C.c_void_p)))

    def set_texture(self, rgba8: Optional[np.ndarray], first_prim: int = 0, n_prims: int = 1,
                    mapping: int = TEX_MAP_PLANAR_XZ, scale=(1.0, 1.0), srgb: bool = True) -> None:
        """rgba8 (h,w,4) uint8 bound to primitives first_prim .. first_prim+n_prims-1;
        None removes the texture."""
        if rgba8 is None:
            _check(lib.rt_set_texture(self._ctx, None, None))
            return
        t = np.ascontiguousarray(rgba8, dtype=np.uint8)
        if t.ndim != 3 or t.shape[2] != 4:
            raise ValueError("texture must be (height, width, 4) uint8")
        d = _TextureDesc(C.sizeof(_TextureDesc), t.shape[1], t.shape[0], TEX_SRGB if srgb else 0,
                         mapping, float(scale[0]), float(scale[1]), first_prim, n_prims)
        _check(lib.rt_set_texture(self._ctx, C.byref(d), t.ctypes.data_as(C.c_void_p)))

    def set_dielectric(self, ior: Optional[np.ndarray]) -> None:
        """ior (n_spheres,) float32: > 0 makes that sphere a dielectric; None removes the table."""
        if ior is None:
            _check(lib.rt_set_dielectric(self._ctx, None, 0))
            return
        v = np.ascontiguousarray(ior, dtype=np.float32).reshape(-1)
        _check(lib.rt_set_dielectric(self._ctx, v.ctypes.data_as(C.c_void_p), v.size))

    def render(self, first_frame: int = 0, n_frames: int = 1, stride: int = 1) -> None:
        _check(lib.rt_render(self._ctx, first_frame, n_frames, stride))

    def sync(self) -> None:
        _check(lib.rt_sync(self._ctx))

    def clear(self) -> None:
        _check(lib.rt_clear(self._ctx))

    def resolve(self) -> None:
        _check(lib.rt_resolve(self._ctx))

    def read_rgba8(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        _check(lib.rt_read_rgba8(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def read_accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        _check(lib.rt_read_accum_f32(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def read_primary_ids(self) -> np.ndarray:
        out = np.empty((self.height, self.width), dtype=np.int32)
        _check(lib.rt_read_primary_ids(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def accum_device_ptr(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib.rt_accum_device_ptr(self._ctx, C.byref(p), C.byref(n)))
        return int(p.value or 0), int(n.value)

    def stats(self) -> Stats:
        st = Stats()
        _check(lib.rt_get_stats(self._ctx, C.byref(st)))
        return st


STATIC_MAX_SPHERES = 1024
STATIC_MAX_TILES = 139264


class _StaticJob(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("width", C.c_uint32), ("height", C.c_uint32), ("flags", C.c_uint32),
                ("n_spheres", C.c_uint32), ("ubo272", C.c_void_p), ("d_spheres64", C.c_void_p),
                ("first_frame", C.c_uint32), ("n_frames", C.c_uint32), ("frame_stride", C.c_uint32),
                ("reserved", C.c_uint32), ("d_accum", C.c_void_p), ("d_rgba8", C.c_void_p),
                ("d_ids", C.c_void_p), ("d_rays", C.c_void_p)]


_optional("rt_render_static", [C.POINTER(_StaticJob)])


def render_static(width: int, height: int, ubo: np.ndarray, d_spheres64: int, n_spheres: int,
                  d_accum: int, d_rgba8: int, d_ids: int, d_rays: int = 0, first_frame: int = 0,
                  n_frames: int = 1, stride: int = 1, flags: int = 0, device: int = 0,
                  stream: int = 0, reserved: int = 0) -> None:
    """rt_render_static: the entry point that allocates nothing.  The d_* arguments are device
    addresses (for example torch tensors' data_ptr()); ubo is the 272-byte block on the host."""
    ubo = np.ascontiguousarray(ubo, dtype=np.float32)
    assert ubo.nbytes == 272
    job = _StaticJob(C.sizeof(_StaticJob), device, stream or None, width, height, flags, n_spheres,
                     ubo.ctypes.data, d_spheres64 or None, first_frame, n_frames, stride, reserved,
                     d_accum or None, d_rgba8 or None, d_ids or None, d_rays or None)
    _check(lib.rt_render_static(C.byref(job)))


def exported_symbols() -> List[Tuple[str, bool]]:
    """Every function declared in include/*.h paired with whether the library exports it."""
    names: List[str] = []
    inc = os.path.join(_ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(rt_[a-z0-9_]+)\s*\(", text):
            if m.group(1) not in names:
                names.append(m.group(1))
    out = []
    for n in names:
        try:
            getattr(lib, n)
            out.append((n, True))
        except AttributeError:
            out.append((n, False))
    return out

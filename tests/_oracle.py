"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "librt_oracle.so")
L = C.CDLL(SO)

MATH_PORTABLE, MATH_LIBM = 0, 1
ACCUM_GAMMA, ACCUM_LINEAR = 0, 1

L.rto_next_random.restype = C.c_uint32
L.rto_random_value.restype = C.c_float
for _n in ("sin", "cos", "log", "log2", "exp2"):
    f = getattr(L, "rto_pm_" + _n)
    f.restype = C.c_float
    f.argtypes = [C.c_float]
L.rto_pm_pow.restype = C.c_float
L.rto_pm_pow.argtypes = [C.c_float, C.c_float]
L.rto_unorm8.restype = C.c_uint8
L.rto_unorm8.argtypes = [C.c_float]
L.rto_default_spheres.restype = C.c_uint32
L.rto_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                         C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                         C.c_void_p, C.c_void_p, C.c_int]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint32), ("indices", C.c_void_p),
                ("n_triangles", C.c_uint32), ("material", C.c_void_p)]


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("srgb", C.c_int), ("mapping", C.c_int), ("scale_u", C.c_float),
                ("scale_v", C.c_float), ("first_prim", C.c_uint32), ("n_prims", C.c_uint32)]


TEX_MAP_PLANAR_XZ, TEX_MAP_OCTAHEDRAL = 0, 1

L.rto_render_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Mesh),
                               C.POINTER(Texture), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_int]
class Job(C.Structure):
    _fields_ = [("ubo272", C.c_void_p), ("spheres64", C.c_void_p), ("n_spheres", C.c_uint32),
                ("mesh", C.POINTER(Mesh)), ("texture", C.POINTER(Texture)), ("ior", C.c_void_p),
                ("width", C.c_uint32), ("height", C.c_uint32), ("first_frame", C.c_uint32),
                ("n_frames", C.c_uint32), ("frame_stride", C.c_uint32), ("math_mode", C.c_int),
                ("accum_mode", C.c_int), ("pixels", C.c_void_p), ("n_pixels", C.c_uint32),
                ("accum", C.c_void_p), ("rgba8", C.c_void_p), ("ids", C.c_void_p),
                ("rays", C.POINTER(C.c_uint64)), ("threads", C.c_int)]


L.rto_render_job.argtypes = [C.POINTER(Job)]
L.rto_texture_sample.argtypes = [C.POINTER(Texture), C.c_float, C.c_float, C.c_void_p]
L.rto_texture_sample.restype = None
L.rto_srgb_to_linear.argtypes = [C.c_uint8]
L.rto_srgb_to_linear.restype = C.c_float


class Camera(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "fov", "near_", "far_", "pitch", "yaw", "aspect_ratio", "distance", "viewport_width",
        "viewport_height")]


def camera_create(fov, aspect, near, far):
    cam = Camera()
    L.rto_camera_create(C.c_float(fov), C.c_float(aspect), C.c_float(near), C.c_float(far),
                        C.byref(cam))
    return cam


def _vec(fn, cam, n):
    out = (C.c_float * n)()
    fn(C.byref(cam), out)
    return np.array(out, dtype=np.float32)


def camera_view(cam):
    return _vec(L.rto_camera_view, cam, 16)


def camera_projection(cam):
    return _vec(L.rto_camera_projection, cam, 16)


def camera_up(cam):
    return _vec(L.rto_camera_up, cam, 3)


def camera_rotate(cam, dx, dy):
    L.rto_camera_rotate(C.byref(cam), C.c_float(dx), C.c_float(dy))


def mat4_inverse(m):
    m = np.ascontiguousarray(m, dtype=np.float32)
    out = np.empty(16, dtype=np.float32)
    L.rto_mat4_inverse(m.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def default_ubo(cam=None):
    ubo = np.zeros(68, dtype=np.float32)
    L.rto_default_ubo(C.byref(cam) if cam is not None else None, ubo.ctypes.data_as(C.c_void_p))
    return ubo


def default_spheres():
    s = np.zeros((3, 16), dtype=np.float32)
    assert L.rto_default_spheres(s.ctypes.data_as(C.c_void_p)) == 3
    return s


def pcg_stream(seed, n):
    """[(state_after, raw_u32, float_value)] for n successive draws."""
    s = C.c_uint32(seed)
    out = []
    for _ in range(n):
        before = C.c_uint32(s.value)
        raw = L.rto_next_random(C.byref(s))
        val = L.rto_random_value(C.byref(before))
        out.append((s.value, raw, np.float32(val)))
    return out


def pm(name, x, y=None):
    """Elementwise portable-math kernel over numpy arrays (slow Python loop: small inputs)."""
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    f = getattr(L, "rto_pm_" + name)
    flat_x, flat_o = x.ravel(), out.ravel()
    if y is None:
        for i in range(flat_x.size):
            flat_o[i] = f(float(flat_x[i]))
    else:
        flat_y = np.asarray(y, dtype=np.float32).ravel()
        for i in range(flat_x.size):
            flat_o[i] = f(float(flat_x[i]), float(flat_y[i]))
    return out


def render(ubo, spheres, width, height, first_frame=0, n_frames=1, stride=1,
           math_mode=MATH_PORTABLE, accum_mode=ACCUM_GAMMA, threads=8, accum=None, rgba8=None,
           mesh=None, texture=None, ior=None, pixels=None):
    """Returns dict(accum, rgba8, ids, rays).  accum/rgba8 may be passed in to continue.
    mesh = (vertices (n,3) float32, indices (m,3) uint32, material (12,) float32) or None.
    texture = dict(rgba8=(h,w,4) uint8, first_prim, n_prims, mapping, scale=(su,sv), srgb) or None.
    ior = (n_spheres,) float32, > 0 marks a dielectric sphere, or None.
    pixels = 1-D array of pixel indices (y * width + x): only those are shaded and accum,
    rgba8 and ids come back as compact (len(pixels), ...) arrays in list order."""
    ubo = np.ascontiguousarray(ubo, dtype=np.float32)
    spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 16)
    if pixels is not None:
        pixels = np.ascontiguousarray(pixels, dtype=np.uint32).reshape(-1)
        shape = (pixels.size,)
    else:
        shape = (height, width)
    if accum is None:
        accum = np.zeros(shape + (4,), dtype=np.float32)
    if rgba8 is None:
        rgba8 = np.zeros(shape + (4,), dtype=np.uint8)
    ids = np.full(shape, -1, dtype=np.int32)
    rays = C.c_uint64(0)
    if ior is not None:
        ior = np.ascontiguousarray(ior, dtype=np.float32).reshape(-1)
        assert ior.size == spheres.shape[0]
    mptr = None
    if mesh is not None:
        v = np.ascontiguousarray(mesh[0], dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(mesh[1], dtype=np.uint32).reshape(-1, 3)
        m = np.ascontiguousarray(mesh[2], dtype=np.float32).reshape(12)
        mstruct = Mesh(v.ctypes.data, v.shape[0], i.ctypes.data, i.shape[0], m.ctypes.data)
        mptr = C.byref(mstruct)
    tptr = None
    if texture is not None:
        tex = np.ascontiguousarray(texture["rgba8"], dtype=np.uint8)
        scale = texture.get("scale", (1.0, 1.0))
        tstruct = Texture(tex.ctypes.data, tex.shape[1], tex.shape[0], int(texture.get("srgb", True)),
                          int(texture.get("mapping", TEX_MAP_PLANAR_XZ)), float(scale[0]),
                          float(scale[1]), int(texture.get("first_prim", 0)),
                          int(texture.get("n_prims", 1)))
        tptr = C.byref(tstruct)
    job = Job(ubo.ctypes.data, spheres.ctypes.data if spheres.size else None, spheres.shape[0],
              C.pointer(mstruct) if mesh is not None else None,
              C.pointer(tstruct) if texture is not None else None,
              ior.ctypes.data if ior is not None else None, width, height, first_frame, n_frames,
              stride, math_mode, accum_mode, pixels.ctypes.data if pixels is not None else None,
              pixels.size if pixels is not None else 0, accum.ctypes.data, rgba8.ctypes.data,
              ids.ctypes.data, C.pointer(rays), threads)
    rc = L.rto_render_job(C.byref(job))
    assert rc == 0
    return dict(accum=accum, rgba8=rgba8, ids=ids, rays=int(rays.value))


def texture_sample(rgba8, u, v, srgb=False):
    tex = np.ascontiguousarray(rgba8, dtype=np.uint8)
    t = Texture(tex.ctypes.data, tex.shape[1], tex.shape[0], int(srgb), 0, 1.0, 1.0, 0, 1)
    out = np.zeros(3, dtype=np.float32)
    L.rto_texture_sample(C.byref(t), u, v, out.ctypes.data_as(C.c_void_p))
    return out


def dielectric_scatter(direction, normal, ior, exiting, rng_state):
    """(new direction (3,) float32, refracted bool, rng state after the one draw)"""
    d = np.ascontiguousarray(direction, dtype=np.float32)
    n = np.ascontiguousarray(normal, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    state = C.c_uint32(rng_state)
    refracted = C.c_int(0)
    L.rto_dielectric_scatter(d.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), C.c_float(ior),
                             int(exiting), C.byref(state), out.ctypes.data_as(C.c_void_p),
                             C.byref(refracted))
    return out, bool(refracted.value), state.value


def srgb_to_linear(v):
    return float(L.rto_srgb_to_linear(int(v)))


def resolve_rgba8(accum, accum_mode=ACCUM_GAMMA):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    h, w = accum.shape[:2]
    out = np.empty((h, w, 4), dtype=np.uint8)
    L.rto_resolve_rgba8(accum.ctypes.data_as(C.c_void_p), h * w, accum_mode,
                        out.ctypes.data_as(C.c_void_p))
    return out

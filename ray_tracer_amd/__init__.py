"""MI355X-native path tracer: Python binding of the C-ABI module ``librt_hip.so``.

The product is the shared library (``include/rt_hip.h``, ``include/rt_host.h``); this
package is a thin ctypes layer over it for the tests, ``bench.py`` and
``__graft_entry__.py``.  There is no CPU fallback: importing the package without a
built ``librt_hip.so`` raises, and every compute call goes to the HIP kernels.
"""
from ._capi import (  # noqa: F401
    LIB_PATH,
    MODE_MEGAKERNEL,
    MODE_MEGAKERNEL_NESTED,
    MODE_WAVEFRONT,
    FLAG_ACCUM_LINEAR,
    FLAG_COUNT_TRAVERSAL,
    FLAG_FORCE_BRUTE,
    FLAG_FORCE_BVH,
    FLAG_STRICT_MATH,
    Camera,
    Renderer,
    RtError,
    Stats,
    TEX_MAP_OCTAHEDRAL,
    TEX_MAP_PLANAR_XZ,
    brick_texture,
    debug_eval,
    decode_jpeg,
    load_obj,
    default_scene,
    default_ubo,
    exported_symbols,
    render_static,
    STATIC_MAX_SPHERES,
    STATIC_MAX_TILES,
    lib,
    ubo_set_camera,
    write_png,
    write_ppm,
)
from .scenes import grid_spheres_scene, grid_spheres_scene_glass, make_ubo, matte, terrain_mesh  # noqa: F401

__all__ = [
    "LIB_PATH", "MODE_MEGAKERNEL", "MODE_MEGAKERNEL_NESTED", "MODE_WAVEFRONT",
    "FLAG_ACCUM_LINEAR", "FLAG_COUNT_TRAVERSAL", "FLAG_FORCE_BRUTE", "FLAG_FORCE_BVH", "FLAG_STRICT_MATH",
    "Camera", "Renderer", "RtError", "Stats", "debug_eval", "default_scene", "default_ubo",
    "TEX_MAP_OCTAHEDRAL", "TEX_MAP_PLANAR_XZ", "brick_texture",
    "decode_jpeg", "load_obj", "exported_symbols", "render_static", "STATIC_MAX_SPHERES", "STATIC_MAX_TILES", "lib", "ubo_set_camera", "write_png", "write_ppm", "grid_spheres_scene", "grid_spheres_scene_glass", "make_ubo", "matte", "terrain_mesh",
]

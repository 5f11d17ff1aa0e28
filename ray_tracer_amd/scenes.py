"""Synthetic inputs in the reference's own formats (272-byte UBO, 64-byte spheres)."""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from . import _capi


def make_ubo(width: int, height: int, spp: int, bounces: int = 25, *, pitch: float = 0.0,
             yaw: float = 0.0, diverge: float = 1.0, defocus: float = 0.0,
             viewport: Optional[tuple] = None, sun_intensity: float = 0.0) -> np.ndarray:
    """Default render settings (main.cpp:626-636) with the camera viewport set to the image
    size (the reference leaves it at 800x600, camera.h:12) unless ``viewport`` is given."""
    cam = _capi.Camera.create(90.0, 1.0, 0.01, 10000.0)
    vw, vh = viewport if viewport is not None else (width, height)
    cam.viewport_width, cam.viewport_height = float(vw), float(vh)
    cam.pitch, cam.yaw = float(pitch), float(yaw)
    ubo = _capi.default_ubo(cam)
    ubo[40] = float(spp)
    ubo[41] = float(bounces)
    ubo[65] = float(sun_intensity)
    ubo[66] = float(defocus)
    ubo[67] = float(diverge)
    return ubo


class _Pcg:
    """The shader's RNG (ray_tracing.comp:112-119) on Python integers."""

    def __init__(self, seed: int):
        self.state = seed & 0xFFFFFFFF

    def next_u32(self) -> int:
        self.state = (self.state * 747796405 + 2891336453) & 0xFFFFFFFF
        s = self.state
        r = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
        return ((r >> 22) ^ r) & 0xFFFFFFFF

    def value(self) -> float:
        return float(np.float32(self.next_u32()) * np.float32(2.0 ** -32))


def grid_spheres_scene(n_spheres: int = 10000, seed: int = 1337) -> np.ndarray:
    """Ground sphere + (n_spheres-1) small spheres on a jittered square grid in front of the
    origin camera, materials drawn with the shader's PCG.  Returns (n, 16) float32 records.

    Index 0 is the ground: centre (0,-1002,-60), radius 1000, diffuse grey.  The others sit
    on it over x in [-50,50], z in [-110,-10], radius 0.2..0.5; per sphere one draw picks
    diffuse (<0.60), metal (<0.85: smoothness 0.5..1, specular probability 1) or a glossy
    coat (smoothness 1, specular probability 0.15..0.5) -- the reference's material has no
    refraction term, so there is no dielectric class.  Every 50th sphere emits light.
    """
    rng = _Pcg(seed)
    out = np.zeros((n_spheres, 16), dtype=np.float32)
    out[0, 0:4] = (0.0, -1002.0, -60.0, 1000.0)
    out[0, 4:8] = (0.5, 0.5, 0.5, 0.0)
    out[0, 12:16] = (1.0, 1.0, 1.0, 0.0)
    side = max(1, int(math.ceil(math.sqrt(max(1, n_spheres - 1)))))
    cell = 100.0 / side
    for i in range(1, n_spheres):
        gx, gz = (i - 1) % side, (i - 1) // side
        r = rng.value() * 0.3 + 0.2
        r = min(r, 0.45 * cell) if cell < 1.0 else r
        x = -50.0 + (gx + 0.5 + (rng.value() - 0.5) * 0.6) * cell
        z = -110.0 + (gz + 0.5 + (rng.value() - 0.5) * 0.6) * cell
        dist = 1000.0 + r
        y = -1002.0 + math.sqrt(max(0.0, dist * dist - x * x - (z + 60.0) ** 2))
        out[i, 0:4] = (x, y, z, r)
        kind = rng.value()
        rgb = (rng.value(), rng.value(), rng.value())
        if kind < 0.60:
            out[i, 4:8] = (*rgb, 0.0)
            out[i, 12:16] = (1.0, 1.0, 1.0, 0.0)
        elif kind < 0.85:
            out[i, 4:8] = (*rgb, 0.5 + 0.5 * rng.value())
            out[i, 12:16] = (0.5 + 0.5 * rgb[0], 0.5 + 0.5 * rgb[1], 0.5 + 0.5 * rgb[2], 1.0)
        else:
            out[i, 4:8] = (*rgb, 1.0)
            out[i, 12:16] = (1.0, 1.0, 1.0, 0.15 + 0.35 * rng.value())
        if i % 50 == 0:
            out[i, 8:12] = (*rgb, 4.0)
    return out

#!/usr/bin/env python3
"""Known answers for RADIANCE that do not come from the oracle.

Every expected value in closed_form.json is computed here, in NumPy float64, by evaluating
formulas read directly off the reference's shader text (assets/shaders/ray_tracing.comp,
cited per function below) for inputs whose result does not depend on any random draw, or
depends on the draws only through a distribution whose mean has a closed form.  This file
imports neither the oracle (oracle/rt_oracle.c, tests/_oracle.py) nor the HIP library; the
fixtures it writes are what pins BOTH of them to the shader:

    tests/test_closed_form.py        oracle  vs these numbers   (CPU)
    tests/test_gpu_closed_form.py    kernels vs these numbers and vs the oracle (GPU)

Conventions: float literals of the shader (0.4, 0.35, -0.01, 3.1415926 ...) are 32-bit in
GLSL, so they enter as float32 values; all inputs (uniform block, sphere records) are float32
values; the arithmetic on them is float64, i.e. the shader's function evaluated ~9 digits
more precisely than any float32 implementation of it can.  A float32 implementation must
therefore land within a few ulp of these numbers; the bound of each case is stored with it.
Pixels whose outcome hangs on a near-tie (silhouettes, a value half way between two 8-bit
levels) are left out here instead of being tolerated there.

    python tests/golden/make_closed_form.py
"""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def lit(x):
    """A GLSL float literal: the nearest float32, as a Python float."""
    return float(np.float32(x))


# ------------------------------------------------------------------ inputs (data contract)
def perspective_lh_no(fov_deg, aspect, near, far):
    """glm::perspective under GLM_FORCE_LEFT_HANDED, depth -1..1 (src/camera.cpp:21-24,
    CMakeLists.txt:28-30), column-major float32 -- an INPUT of the shader, not part of it."""
    t = math.tan(math.radians(fov_deg) / 2.0)
    m = np.zeros((4, 4))            # m[col][row]
    m[0][0] = 1.0 / (aspect * t)
    m[1][1] = 1.0 / t
    m[2][2] = (far + near) / (far - near)
    m[2][3] = 1.0
    m[3][2] = -(2.0 * far * near) / (far - near)
    return m.astype(np.float32).ravel()


def make_ubo(vw, vh, spp, bounces, ground, horizon, zenith, sun=(0.5, 0.5, 0.5, 0.5), focus=1.0,
             intensity=0.0, defocus=0.0, diverge=0.0, view=None):
    """The 272-byte block of ray_tracing.comp:38-54 as 68 float32."""
    u = np.zeros(68, dtype=np.float32)
    u[0:16] = np.eye(4, dtype=np.float32).ravel() if view is None else view
    u[16:32] = perspective_lh_no(90.0, vw / vh, 0.01, 10000.0)
    u[32:34] = (vw, vh)
    u[40:42] = (spp, bounces)
    u[48:52] = tuple(ground) + (1.0,)
    u[52:56] = tuple(horizon) + (1.0,)
    u[56:60] = tuple(zenith) + (1.0,)
    u[60:64] = sun
    u[64:68] = (focus, intensity, defocus, diverge)
    return u


def sphere(pos, r, colour=(0, 0, 0), smooth=0.0, emission=(0, 0, 0), strength=0.0,
           spec=(0, 0, 0), spec_prob=0.0):
    """One 64-byte record (ray_tracing.comp:30-34 / src/main.cpp:56-66) as 16 float32."""
    return np.array([*pos, r, *colour, smooth, *emission, strength, *spec, spec_prob], dtype=np.float32)


# ------------------------------------------------------------------ the shader in float64
def smoothstep(e0, e1, x):
    t = min(max((x - e0) / (e1 - e0), 0.0), 1.0)
    return t * t * (3.0 - 2.0 * t)


def mix(a, b, t):
    return a * (1.0 - t) + b * t


def environment(d, ubo):
    """getEnvironmentLight, ray_tracing.comp:239-250 (d is NOT normalised there either)."""
    u = ubo.astype(np.float64)
    ground, horizon, zenith, sunpos = u[48:51], u[52:55], u[56:59], u[60:63]
    focus, intensity = u[64], u[65]
    s = smoothstep(0.0, lit(0.4), d[1])
    t1 = s ** lit(0.35) if s > 0.0 else 0.0
    t2 = smoothstep(lit(-0.01), 0.0, d[1])
    sky = mix(horizon, zenith, t1)
    sun = max(0.0, float(np.dot(d, sunpos))) ** focus * intensity
    return mix(ground, sky, t2) + sun * (1.0 if t2 >= 1.0 else 0.0), t1, t2


def primary_direction(x, y, ubo):
    """screenToWorldDirection, ray_tracing.comp:224-237: integer pixel coordinates, the
    viewport (not the image) in the denominator, two general inverses."""
    u = ubo.astype(np.float64)
    view = u[0:16].reshape(4, 4).T          # column-major storage -> mathematical matrix
    proj = u[16:32].reshape(4, 4).T
    ndc = np.array([2.0 * x / u[32] - 1.0, 2.0 * y / u[33] - 1.0])
    eye = np.linalg.inv(proj) @ np.array([ndc[0], ndc[1], -1.0, 1.0])
    eye = np.array([eye[0], eye[1], -1.0, 0.0])
    world = (np.linalg.inv(view) @ eye)[:3]
    return world / np.linalg.norm(world)


class Unstable(Exception):
    """The outcome hangs on a comparison too close to call in float32: leave the pixel out."""


class Rounding:
    """A model of float32 rounding for conditioning estimates: every quantity passed through
    `self(x)` is multiplied by 1 + 2^-24 * N(0, 1), per component.  `EXACT` leaves values alone."""

    def __init__(self, seed=None):
        self.rng = None if seed is None else np.random.default_rng(seed)

    def __call__(self, x):
        if self.rng is None:
            return x
        return x * (1.0 + 2.0 ** -24 * self.rng.standard_normal(np.shape(x)))


EXACT = Rounding()


def closest_hit(o, d, spheres, eps=1e-5, rd=EXACT):
    """raySphere + calculateRayCollision, ray_tracing.comp:145-190: near root only, strict <
    so the lowest index wins ties.  Raises Unstable near a silhouette or a tie."""
    best, best_t, second_t = -1, np.inf, np.inf
    a = float(rd(np.sum(rd(d * d))))
    for i, s in enumerate(spheres.astype(np.float64)):
        oc = rd(o - s[0:3])
        b = 2.0 * float(rd(np.sum(rd(oc * d))))
        c = float(rd(rd(np.sum(rd(oc * oc))) - rd(s[3] * s[3])))
        disc = float(rd(rd(b * b) - rd(4.0 * a * c)))
        scale = b * b + abs(4.0 * a * c)
        if abs(disc) <= eps * scale:
            raise Unstable("grazing")
        if disc < 0.0:
            continue
        t = float(rd(rd(-b - rd(math.sqrt(disc))) / (2.0 * a)))
        span = abs(b) / (2.0 * a) + 1e-30
        if abs(t) <= eps * span and abs(c) > 1e-5 * float(np.dot(oc, oc)):
            raise Unstable("starts on a surface it did not just leave")
        if t < 0.0:
            continue
        if t < best_t:
            best, best_t, second_t = i, t, best_t
        elif t < second_t:
            second_t = t
    if best >= 0 and second_t - best_t <= eps * best_t:
        raise Unstable("two spheres at the same distance")
    return best, best_t


def is_mirror(s):
    return s[7] == 1.0 and s[15] == 1.0       # smoothness 1, specular probability 1


def trace_deterministic(o, d, ubo, spheres, rd=EXACT):
    """trace(), ray_tracing.comp:192-222, for paths on which every scattering surface is a perfect
    mirror: mix(diffuse, reflect(d, n), 1 * 1) = reflect(d, n) whatever the seven draws were, and
    `1 >= randomValue` always holds.  A non-mirror surface may only be the LAST one the bounce
    limit allows (its emission counts, its scattered direction is never used).
    Returns (radiance, primary id, segments traced)."""
    light = np.zeros(3)
    through = np.ones(3)
    first = None
    segments = 0
    for bounce in range(int(ubo[41])):
        segments += 1
        idx, t = closest_hit(o, d, spheres, rd=rd)
        if first is None:
            first = idx
        if idx < 0:
            light = light + environment(rd(d), ubo)[0] * through
            break
        s = spheres[idx].astype(np.float64)
        p = rd(o + rd(d * t))
        n = rd(p - s[0:3])
        n = rd(n / np.linalg.norm(n))
        light = light + s[8:11] * s[11] * through            # :209-211, BEFORE the throughput update
        if not is_mirror(s):
            if bounce != int(ubo[41]) - 1:
                raise Unstable("stochastic surface before the bounce limit")
            break
        through = through * s[12:15]                          # :212-213 with isSpecularBounce = 1
        o, d = p, rd(d - rd(2.0 * float(rd(np.sum(rd(n * d)))) * n))   # reflect(ray.dir, normal), :203
    return light, first, segments


def conditioning(o, d, ubo, spheres, exact_gamma, trials=8):
    """How far float32-sized rounding errors, injected at every operation of the path, move the
    gamma-space pixel: the largest deviation over `trials` noisy evaluations.  inf when a noisy
    path takes another branch."""
    worst = 0.0
    for k in range(trials):
        rd = Rounding(seed=1000 + k)
        try:
            light, _, _ = trace_deterministic(o, rd(d), ubo, spheres, rd=rd)
        except Unstable:
            return np.inf
        worst = max(worst, float(np.max(np.abs(gamma_pixel(light) - exact_gamma))))
    return worst


def unorm8(v, guard=2e-3):
    """imageStore to rgba8: clamp, scale, round to nearest.  None when v*255 is within `guard`
    of a half-way point, where a float32 evaluation may legitimately round the other way."""
    x = min(max(v, 0.0), 1.0) * 255.0
    if abs((x % 1.0) - 0.5) < guard:
        return None
    return int(math.floor(x + 0.5))


def gamma_pixel(radiance_mean):
    """linearToGamma on the mean, ray_tracing.comp:98-101,252."""
    return np.sqrt(np.maximum(radiance_mean, 0.0))


# ------------------------------------------------------------------ cases
# what a float32 evaluation of environment + gamma may be off by when the path itself is exact
BASE_TOL = dict(rel=1e-6, abs=1e-7)
CUSTOM_SKY = dict(ground=(0.35, 0.3, 0.35), horizon=(0.5, 0.6, 0.7), zenith=(0.078, 0.36, 0.72))
DEFAULT_SKY = dict(ground=(0.35, 0.3, 0.35), horizon=(1.0, 1.0, 1.0), zenith=(0.078, 0.36, 0.72))


def default_spheres(mirror=False):
    """src/main.cpp:517-538."""
    if not mirror:
        return np.stack([
            sphere((0, 0, -5), 1, (0.5, 0.5, 0.5), 1.0, spec=(1, 1, 1), spec_prob=0.5),
            sphere((3, 0, -5), 1, (0.8, 0.2, 0.2), 0.5, spec=(1, 1, 1), spec_prob=0.0),
            sphere((0, -101, -5), 100, (0.2, 0.8, 0.05), 0.0, spec=(1, 1, 1), spec_prob=0.0)])
    return np.stack([
        sphere((0, 0, -5), 1, (0.5, 0.5, 0.5), 1.0, spec=(0.9, 0.6, 0.3), spec_prob=1.0),
        sphere((3, 0, -5), 1, (0.8, 0.2, 0.2), 1.0, spec=(0.4, 0.8, 0.7), spec_prob=1.0),
        sphere((0, -101, -5), 100, (0.2, 0.8, 0.05), 1.0, spec=(0.7, 0.5, 0.9), spec_prob=1.0)])


def deterministic_case(name, lines, note, ubo, spheres, w, h, pixels, tol, max_segments=None,
                       max_sensitivity=2e-5):
    """One frame (frame 0, blend weight 1), jitter off: pixel = sqrt(trace), rgba8 = unorm8.
    Each pixel's bound = the case's base tolerance (what a float32 environment / gamma evaluation
    may be off by) + 8 x its conditioning (how far float32 rounding along the path moves it)."""
    rows = []
    for (x, y) in pixels:
        d = primary_direction(x, y, ubo)
        try:
            light, first, segs = trace_deterministic(np.zeros(3), d, ubo, spheres)
        except Unstable:
            continue
        if max_segments is not None and segs > max_segments:
            continue
        g = gamma_pixel(light)
        sens = conditioning(np.zeros(3), d, ubo, spheres, g)
        if not sens <= max_sensitivity:
            continue                                          # ill-conditioned path: says little about correctness
        bound = float(tol["abs"] + tol["rel"] * np.max(np.abs(g)) + 8.0 * sens)
        rows.append(dict(x=int(x), y=int(y), rgb=[float(v) for v in g],
                         rgba8=[unorm8(v, guard=max(2e-3, 4 * 255 * bound)) for v in g] + [255], id=int(first),
                         segments=int(segs) * int(ubo[40]), bound=bound))
    return dict(name=name, kind="deterministic", shader_lines=lines, note=note, width=w, height=h,
                ubo=[float(v) for v in ubo], spheres=[[float(v) for v in s] for s in spheres],
                tol=tol, pixels=rows)


def case_environment():
    """Empty scene: pixel = sqrt(getEnvironmentLight(primary direction)).  Rows 297..299 lie in the
    ground-to-sky band (t2 < 1), 300 is the horizon itself (t1 = 0, t2 = 1), 301..~430 the
    gradient (0 < t1 < 1), above that the zenith colour; the sun term needs t2 >= 1."""
    ubo = make_ubo(800, 600, 1, 25, **CUSTOM_SKY, focus=2.0, intensity=0.25)
    ys = [0, 100, 250, 296, 297, 298, 299, 300, 301, 302, 310, 330, 360, 400, 430, 460, 520, 607]
    xs = [0, 130, 400, 655, 799]
    pix = [(x, y) for y in ys for x in xs]
    c = deterministic_case("environment_regimes", "224-250, 98-109", case_environment.__doc__, ubo,
                           np.zeros((0, 16), np.float32), 800, 608, pix, tol=BASE_TOL)
    regimes = set()
    for r in c["pixels"]:
        _, t1, t2 = environment(primary_direction(r["x"], r["y"], ubo), ubo)
        regimes.add(("t2<1" if 0 < t2 < 1 else "t2=0" if t2 == 0 else "t2=1",
                     "t1=0" if t1 == 0 else "t1=1" if t1 == 1 else "0<t1<1"))
    assert {("t2<1", "t1=0"), ("t2=1", "0<t1<1"), ("t2=1", "t1=1"), ("t2=0", "t1=0"),
            ("t2=1", "t1=0")} <= regimes, regimes
    return c


def case_environment_rotated():
    """The same with a rotated camera: inverse(view) of a general rotation (yaw 0.3, pitch -0.2 as
    src/camera.cpp:27-50 composes them) and the default sky with the sun switched on."""
    ex, ey = 0.2, -0.3                      # euler (-pitch, -yaw, 0)
    cx, sx, cy, sy = math.cos(ex / 2), math.sin(ex / 2), math.cos(ey / 2), math.sin(ey / 2)
    qw, qx, qy, qz = cx * cy, sx * cy, cx * sy, -sx * sy
    rot = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                    [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                    [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]])
    view = np.eye(4)
    view[:3, :3] = rot.T                    # inverse of a rotation
    ubo = make_ubo(800, 600, 1, 25, **DEFAULT_SKY, sun=(0.3, 0.8, -0.5, 1.0), focus=8.0,
                   intensity=0.5, view=view.T.astype(np.float32).ravel())
    pix = [(x, y) for y in range(4, 608, 67) for x in range(3, 800, 113)]
    return deterministic_case("environment_rotated_camera", "224-250", case_environment_rotated.__doc__,
                              ubo, np.zeros((0, 16), np.float32), 800, 608, pix, tol=BASE_TOL)


def case_default_scene_sky():
    """The reference's own window and scene (800x608, viewport 800x600, defaults of
    src/main.cpp:626-636) with jitter off and one bounce: pixels whose primary ray misses show the
    default sky; pixels that hit show black (no emission, throughput never reaches a light)."""
    ubo = make_ubo(800, 600, 1, 1, **DEFAULT_SKY)
    pix = [(x, y) for y in range(2, 608, 23) for x in range(1, 800, 37)]
    return deterministic_case("default_scene_one_bounce", "192-222, 239-250", case_default_scene_sky.__doc__,
                              ubo, default_spheres(), 800, 608, pix, tol=BASE_TOL)


def case_mirror():
    """A perfect mirror (smoothness 1, specular probability 1): pixel = sqrt(specularColour *
    env(reflect(d, n))), independent of all seven draws of the bounce.  One sphere, two samples."""
    ubo = make_ubo(800, 600, 2, 25, **CUSTOM_SKY, focus=2.0, intensity=0.25)
    sph = default_spheres(mirror=True)[:1]
    pix = [(x, y) for y in range(236, 366, 9) for x in range(336, 466, 9)]
    c = deterministic_case("mirror_single_bounce", "145-170, 199-213, 239-250", case_mirror.__doc__, ubo, sph,
                           800, 608, pix, tol=BASE_TOL)
    assert sum(1 for r in c["pixels"] if r["id"] == 0) > 60
    return c


def case_mirror_chain():
    """The default scene's geometry with every sphere a mirror of its own colour: paths of up to
    four segments (sphere -> ground -> sphere -> sky ...), throughput = product of the specular
    colours in path order.  Paths longer than four segments are left out: every reflection off a
    curved surface multiplies the rounding error of the direction."""
    ubo = make_ubo(800, 600, 1, 25, **CUSTOM_SKY, focus=2.0, intensity=0.25)
    pix = [(x, y) for y in range(5, 608, 19) for x in range(7, 800, 29)]
    c = deterministic_case("mirror_chain", "192-222", case_mirror_chain.__doc__, ubo, default_spheres(mirror=True),
                           800, 608, pix, tol=BASE_TOL, max_segments=4)
    assert max(r["segments"] for r in c["pixels"]) == 4
    return c


def case_emission_order():
    """Emission is added BEFORE the throughput is multiplied (:209-213) and the loop ends at the
    bounce limit without an environment term (:196).  Mirror sphere 0 glows (e0 * s0) and tints
    (c0); what it reflects is the glowing ground sphere (e2 * s2) or the sky.  Bounce limit 2:
    L = e0 s0 + c0 * (e2 s2)   on paths sphere 0 -> ground (no sky term: the loop is over),
    L = e0 s0 + c0 * env       on paths sphere 0 -> sky,  L = e2 s2 on ground pixels."""
    ubo = make_ubo(800, 600, 1, 2, **CUSTOM_SKY)
    sph = np.stack([
        sphere((0, 0, -5), 1, (0.5, 0.5, 0.5), 1.0, emission=(0.2, 0.1, 0.05), strength=0.5,
               spec=(0.9, 0.6, 0.3), spec_prob=1.0),
        sphere((0, -101, -5), 100, (0.2, 0.8, 0.05), 0.0, emission=(0.1, 0.3, 0.6), strength=1.25,
               spec=(1, 1, 1), spec_prob=0.0)])
    pix = [(x, y) for y in range(230, 372, 7) for x in range(330, 472, 7)]
    c = deterministic_case("emission_before_throughput", "196, 209-213", case_emission_order.__doc__, ubo, sph,
                           800, 608, pix, tol=BASE_TOL)
    two = [r for r in c["pixels"] if r["id"] == 0 and r["segments"] == 2]
    assert len(two) > 100
    return c


def case_emission_digits():
    """Primary-hit ids read out of the colour channel through the data contract alone: one sample,
    one bounce, black environment, sphere i emits ((r/255)^2, (g/255)^2, (b/255)^2) with (r, g, b) the
    digits of i + 1 -- the stored texel is exactly (r, g, b, 255), a miss is (0, 0, 0, 255)."""
    ubo = make_ubo(800, 600, 1, 1, ground=(0, 0, 0), horizon=(0, 0, 0), zenith=(0, 0, 0))
    geo = [((0, 0, -5), 1), ((3, 0, -5), 1), ((0, -101, -5), 100), ((-2.5, 0.5, -4), 0.75),
           ((-2.0, 0.25, -6), 1.5), ((1.2, 1.4, -3.5), 0.4), ((1.5, 1.5, -7), 1.0),
           ((0.3, -0.6, -2.5), 0.3), ((-4, 3, -9), 2.0), ((4.5, 2.5, -8), 1.25),
           ((0, 0, -5), 0.5), ((0, 0, 6), 2.0)]           # the last two: hidden inside sphere 0 / behind the camera
    sph = []
    for i, (p, r) in enumerate(geo):
        k = i + 1
        digits = (1 + k % 253, 1 + (k * 7) % 253, 1 + (k * 31) % 253)
        sph.append(sphere(p, r, (0.5, 0.5, 0.5), emission=[(v / 255.0) ** 2 for v in digits], strength=1.0))
    sph = np.stack(sph)
    pix = [(x, y) for y in range(0, 608, 11) for x in range(0, 800, 11)] + [(400, y) for y in range(608)]
    c = deterministic_case("emission_digits_primary_ids", "145-190, 196-213, 98-109", case_emission_digits.__doc__,
                           ubo, sph, 800, 608, pix, tol=BASE_TOL)
    for r in c["pixels"]:
        if r["id"] >= 0:
            k = r["id"] + 1
            want = [1 + k % 253, 1 + (k * 7) % 253, 1 + (k * 31) % 253, 255]
            assert r["rgba8"] == want, (r, want)
        else:
            assert r["rgba8"] == [0, 0, 0, 255]
    seen = {r["id"] for r in c["pixels"]}
    assert seen == set(range(-1, 10)), seen               # every visible sphere, no hidden one
    return c


def case_exact_ties():
    """`if (hitInfo.didHit && hitInfo.dst < closestHit.dst)`, ray_tracing.comp:183: among spheres at the
    same distance the strict < keeps the first in index order.  Records with identical geometry give
    bit-identical distances in ANY evaluation, so the answer needs no arithmetic: the scene is the
    emission-digit scene with spheres repeated under other indices and other digits (indices 1 and 4
    are one ball, 2 and 6 another, 3 / 5 / 7 a third, in orders that put the brighter copy first in one
    group and last in another); every pixel shows the digits of the LOWEST index of its group."""
    ubo = make_ubo(800, 600, 1, 1, ground=(0, 0, 0), horizon=(0, 0, 0), zenith=(0, 0, 0))
    balls = {"a": ((0, 0, -5), 1), "b": ((3, 0, -5), 1), "c": ((-2.5, 0.5, -4), 0.75), "ground": ((0, -101, -5), 100)}
    order = ["ground", "a", "b", "c", "a", "c", "b", "c"]
    digits = [(7, 9, 11), (20, 200, 40), (250, 3, 90), (60, 61, 62), (240, 10, 220), (5, 250, 5), (1, 2, 3), (99, 98, 97)]
    full = np.stack([sphere(*balls[k], (0.5, 0.5, 0.5), emission=[(v / 255.0) ** 2 for v in dg], strength=1.0)
                     for k, dg in zip(order, digits)])
    first_of = {}
    for i, k in enumerate(order):
        first_of.setdefault(k, i)
    unique = sorted(first_of.values())                                 # [0, 1, 2, 3]
    pix = [(x, y) for y in range(150, 420, 9) for x in range(60, 760, 9)]
    c = deterministic_case("exact_ties_lowest_index", "172-190", case_exact_ties.__doc__, ubo, full[unique],
                           800, 608, pix, tol=BASE_TOL)
    seen = set()
    for r in c["pixels"]:
        if r["id"] >= 0:
            r["id"] = unique[r["id"]]
            assert r["rgba8"] == list(digits[r["id"]]) + [255], r
            seen.add(r["id"])
    assert seen == set(unique), seen
    c["spheres"] = [[float(v) for v in sp] for sp in full]
    return c


def case_mirror_image_ties():
    """The same strict < between DIFFERENT spheres: two balls that are mirror images of each other in the
    plane x = 0, seen along rays that lie in that plane (the pixel column x = 400 of the 800-wide window:
    ndc.x = 2 * 400 / 800 - 1 = 0, so the direction's x is exactly 0).  x enters a ray's quadratic only
    as oc.x * d.x = +-0 and oc.x^2, so both balls give the same distance bit for bit in any evaluation,
    and the lower index must win: pair one lists the ball at +x first, pair two the ball at -x."""
    ubo = make_ubo(800, 600, 1, 1, ground=(0, 0, 0), horizon=(0, 0, 0), zenith=(0, 0, 0))
    geo = [((0.6, 2.2, -5), 1), ((-0.6, 2.2, -5), 1), ((-0.7, -2.0, -6), 1.25), ((0.7, -2.0, -6), 1.25)]
    digits = [(10, 20, 30), (200, 100, 50), (77, 3, 250), (9, 99, 199)]
    full = np.stack([sphere(p, r, (0.5, 0.5, 0.5), emission=[(v / 255.0) ** 2 for v in dg], strength=1.0)
                     for (p, r), dg in zip(geo, digits)])
    first = [0, 2]                                             # the lower index of each pair
    pix = [(400, y) for y in range(608)]
    assert all(primary_direction(x, y, ubo)[0] == 0.0 for x, y in pix)
    c = deterministic_case("mirror_image_ties_lowest_index", "172-190", case_mirror_image_ties.__doc__, ubo,
                           full[first], 800, 608, pix, tol=BASE_TOL)
    twin = deterministic_case("twin", "", "", ubo, full[[1, 3]], 800, 608, pix, tol=BASE_TOL)
    assert [r["rgb"] == [0, 0, 0] for r in c["pixels"]] == [r["rgb"] == [0, 0, 0] for r in twin["pixels"]]
    seen = set()
    for r in c["pixels"]:
        if r["id"] >= 0:
            r["id"] = first[r["id"]]
            assert r["rgba8"] == list(digits[r["id"]]) + [255], r
            seen.add(r["id"])
    assert seen == {0, 2}, seen
    c["spheres"] = [[float(v) for v in sp] for sp in full]
    return c


def case_constant_radiance():
    """ground = horizon = zenith = c makes the environment the constant c; a lone convex sphere cannot
    be hit twice (near root only, scattered directions point outwards), so EVERY path through an
    interior pixel is sphere -> sky whatever it draws: pixel = sqrt(throughput * c), 2 segments per
    sample.  Three materials: diffuse (throughput = colour); smoothness 0.3 with specular probability
    0.5 and specularColour = colour (either branch, same throughput); with jitter at its default."""
    c = (0.6, 0.5, 0.4)
    out = []
    for name, sp, diverge in [
            ("diffuse", sphere((0, 0, -5), 1, (0.8, 0.5, 0.2), 0.0, spec=(1, 1, 1), spec_prob=0.0), 1.0),
            ("either_branch", sphere((0, 0, -5), 1, (0.3, 0.9, 0.6), 0.3, spec=(0.3, 0.9, 0.6), spec_prob=0.5), 1.0)]:
        ubo = make_ubo(320, 240, 8, 25, ground=c, horizon=c, zenith=c, diverge=diverge)
        rows = []
        for y in range(104, 137, 4):
            for x in range(148, 173, 4):      # |ndc| <= 0.075 * 4/3: well inside the sphere's 0.2 silhouette
                g = gamma_pixel(np.array(sp[4:7], dtype=np.float64) * np.array([lit(v) for v in c]))
                rows.append(dict(x=x, y=y, rgb=[float(v) for v in g], rgba8=[unorm8(v) for v in g] + [255],
                                 id=0, segments=16, bound=float(BASE_TOL["abs"] + BASE_TOL["rel"] * np.max(g))))
        out.append(dict(name="constant_radiance_" + name, kind="deterministic", shader_lines="199-218",
                        note=case_constant_radiance.__doc__, width=320, height=240,
                        ubo=[float(v) for v in ubo], spheres=[[float(v) for v in sp]],
                        tol=BASE_TOL, pixels=rows))
    return out


# ------------------------------------------------------------------ replay of the shader's own random stream
class Pcg:
    """nextRandom / randomValue, ray_tracing.comp:112-119, on Python integers: the stream is exact, and
    randomValue = float(u) / 4294967295.0 with the literal rounding to 2^32 in float32."""

    def __init__(self, state):
        self.state = state & 0xFFFFFFFF

    def value(self):
        self.state = (self.state * 747796405 + 2891336453) & 0xFFFFFFFF
        r = (((self.state >> ((self.state >> 28) + 4)) ^ self.state) * 277803737) & 0xFFFFFFFF
        r = (r >> 22) ^ r
        return float(np.float32(r)) * 2.0 ** -32


def normal_deviate(rng):
    """randomValueNormalDistribution, :121-125: theta is drawn before rho."""
    theta = 2.0 * lit(3.1415926) * rng.value()
    u = rng.value()
    if u <= 0.0:
        raise Unstable("log(0)")
    rho = math.sqrt(-2.0 * math.log(u))
    return rho * math.cos(theta)


def random_direction(rng):
    """randomDirection, :127-132: x, then y, then z."""
    v = np.array([normal_deviate(rng), normal_deviate(rng), normal_deviate(rng)])
    return v / np.linalg.norm(v)


def point_in_circle(rng):
    """randomPointInCircle, :134-138: the angle is drawn before the radius."""
    angle = rng.value() * 2.0 * lit(3.1415926)
    return np.array([math.cos(angle), math.sin(angle)]) * math.sqrt(rng.value())


def trace_replay(o, d, ubo, spheres, rng, rd=EXACT):
    """trace(), :192-222, with the draws of the shader's own stream.  Returns (radiance, primary id,
    segments)."""
    light = np.zeros(3)
    through = np.ones(3)
    first = None
    segments = 0
    for _ in range(int(ubo[41])):
        segments += 1
        idx, t = closest_hit(o, d, spheres, rd=rd)
        if first is None:
            first = idx
        if idx < 0:
            light = light + environment(rd(d), ubo)[0] * through
            break
        s = spheres[idx].astype(np.float64)
        p = rd(o + rd(d * t))
        n = rd(p - s[0:3])
        n = rd(n / np.linalg.norm(n))
        diffuse = rd(n + random_direction(rng))
        diffuse = rd(diffuse / np.linalg.norm(diffuse))                       # :202
        specular = rd(d - rd(2.0 * float(rd(np.sum(rd(n * d)))) * n))          # :203, the INCOMING direction
        is_spec = 1.0 if s[15] >= rng.value() else 0.0                        # :204-205
        a = s[7] * is_spec
        o, d = p, rd(rd(diffuse * (1.0 - a)) + rd(specular * a))              # :201,206-207, not renormalised
        light = light + s[8:11] * s[11] * through                            # :209-211
        through = through * (s[4:7] * (1.0 - is_spec) + s[12:15] * is_spec)  # :212-213
    return light, first, segments


def pixel_replay(x, y, w, h, frame, ubo, spheres, rd=EXACT):
    """main(), :74-101: seed, the sample loop with its two jitters, mean, gamma."""
    rng = Pcg((x * w + y) + int(frame) * 719393)                              # :80-82, x * W + y
    u = ubo.astype(np.float64)
    total = np.zeros(3)
    segments = 0
    first = None
    world = primary_direction(x, y, ubo)
    for _ in range(int(ubo[40])):
        defocus = point_in_circle(rng) * u[66] / w                            # :87-88
        o = u[36:39] + np.array([defocus[0] / w, defocus[1] / h, 0.0])       # :89
        jitter = point_in_circle(rng) * u[67] / w                             # :90-91
        d = rd(world) + np.array([jitter[0], jitter[1], 0.0])                 # :92-93
        light, idx, segs = trace_replay(o, d, ubo, spheres, rng, rd=rd)
        if first is None:
            first = idx                                                       # the id buffer keeps the first sample's hit
        total = total + light
        segments += segs
    return gamma_pixel(total / u[40]), first, segments


def replay_case(name, lines, note, ubo, spheres, w, h, frame, pixels, max_sensitivity=2e-5, trials=6):
    rows = []
    for (x, y) in pixels:
        try:
            g, first, segs = pixel_replay(x, y, w, h, frame, ubo, spheres)
            sens = 0.0
            for k in range(trials):
                gn, fn, sn = pixel_replay(x, y, w, h, frame, ubo, spheres, rd=Rounding(seed=77 + k))
                if fn != first or sn != segs:
                    raise Unstable("a rounding error changes the path")
                sens = max(sens, float(np.max(np.abs(gn - g))))
        except Unstable:
            continue
        if not sens <= max_sensitivity:
            continue
        bound = float(BASE_TOL["abs"] + BASE_TOL["rel"] * np.max(np.abs(g)) + 8.0 * sens)
        weight = lit(1.0 / (frame + 1.0))                     # :106-107 onto a zeroed image: texel = 0 * (1 - w) + pixel * w
        rows.append(dict(x=int(x), y=int(y), rgb=[float(v) for v in g],
                         rgba8=[unorm8(v * weight, guard=max(2e-3, 4 * 255 * bound)) for v in g] + [255],
                         id=int(first), segments=int(segs), bound=bound))
    return dict(name=name, kind="deterministic", shader_lines=lines, note=note, width=w, height=h, frame=int(frame),
                ubo=[float(v) for v in ubo], spheres=[[float(v) for v in s] for s in spheres],
                tol=BASE_TOL, pixels=rows)


def case_replay_lone_sphere():
    """Every sample replayed: the integer stream of the pixel (seed x * W + y + frame * 719393), the four
    jitter draws (defocus 2, diverge 1), the seven draws of the bounce in the shader's order, the `>=`
    specular test, mix() of the diffuse and the mirrored direction with smoothness 0.7, either throughput
    -- evaluated in float64 for a lone sphere, whose scattered ray always escapes.  Frame 3."""
    ubo = make_ubo(96, 64, 16, 25, **CUSTOM_SKY, focus=3.0, intensity=0.4, defocus=2.0, diverge=1.0)
    sp = sphere((0.1, -0.05, -4), 1.25, (0.8, 0.5, 0.2), 0.7, emission=(0.3, 0.2, 0.1), strength=0.25,
                spec=(0.2, 0.9, 0.6), spec_prob=0.4)
    pix = [(x, y) for y in range(0, 64, 5) for x in range(1, 96, 6)] + \
          [(x, y) for y in range(18, 46, 2) for x in range(34, 64, 2)]
    c = replay_case("stream_replay_lone_sphere", "74-101, 112-138, 192-222", case_replay_lone_sphere.__doc__, ubo,
                    sp[None, :], 96, 64, 3, pix)
    hits = sum(1 for r in c["pixels"] if r["id"] == 0)
    assert hits > 30 and len(c["pixels"]) - hits > 100, (hits, len(c["pixels"]))
    return c


def case_replay_default_scene():
    """The same replay for the reference's own scene, window and settings (src/main.cpp:517-538,626-636:
    800x608 image, 800x600 viewport, 25 bounces, diverge 1) at 4 samples per pixel, frame 0: paths bounce
    between the three spheres.  A pixel is kept only if no closest-hit decision on any of its paths is
    near a tie and float32-sized rounding errors injected along the paths move it by < 2e-5."""
    ubo = make_ubo(800, 600, 4, 25, **DEFAULT_SKY, diverge=1.0)
    pix = [(x, y) for y in range(3, 608, 17) for x in range(5, 800, 21)]
    c = replay_case("stream_replay_default_scene", "74-101, 112-138, 145-222", case_replay_default_scene.__doc__,
                    ubo, default_spheres(), 800, 608, 0, pix)
    deep = sum(1 for r in c["pixels"] if r["segments"] >= 12)
    assert len(c["pixels"]) > 800 and deep > 20, (len(c["pixels"]), deep)
    return c


def blend_by_hand(colours, frames):
    """ray_tracing.comp:103-109 on an rgba8 image: old texel (read back as k/255) * (1 - w) + new * w,
    w = 1 / (frame + 1), stored with round-to-nearest.  Returns the texel after each frame, or None
    where a half-way case makes the level ambiguous."""
    old = [0, 0, 0]
    history = []
    for c, f in zip(colours, frames):
        w = 1.0 / (f + 1.0)
        new = []
        for ch in range(3):
            v = (old[ch] / 255.0) * (1.0 - w) + c[ch] * w
            q = unorm8(v)
            if q is None:
                return None
            new.append(q)
        old = new
        history.append(new + [255])
    return history


def case_blend():
    """The 8-bit running mean by hand.  The centre pixel (400, 300) looks straight at sphere 0
    (dst = 4 exactly); with one bounce its colour is sqrt(emission), so changing the emission between
    launches gives a pixel whose per-frame colour is known.  Frames 0, 1, 2 with three emissions, then
    frame 600 with a fourth: by then w = 1/601 < 1/510 and the stored level can no longer move by
    more than it rounds away -- the stall the reference's accumulation has."""
    ubo = make_ubo(800, 600, 1, 1, ground=(0, 0, 0), horizon=(0, 0, 0), zenith=(0, 0, 0))
    gammas = [(0.2, 0.45, 0.91), (0.83, 0.1, 0.61), (1.0, 0.7, 0.03), (0.0, 1.0, 0.0)]
    frames = [0, 1, 2, 600]
    emissions = [[g * g for g in row] for row in gammas]
    # what the shader sees is sqrt(float32(g*g)), not g: evaluate that
    seen = [[math.sqrt(lit(e)) for e in row] for row in emissions]
    history = blend_by_hand(seen, frames)
    assert history is not None
    assert history[0] == [51, 115, 232, 255], history
    # the stall: frame 600's colour is 100+ levels away from the stored one in every channel and moves it by at most one
    assert all(abs(history[3][ch] - history[2][ch]) <= 1 for ch in range(3)), history
    assert all(abs(seen[3][ch] * 255 - history[2][ch]) > 100 for ch in range(3)), history
    steps = []
    for e, f, texel, g in zip(emissions, frames, history, seen):
        steps.append(dict(frame=f, sphere0_emission=[float(np.float32(v)) for v in e], rgba8_after=texel,
                          frame_colour=[float(v) for v in g]))
    return dict(name="eight_bit_blend_by_hand", kind="blend", shader_lines="103-109",
                note=case_blend.__doc__, width=800, height=608, ubo=[float(v) for v in ubo],
                spheres=[[float(v) for v in s] for s in default_spheres()], x=400, y=300, steps=steps,
                accum_sum=[float(sum(s[ch] for s in seen)) for ch in range(3)], tol=BASE_TOL)


def case_blend_clamp():
    """Where the clamp sits, ray_tracing.comp:103-109: the old texel comes back from the rgba8 image
    (<= 1), the new colour enters the weighted sum unclamped, and the store clamps the result.  Frame 0
    stores a colour of (2, 0.6, 3) as (255, 153, 255); frame 1 adds (0.2, 0.6, 0.2) at weight 1/2:
    (1 + 0.2) / 2 = 0.6 -> 153 (an unclamped old value would give 1.1 -> 255); frame 2 adds (4, 0.6, 0.15)
    at weight 1/3: 0.6 * 2/3 + 4/3 > 1 -> 255 (a new colour clamped before the sum would give 187)."""
    ubo = make_ubo(800, 600, 1, 1, ground=(0, 0, 0), horizon=(0, 0, 0), zenith=(0, 0, 0))
    gammas = [(2.0, 0.6, 3.0), (0.2, 0.6, 0.2), (4.0, 0.6, 0.15)]
    frames = [0, 1, 2]
    emissions = [[g * g for g in row] for row in gammas]
    seen = [[math.sqrt(lit(e)) for e in row] for row in emissions]
    history = blend_by_hand(seen, frames)
    assert history is not None
    assert history[0] == [255, 153, 255, 255] and history[1][0] == 153 and history[2][0] == 255, history
    steps = [dict(frame=f, sphere0_emission=[float(np.float32(v)) for v in e], rgba8_after=texel,
                  frame_colour=[float(v) for v in g]) for e, f, texel, g in zip(emissions, frames, history, seen)]
    return dict(name="eight_bit_blend_clamps_at_the_store", kind="blend", shader_lines="103-109",
                note=case_blend_clamp.__doc__, width=800, height=608, ubo=[float(v) for v in ubo],
                spheres=[[float(v) for v in sp] for sp in default_spheres()], x=400, y=300, steps=steps,
                accum_sum=[float(sum(st[ch] for st in seen)) for ch in range(3)], tol=BASE_TOL)


def lobe_mean(n, ubo, grid=1800):
    """E[env(normalize(n + u))] and its per-sample variance for u uniform on the unit sphere --
    the distribution of `normalize(normal + randomDirection)`, ray_tracing.comp:127-132,202
    (a normalised vector of three independent normal deviates is uniform on the sphere).
    Midpoint rule over (cos(theta), phi)."""
    zs = (np.arange(grid) + 0.5) / grid * 2.0 - 1.0
    ph = (np.arange(2 * grid) + 0.5) / (2 * grid) * 2.0 * math.pi
    u64 = ubo.astype(np.float64)
    ground, horizon, zenith = u64[48:51], u64[52:55], u64[56:59]
    acc = np.zeros(3)
    acc2 = np.zeros(3)
    cnt = 0
    for z in zs:
        r = math.sqrt(1.0 - z * z)
        u = np.stack([r * np.cos(ph), np.full_like(ph, z), r * np.sin(ph)], axis=1)
        w = n[None, :] + u
        w = w / np.linalg.norm(w, axis=1, keepdims=True)
        y = w[:, 1]
        s = np.clip(y / lit(0.4), 0.0, 1.0)
        s = s * s * (3.0 - 2.0 * s)
        t1 = np.where(s > 0.0, s ** lit(0.35), 0.0)
        t = np.clip((y - lit(-0.01)) / (0.0 - lit(-0.01)), 0.0, 1.0)
        t2 = t * t * (3.0 - 2.0 * t)
        sky = horizon[None, :] * (1.0 - t1[:, None]) + zenith[None, :] * t1[:, None]
        env = ground[None, :] * (1.0 - t2[:, None]) + sky * t2[:, None]
        acc += env.sum(0)
        acc2 += (env * env).sum(0)
        cnt += env.shape[0]
    mean = acc / cnt
    return mean, acc2 / cnt - mean * mean


def case_statistical():
    """Where the draws matter, their DISTRIBUTION has a closed form.  A lone sphere, jitter off, so the
    hit point and normal n are known; the scattered ray always escapes (convex, near root only).
      diffuse:        E[L] = colour * E_u[env(normalize(n + u))]
      specular mix:   smoothness 0 (direction always diffuse), specular probability p:
                      E[L] = ((1 - p) colour + p specularColour) * E_u[env(...)]   (`p >= u` holds with probability p)
      half mirror:    smoothness 1, probability 1/2: E[L] = (colour * E_u[env] + specularColour * env(reflect(d, n))) / 2
    Sampling plan: 64 progressive frames of 4,096 samples.  A frame's pixel is sqrt(m), m the mean of its
    samples, so E[pixel] = sqrt(mu) * (1 - var / (8 spp mu^2)) to second order, and the mean over the
    frames has standard error sqrt(var / (frames spp)) / (2 sqrt(mu)) (variance from the same integral).
    (One frame of 262,144 samples would not do: the shader sums a pixel's samples in float32, and with
    radiance values that repeat -- zenith, ground -- that running sum drifts by 1e-3 relative at such
    counts; at 4,096 per frame the drift is 2e-5.)  The pixel must agree to `sigmas` standard errors:
    a test of Box-Muller, of the draw order and of the specular test, not of any one draw."""
    spp, frames = 4096, 64
    sky = dict(ground=(0.1, 0.05, 0.02), horizon=(0.9, 0.8, 0.7), zenith=(0.2, 0.4, 1.0))
    out = []
    mats = [("diffuse", sphere((0, 0, -5), 1, (0.8, 0.5, 0.2), 0.0, spec=(1, 1, 1), spec_prob=0.0)),
            ("specular_probability", sphere((0, 0, -5), 1, (0.8, 0.5, 0.2), 0.0, spec=(0.1, 0.9, 0.4), spec_prob=0.3)),
            ("half_mirror", sphere((0, 0, -5), 1, (0.8, 0.5, 0.2), 1.0, spec=(0.1, 0.9, 0.4), spec_prob=0.5))]
    pixels = [(32, 24), (32, 27), (32, 21), (35, 24), (30, 26), (34, 22)]
    for name, sp in mats:
        ubo = make_ubo(64, 48, spp, 25, **sky)
        rows = []
        s64 = sp.astype(np.float64)
        p = s64[15]
        for (x, y) in pixels:
            d = primary_direction(x, y, ubo)
            idx, t = closest_hit(np.zeros(3), d, sp[None, :])
            assert idx == 0
            hit = d * t
            n = (hit - s64[0:3]) / np.linalg.norm(hit - s64[0:3])
            mean_env, var_env = lobe_mean(n, ubo)
            if name == "half_mirror":
                refl = d - 2.0 * float(np.dot(n, d)) * n
                e_r = environment(refl, ubo)[0]
                a, b = s64[4:7] * mean_env, s64[12:15] * e_r
                mean = 0.5 * a + 0.5 * b
                second = 0.5 * s64[4:7] ** 2 * (var_env + mean_env ** 2) + 0.5 * b ** 2
            else:
                tint_mean = (1.0 - p) * s64[4:7] + p * s64[12:15]
                tint_sq = (1.0 - p) * s64[4:7] ** 2 + p * s64[12:15] ** 2
                mean = tint_mean * mean_env
                second = tint_sq * (var_env + mean_env ** 2)
            var = np.maximum(second - mean ** 2, 0.0)
            expect = np.sqrt(mean) * (1.0 - var / (8.0 * spp * mean ** 2))
            sigma = np.sqrt(var / (spp * frames)) / (2.0 * np.sqrt(mean))
            rows.append(dict(x=x, y=y, mean_radiance=[float(v) for v in mean],
                             mean_gamma_pixel=[float(v) for v in expect],
                             standard_error=[float(v) for v in sigma], id=0, segments=2 * spp * frames))
        out.append(dict(name="scatter_distribution_" + name, kind="statistical", shader_lines="119-132, 199-213",
                        note=case_statistical.__doc__, width=64, height=48, ubo=[float(v) for v in ubo],
                        spheres=[[float(v) for v in sp]], frames=frames, sigmas=4.5, abs=2e-5, pixels=rows))
    return out


def main():
    cases = [case_environment(), case_environment_rotated(), case_default_scene_sky(), case_mirror(),
             case_mirror_chain(), case_emission_order(), case_emission_digits(), case_exact_ties(), case_mirror_image_ties()]
    cases += case_constant_radiance()
    cases += [case_replay_lone_sphere(), case_replay_default_scene()]
    cases.append(case_blend())
    cases.append(case_blend_clamp())
    cases += case_statistical()
    for c in cases:                       # rows -> columns: a third of the bytes
        if "pixels" in c:
            rows = c["pixels"]
            c["pixels"] = {k: [r[k] for r in rows] for k in rows[0]}
    path = os.path.join(HERE, "closed_form.json")
    with open(path, "w") as f:
        f.write('{"generator":"tests/golden/make_closed_form.py","cases":[\n')
        f.write(",\n".join(json.dumps(c, separators=(",", ":")) for c in cases))
        f.write("\n]}\n")
    for c in cases:
        n = len(c["pixels"]["x"]) if "pixels" in c else len(c["steps"])
        print(f"{c['name']:44s} {n:5d} rows   lines {c['shader_lines']}")
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

"""GPU: the HIP kernels against radiance known answers that do NOT come from the oracle.

tests/golden/closed_form.json holds float64 evaluations of the reference shader's formulas
(tests/golden/make_closed_form.py; it imports neither the oracle nor this library).  The
kernels, called through the C ABI, must (1) land within each pixel's stored bound in strict
math -- the same bound the oracle is held to in tests/test_closed_form.py, (2) equal the
oracle bit for bit on those pixels in strict math, and (3) stay within the wider bound
written below with the hardware transcendentals of the default fast math.
"""
import numpy as np
import pytest

import ray_tracer_amd as rt
from test_closed_form import (BLENDS, CASES, DETERMINISTIC, STATISTICAL, case_inputs, check_deterministic,
                              check_statistical, pixel_list)

pytestmark = pytest.mark.gpu

MODES = [rt.MODE_MEGAKERNEL, rt.MODE_MEGAKERNEL_NESTED, rt.MODE_WAVEFRONT]

# Fast math: v_sin / v_cos / v_log / v_exp / v_sqrt / v_rsq are good to ~1 ulp each and the hit point
# comes from approximate roots (~2 ulp), so a pixel may sit this much further from the exact value
# than a correctly rounded float32 evaluation: 8 x the strict bound + 4e-6 absolute (gamma space).
FAST_SCALE, FAST_ABS = 8.0, 4e-6


def gpu_pixels(c, mode, flags, frames=1):
    ubo, sph = case_inputs(c)
    w, h = c["width"], c["height"]
    with rt.Renderer(w, h, mode=mode, flags=flags) as r:
        r.set_params(ubo)
        r.set_spheres(sph)
        r.render(c.get("frame", 0), frames, 1)
        r.sync()
        pix = pixel_list(c).astype(np.int64)
        return dict(accum=r.read_accum().reshape(-1, 4)[pix], rgba8=r.read_rgba8().reshape(-1, 4)[pix],
                    ids=r.read_primary_ids().reshape(-1)[pix], rays=int(r.stats().rays))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", DETERMINISTIC)
def test_strict_kernels_against_closed_form_and_oracle(oracle, name, mode):
    c = CASES[name]
    got = gpu_pixels(c, mode, rt.FLAG_STRICT_MATH)
    check_deterministic(c, got["accum"], got["rgba8"], got["ids"])
    ubo, sph = case_inputs(c)
    want = oracle.render(ubo, sph, c["width"], c["height"], first_frame=c.get("frame", 0), pixels=pixel_list(c))
    assert np.array_equal(got["accum"].view(np.uint32), want["accum"].view(np.uint32))
    assert np.array_equal(got["rgba8"], want["rgba8"])
    assert np.array_equal(got["ids"], want["ids"])


@pytest.mark.parametrize("mode", [rt.MODE_MEGAKERNEL, rt.MODE_WAVEFRONT])
@pytest.mark.parametrize("name", DETERMINISTIC)
def test_fast_kernels_against_closed_form(name, mode):
    c = CASES[name]
    got = gpu_pixels(c, mode, 0)
    p = c["pixels"]
    want = np.array(p["rgb"], dtype=np.float64)
    bound = FAST_SCALE * np.array(p["bound"], dtype=np.float64)[:, None] + FAST_ABS
    err = np.abs(got["accum"][:, :3].astype(np.float64) - want)
    worst = int(np.argmax((err / bound).max(axis=1)))
    assert (err <= bound).all(), (name, p["x"][worst], p["y"][worst], got["accum"][worst], want[worst])
    assert np.array_equal(got["ids"], np.array(p["id"], dtype=np.int32))
    levels = np.array([[np.nan if v is None else v for v in row] for row in p["rgba8"]], dtype=np.float64)
    known = ~np.isnan(levels)
    assert np.all(np.abs(got["rgba8"].astype(np.float64) - levels)[known] <= 1)
    assert float((got["rgba8"].astype(np.float64) != levels)[known].mean()) <= 2e-3


@pytest.mark.parametrize("flags", [rt.FLAG_STRICT_MATH, 0])
@pytest.mark.parametrize("name", BLENDS)
def test_eight_bit_blend_by_hand(name, flags):
    """ray_tracing.comp:103-109 with a scene edited between launches; frame 600 shows the stall, the
    second case where the clamp sits."""
    c = CASES[name]
    ubo, sph = case_inputs(c)
    with rt.Renderer(c["width"], c["height"], flags=flags) as r:
        r.set_params(ubo)
        for step in c["steps"]:
            sph[0, 8:11] = step["sphere0_emission"]
            sph[0, 11] = 1.0
            r.set_spheres(sph)
            r.render(step["frame"], 1, 1)
            r.sync()
            assert r.read_rgba8()[c["y"], c["x"]].tolist() == step["rgba8_after"], step["frame"]
        accum = r.read_accum()[c["y"], c["x"]]
    want = np.array(c["accum_sum"])
    assert np.all(np.abs(accum[:3] - want) <= 1e-6 * (1 + want))
    assert accum[3] == len(c["steps"])


@pytest.mark.parametrize("flags", [rt.FLAG_STRICT_MATH, 0])
@pytest.mark.parametrize("mode", [rt.MODE_MEGAKERNEL, rt.MODE_WAVEFRONT])
@pytest.mark.parametrize("name", STATISTICAL)
def test_scatter_distribution(oracle, name, mode, flags):
    """64 frames x 4,096 samples in one launch; the mean over the frames against the closed-form
    expectation, to the same number of standard errors as the oracle."""
    c = CASES[name]
    got = gpu_pixels(c, mode, flags, frames=c["frames"])
    check_statistical(c, got["accum"])
    assert (got["ids"] == 0).all()
    if flags & rt.FLAG_STRICT_MATH and mode == rt.MODE_MEGAKERNEL:
        ubo, sph = case_inputs(c)
        want = oracle.render(ubo, sph, c["width"], c["height"], 0, c["frames"], 1, pixels=pixel_list(c))
        assert np.array_equal(got["accum"].view(np.uint32), want["accum"].view(np.uint32))

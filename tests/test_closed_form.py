"""The CPU oracle against radiance known answers that do NOT come from the oracle.

tests/golden/closed_form.json is written by tests/golden/make_closed_form.py, which evaluates
formulas read off the reference shader (assets/shaders/ray_tracing.comp) in float64 and imports
neither the oracle nor the HIP library.  Here the oracle renders the same inputs and must land
within the bound stored with each case (a few float32 ulp for one-bounce cases, more where
reflections off curved surfaces amplify rounding), and exactly on the 8-bit levels.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    with open(os.path.join(HERE, "golden", "closed_form.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


CASES = load_cases()
DETERMINISTIC = sorted(n for n, c in CASES.items() if c["kind"] == "deterministic")
STATISTICAL = sorted(n for n, c in CASES.items() if c["kind"] == "statistical")


def case_inputs(c):
    ubo = np.array(c["ubo"], dtype=np.float32)
    sph = np.array(c["spheres"], dtype=np.float32).reshape(-1, 16)
    return ubo, sph


def check_deterministic(c, accum, rgba8, ids, rays_per_pixel=None):
    """accum / rgba8 / ids: (n, 4) / (n, 4) / (n,) for the case's pixel list, in list order."""
    p = c["pixels"]
    want = np.array(p["rgb"], dtype=np.float64)
    got = accum[:, :3].astype(np.float64)
    bound = np.array(p["bound"], dtype=np.float64)[:, None]
    err = np.abs(got - want)
    worst = int(np.argmax((err / bound).max(axis=1)))
    assert (err <= bound).all(), (c["name"], p["x"][worst], p["y"][worst], got[worst], want[worst])
    assert (accum[:, 3] == 1).all()
    assert np.array_equal(ids, np.array(p["id"], dtype=np.int32))
    levels = p["rgba8"]
    for i, row in enumerate(levels):
        for ch in range(4):
            if row[ch] is not None:
                assert rgba8[i, ch] == row[ch], (c["name"], p["x"][i], p["y"][i], ch, rgba8[i], row)
    return float((err / bound).max())


def pixel_list(c):
    p = c["pixels"]
    return (np.array(p["y"], dtype=np.int64) * c["width"] + np.array(p["x"], dtype=np.int64)).astype(np.uint32)


@pytest.mark.parametrize("name", DETERMINISTIC)
@pytest.mark.parametrize("math", ["portable", "libm"])
def test_oracle_deterministic_radiance(oracle, name, math):
    c = CASES[name]
    ubo, sph = case_inputs(c)
    pix = pixel_list(c)
    mode = oracle.MATH_PORTABLE if math == "portable" else oracle.MATH_LIBM
    r = oracle.render(ubo, sph, c["width"], c["height"], first_frame=c.get("frame", 0), pixels=pix,
                      math_mode=mode)
    check_deterministic(c, r["accum"], r["rgba8"], r["ids"])
    assert r["rays"] == sum(c["pixels"]["segments"])


BLENDS = sorted(n for n, c in CASES.items() if c["kind"] == "blend")


@pytest.mark.parametrize("name", BLENDS)
def test_oracle_eight_bit_blend_by_hand(oracle, name):
    """ray_tracing.comp:103-109: each launch continues the rgba8 image of the one before."""
    c = CASES[name]
    ubo, sph = case_inputs(c)
    pix = np.array([c["y"] * c["width"] + c["x"]], dtype=np.uint32)
    accum = np.zeros((1, 4), np.float32)
    rgba8 = np.zeros((1, 4), np.uint8)
    for step in c["steps"]:
        sph[0, 8:11] = step["sphere0_emission"]
        sph[0, 11] = 1.0
        r = oracle.render(ubo, sph, c["width"], c["height"], first_frame=step["frame"], pixels=pix,
                          accum=accum, rgba8=rgba8)
        assert r["rgba8"][0].tolist() == step["rgba8_after"], (step["frame"], r["rgba8"][0])
    want = np.array(c["accum_sum"])
    assert np.all(np.abs(accum[0, :3] - want) <= c["tol"]["abs"] * 4 + c["tol"]["rel"] * want * 4)
    assert accum[0, 3] == len(c["steps"])


def check_statistical(c, accum):
    p = c["pixels"]
    mean = np.array(p["mean_gamma_pixel"])
    se = np.array(p["standard_error"])
    assert (accum[:, 3] == c["frames"]).all()
    got = accum[:, :3].astype(np.float64) / c["frames"]  # mean over the frames of sqrt(mean over samples)
    z = np.abs(got - mean) / (se + 1e-30)
    ok = np.abs(got - mean) <= c["sigmas"] * se + c["abs"]
    assert ok.all(), (c["name"], got, mean, z)
    return float(z.max())


@pytest.mark.parametrize("name", STATISTICAL)
def test_oracle_scatter_distribution(oracle, name):
    c = CASES[name]
    ubo, sph = case_inputs(c)
    r = oracle.render(ubo, sph, c["width"], c["height"], 0, c["frames"], 1, pixels=pixel_list(c))
    check_statistical(c, r["accum"])
    assert r["rays"] == sum(c["pixels"]["segments"])
    assert (r["ids"] == 0).all()

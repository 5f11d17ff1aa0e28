"""A bounded, fixed-seed slice of tests/fuzz_parity.py under pytest, so that every `pytest -m gpu`
run carries randomised evidence: strict math bit for bit against the CPU oracle (images, ids,
segment counts; brute force and LBVH; megakernel, nested, wavefront; glass, meshes, textures,
scenes edited between launches), the fast-math invariant (all organisations and searches produce
the same bits), and brute force against LBVH on scenes of thousands of primitives.  The cases are
the generator's own (`random_case`, `deep_case`), drawn with the seeds below; the long hunts stay
a by-hand tool (`python tests/fuzz_parity.py`).  Time budget on an MI355X box: about 20 s.
"""
import time

import numpy as np
import pytest

import fuzz_parity as fz

pytestmark = pytest.mark.gpu

SEED = 20261112
N_RANDOM, N_DEEP = 1000, 100
BUDGET_S = 120.0          # a slow box shortens the slice instead of failing it; the floor below still holds
MIN_RANDOM, MIN_DEEP = 300, 30


def test_random_cases_strict_math_equal_the_oracle():
    t0 = time.monotonic()
    done = 0
    kinds = dict(glass=0, mesh=0, tex=0, lattice=0, fast=0, edited=0)
    for i in range(N_RANDOM):
        if time.monotonic() - t0 > BUDGET_S:
            break
        c = fz.random_case(np.random.default_rng([SEED, i]))
        problems = fz.run_case(c)
        assert not problems, (f"case {i} of seed {SEED}: {c['w']}x{c['h']} spheres {c['sph'].shape[0]} mode {c['mode']} "
                              f"search {c['search']} frames {c['frames']} x {c['calls']}", problems)
        done += 1
        kinds["glass"] += c["ior"] is not None
        kinds["mesh"] += c["mesh"] is not None
        kinds["tex"] += c["tex"] is not None
        kinds["lattice"] += bool(c["lattice"])
        kinds["fast"] += bool(c["fast_check"])
        kinds["edited"] += any(e != "none" for e in c["edits"])
    print(f"{done} random cases in {time.monotonic() - t0:.1f} s: {kinds}")
    assert done >= MIN_RANDOM
    assert all(v > 0 for v in kinds.values())


def test_deep_cases_brute_force_equals_lbvh():
    t0 = time.monotonic()
    done = 0
    tallest = 0
    for i in range(N_DEEP):
        if time.monotonic() - t0 > BUDGET_S:
            break
        c = fz.deep_case(np.random.default_rng([SEED, i, 77]))
        problems, height = fz.run_deep(c)
        assert not problems, (f"deep case {i} of seed {SEED}: {c['kind']} spheres {c['sph'].shape[0]}", problems)
        tallest = max(tallest, int(height))
        done += 1
    print(f"{done} deep cases in {time.monotonic() - t0:.1f} s, tallest tree {tallest}")
    assert done >= MIN_DEEP

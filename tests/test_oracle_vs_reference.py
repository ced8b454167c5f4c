"""Randomised pinning of the oracle against the REFERENCE ITSELF.

Runs only where ``/root/reference`` exists (the build container): the reference's own ``take_turn`` /
``Agent.transition`` / ``Gridworld.move`` / ``visual_field`` are driven through ``oracle/make_golden.py``'s
plugin classes on random shapes, agent counts, radii, rates, seeds and env ids, and every array they produce
is compared with the Python restatement.  Skipped on the GPU box (the reference never travels); the committed
fixtures under ``tests/golden/`` carry the pin there.
"""
import numpy as np
import pytest

from oracle import gridstep_oracle as O
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(), reason="reference tree not present on this machine")


@pytest.fixture(scope="module")
def ref():
    from oracle import make_golden as G

    return G, G._import_reference()


@pytest.mark.parametrize("case", range(10))
def test_treasurehunt_random_configs_match_the_reference(ref, case):
    G, R = ref
    rng = np.random.default_rng(31000 + case)
    h, w = int(rng.integers(6, 20)), int(rng.integers(6, 20))
    r = int(rng.integers(1, min((min(h, w) - 1) // 2, 4) + 1))
    a = int(rng.integers(1, min(7, (h - 2) * (w - 2) // 3) + 1))
    spec = O.treasurehunt_spec(h, w, a, r, spawn_prob=float(rng.choice([0.0, 0.01, 0.08, 0.5])), seed=int(rng.integers(0, 2**40)),
                               dense_prob=float(rng.choice([0.0, 0.0, 0.2, 0.6])))
    ids = [int(v) for v in rng.integers(0, 2**31, size=2)]
    turns, epoch = int(rng.integers(3, 9)), int(rng.integers(0, 12))
    out = G.run_reference_treasurehunt(R, spec, ids, turns, epoch=epoch)
    G.check_against_oracle(spec, ids, turns, out, epoch=epoch)


@pytest.mark.parametrize("case", range(4))
def test_tag_random_configs_match_the_reference(ref, case):
    G, R = ref
    rng = np.random.default_rng(32000 + case)
    h, w = int(rng.integers(6, 13)), int(rng.integers(6, 13))
    r = int(rng.integers(1, min((min(h, w) - 1) // 2, 3) + 1))
    a = int(rng.integers(2, 7))
    spec = G.tag_spec(h, w, a, r, seed=int(rng.integers(0, 2**40)), reward_per_turn=int(rng.choice([10, 1, 3])))
    ids = [int(v) for v in rng.integers(0, 2**31, size=2)]
    turns = int(rng.integers(4, 12))
    out = G.run_reference_tag(R, spec, ids, turns)
    G.check_against_oracle(spec, ids, turns, out)


@pytest.mark.parametrize("case", range(3))
def test_cleanup_random_configs_match_the_reference(ref, case):
    G, R = ref
    rng = np.random.default_rng(33000 + case)
    h, w = int(rng.integers(12, 19)), int(rng.integers(12, 19))
    r = int(rng.integers(1, 4))
    a = int(rng.integers(1, 5))
    spec = G.cleanup_spec(h, w, a, r, seed=int(rng.integers(0, 2**40)), beam_radius=int(rng.integers(1, 4)),
                          pollution_p=float(rng.choice([0.02, 0.1])), apple_p=float(rng.choice([0.01, 0.06])))
    ids = [int(v) for v in rng.integers(0, 2**31, size=2)]
    turns = int(rng.integers(6, 15))
    out = G.run_reference_cleanup(R, spec, ids, turns, initial_apples=int(rng.integers(1, 6)))
    G.check_against_oracle(spec, ids, turns, out, injected=True)

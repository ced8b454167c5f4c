"""sgw_turn_resolve: speculative policy turns of plain movers against the C oracle's sequential turn (the Environment-level cases: test_gpu_turn_loops.py).
(Round 6: regrouped by component from the by-round files of rounds 2-5; no test body changed.)"""
import json  # noqa: F401
import os  # noqa: F401
import subprocess  # noqa: F401
import sys  # noqa: F401

import numpy as np  # noqa: F401
import pytest

from oracle import gridstep_oracle as O  # noqa: F401
from sorrel_amd import _native as N  # noqa: F401
from tests import helpers as H  # noqa: F401
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", SPEC_CASES, ids=[c[0] for c in SPEC_CASES])
def test_speculative_turn_reaches_the_sequential_turn(torch_cuda, case):
    """sgw_turn_resolve through the C ABI: sweep, every pre-move window, one batched policy evaluation, then resolve / re-evaluate the
    dirty rows until nobody is dirty.  Against the C oracle's agent-after-agent turn: (1) stepping the actions the speculation ended
    on gives the engine's grid, positions, rewards and totals; (2) the window each agent had when ITS turn came (the oracle's
    observation) is the row its action was computed on, and the policy of that window is that action -- i.e. the fixed point IS the
    sequential policy-driven turn.  Also: committed envs are skipped, passes stay far below A, every env ends done."""
    name, mk, E, T = case
    most = _speculative_vs_oracle(torch_cuda, name, mk(), E, T)
    assert most <= (5 if name != "crowded_6x6" else 7), most


@pytest.mark.parametrize("case", range(int(os.environ.get("SGW_SOAK", "24"))))
def test_speculative_turn_soak_random_worlds(torch_cuda, case):
    """The same check on random Treasurehunt-like worlds: maps from 5x5 to 90x90 (wave-per-env and workgroup-per-env step kernels; the
    resolve kernel with one wave and with four per env from 16 agents on), 1 ... 64 agents, radii 1 ... 6, sparse to crowded, random
    batch sizes, global env ids and epochs; every eighth case has more than 8 192 envs (dirty list by scan)."""
    from sorrel_amd.spec import treasurehunt_spec

    rng = np.random.default_rng(77000 + case)
    h, w = int(rng.integers(5, 91)), int(rng.integers(5, 91))
    if case % 8 == 7:
        h, w = int(rng.integers(5, 20)), int(rng.integers(5, 20))
    free = (h - 2) * (w - 2)
    a = int(min(rng.integers(1, 65), max(1, free // 3)))
    r = min(int(rng.integers(1, 5 if case % 8 == 7 else 7)), (min(h, w) - 1) // 2)      # (visual_field's own limit)
    ws = treasurehunt_spec(h, w, a, r, spawn_prob=float(rng.choice([0.0, 0.01, 0.1, 0.5])), seed=int(rng.integers(0, 2**31)),
                           dense_prob=float(rng.choice([0.0, 0.2, 0.6])), gem_value=int(rng.integers(1, 20)), bone_value=-int(rng.integers(1, 20)))
    E = int(rng.integers(8193, 9000)) if case % 8 == 7 else int(rng.integers(1, 70))
    T = 2 if case % 8 == 7 else int(rng.integers(2, 7))
    _speculative_vs_oracle(torch_cuda, f"soak {case} ({h}x{w}, {a} agents, r {r}, {E} envs)", ws, E, T, first=int(rng.integers(0, 2**31)),
                           epoch=int(rng.integers(0, 9)), seed=case)

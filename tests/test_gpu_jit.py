"""The in-process specialiser (hipRTC): cache, fallback to the prebuilt instances, refusals, damaged cache files, code objects with more than 64 KiB of LDS.
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


# ------------------------------------------------------------------ the specialiser: cache, fallback, plan == what is created
def test_specialised_instances_are_cached_and_fall_back(torch_cuda, tmp_path):
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(19, 23, 5, 3, spawn_prob=0.05, seed=8)
    N.set_option("jit_cache_dir", str(tmp_path))
    s0 = N.jit_stats()
    e1 = make_engine(ws, 40)
    s1 = N.jit_stats()
    want = N.plan(e1.config)["kernel"]
    assert "specialised=1" in e1.launch_info() and e1.launch_info().startswith(want), e1.launch_info()
    files = sorted(os.listdir(tmp_path))
    assert s1["compiled"] + s1["mem_hits"] + s1["disk_hits"] > s0["compiled"] + s0["mem_hits"] + s0["disk_hits"] and (files or s1["mem_hits"] > s0["mem_hits"])
    e2 = make_engine(ws, 40)                               # the same world again: the loaded function is reused, nothing is compiled
    s2 = N.jit_stats()
    assert s2["compiled"] == s1["compiled"] and s2["mem_hits"] > s1["mem_hits"]
    N.set_option("jit", 0)
    e3 = make_engine(ws, 40)                               # the prebuilt run-time-shape instance
    assert "specialised=0" in e3.launch_info() and "step_fast<true, 2, 6, 0, 0, 0" in e3.launch_info(), e3.launch_info()
    N.set_option("jit", 1)
    N.set_option("jit_cache_dir", "/proc/this/cannot/be/written")   # no disk cache: compiles (or reuses) all the same
    e4 = make_engine(treasurehunt_spec(19, 25, 5, 3, spawn_prob=0.05, seed=8), 40)
    assert "specialised=1" in e4.launch_info()
    engines = [e1, e2, e3]
    co = H.COracle(ws, 40, first_env_id=0)
    for e in engines:
        e.reset(0)
    co.reset(0)
    for t in range(1, 6):
        assert co.step(0, t, random_actions=True) == 0
        for e in engines:
            e.step(random_actions=True)
            assert_same(e, co, ctx=f"turn {t} {e.launch_info().split(' group')[0]}")
    with pytest.raises(ValueError):
        N.set_option("no_such_key", 1)
    with pytest.raises(ValueError):
        N.set_option("group", 48)
    with pytest.raises(ValueError):
        N.set_option("group", 16, engine=e1._h)           # a plan-shaping key on a live engine
    N.set_option("rows_mode", 1, engine=e1._h)             # a live key


def test_specialised_step_big_with_more_than_64_kib_of_lds(torch_cuda):
    """A 180x200x2 world keeps 72 KB per env in LDS: the specialised step_big instance (loaded as a module function) gets that much
    dynamic LDS, and agrees with the oracle and with the prebuilt instance."""
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(180, 200, 6, 4, spawn_prob=0.03, seed=12, dense_prob=0.1)
    for jit in (1, 0):
        N.set_option("jit", jit)
        eng, co = make_engine(ws, 6, first=1), H.COracle(ws, 6, first_env_id=1)
        assert "step_big<" in eng.launch_info() and f"specialised={jit}" in eng.launch_info() and int(eng.launch_info().split("lds=")[1].split()[0]) > 65536, eng.launch_info()
        eng.reset(0)
        co.reset(0)
        for t in range(1, 4):
            eng.step(random_actions=True)
            assert co.step(0, t, random_actions=True) == 0
            assert_same(eng, co, ctx=f"jit={jit} turn {t}")
        assert eng.status() == 0


def test_a_refused_side_kernel_replans_the_whole_engine_for_the_prebuilt_instances(torch_cuda, tmp_path):
    """Round-4 advisor: only the whole-turn kernel used to be compiled at sgw_create; when the rollout instance (or any other) was refused
    later, its prebuilt run-time-shape twin ran under a plan laid out for compile-time shapes -- for a world with C % 4 != 0 (Cleanup:
    nine channels) the twin's grouped plane writes then ran past the staging area.  Now every instance is resolved at create and ANY
    refusal re-plans with jit = 0.  The Cleanup fixture's world: rollout instance refused -> prebuilt plan, results = the oracle."""
    torch = torch_cuda
    from sorrel_amd import _native as N
    from tests.test_gpu_parity import make_engine

    d, spec = H.load_golden("cleanup_15x16")
    ws = H.world_spec(spec)
    E = 24
    N.set_option("jit_cache_dir", str(tmp_path))
    plan = N.plan(ws.to_config(E, 0))
    assert plan["specialised"] == 1 and ws.num_channels % 4 != 0 and plan["kernel_rollout"] != plan["kernel"]
    good = make_engine(ws, E)
    assert "specialised=1" in good.launch_info()
    _rollout_vs_oracle(torch, ws, good, E, d)
    N.set_option("jit_refuse", plan["kernel_rollout"])
    eng = make_engine(ws, E)
    assert "specialised=0" in eng.launch_info(), eng.launch_info()          # the WHOLE engine runs the prebuilt plan
    with N.options(jit=0):
        assert eng.launch_info().split(" group")[0] == N.plan(ws.to_config(E, 0))["kernel"]
    _rollout_vs_oracle(torch, ws, eng, E, d)
    N.set_option("jit_refuse", None)
    # a row kernel that exists only specialised: refused -> the capability is not advertised (it used to be, and the first call failed)
    from sorrel_amd.spec import treasurehunt_spec

    own = treasurehunt_spec(20, 22, 3, 4, spawn_prob=0.05, seed=2)
    p2 = N.plan(own.to_config(E, 0))
    assert p2["kernel_observe_rows"].startswith("observe_rows<")
    N.set_option("jit_refuse", "observe_rows<")
    e2 = make_engine(own, E)
    assert "specialised=0" in e2.launch_info() and not (e2.capabilities() & N.CAP_OBSERVE_ROWS)
    N.set_option("jit_refuse", None)
    e3 = make_engine(own, E)
    assert "specialised=1" in e3.launch_info() and (e3.capabilities() & N.CAP_OBSERVE_ROWS)


def test_a_damaged_cache_file_is_recompiled_not_remembered_as_a_failure(torch_cuda, tmp_path):
    """A cached code object that does not load (truncated, bit-flipped, someone else's) is dropped and the instance compiled again on the
    spot -- the engine still runs its specialised plan -- and a file with a wrong checksum is never handed to hipModuleLoadData."""
    torch = torch_cuda
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec
    from tests.test_gpu_parity import make_engine, assert_same

    N.set_option("jit_cache_dir", str(tmp_path))
    ws = treasurehunt_spec(17, 27, 4, 3, spawn_prob=0.05, seed=9)      # a shape no other test uses: nothing of it is loaded yet
    inst = N.plan(ws.to_config(16, 0))["kernel"]
    path = N.jit_compile(inst)
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0x5A                                       # flip a byte inside the code object: the checksum no longer matches
    open(path, "wb").write(bytes(blob))
    s0 = N.jit_stats()
    eng = make_engine(ws, 16)
    s1 = N.jit_stats()
    # (the damaged whole-turn instance is compiled again; the engine's other instances -- rollout, ... -- are new to this cache as well)
    assert "specialised=1" in eng.launch_info() and s1["compiled"] >= s0["compiled"] + 1 and s1["failed"] == s0["failed"]
    lowered, code = N.jit_code_object(path)                            # rewritten whole
    assert code[:4] == b"\x7fELF"
    co = H.COracle(ws, 16, first_env_id=0)
    eng.reset(0)
    co.reset(0)
    for t in range(1, 4):
        assert co.step(0, t, random_actions=True) == 0
        eng.step(random_actions=True)
        assert_same(eng, co, ctx=f"turn {t}")

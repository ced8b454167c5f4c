"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol
include/sgw.h declares, and the ctypes mirror of sgw_config has the C layout.
No compute calls are made (there is no GPU in the build container)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from tests import helpers as H
from sorrel_amd import _native as N

HEADER = os.path.join(H.ROOT, "include", "sgw.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgw_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_what_binding_expects():
    assert set(declared_symbols()) == set(N.EXPORTS)


def test_flag_constants_match_the_header():
    """The ctypes binding's flag / rule constants are the header's macros (the header is the interface)."""
    text = open(HEADER).read()
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(SGW_[A-Z_0-9]+)\s+(\d+)u?\b", text)}
    for name in ("STEP_SWEEP", "STEP_RANDOM_ACTIONS", "STEP_NO_OBS", "STEP_OBS_NEXT", "STEP_OBS_NEXT_PACKED", "STEP_NO_MOVE", "STEP_OBS_AGENT_MAJOR", "CAP_OBSERVE_ROWS", "CAP_ACT", "CAP_RESOLVE", "CAP_OBS_AGENT_MAJOR", "CAP_SWEEP_ROWS"):
        assert getattr(N, name) == macros["SGW_" + name], name
    flags = [macros[k] for k in macros if k.startswith("SGW_STEP_") and k != "SGW_STEP_DEFAULT"]
    assert len(set(flags)) == len(flags) and all(f & (f - 1) == 0 for f in flags)      # distinct single bits
    for name in ("ACT_U8", "ACT_I32", "ACT_I64", "ACT_QF32", "TAIL_NONE", "TAIL_AGENT_IS_IT", "TAIL_POSITION_TABLE"):
        assert getattr(N, name) == macros["SGW_" + name], name
    # the counter-RNG streams: the oracle's numbering is the header's (tests compare draws made by either)
    from oracle import gridstep_oracle as O
    for name in ("SPAWN", "SPAWN_KIND", "ACTION", "PLACE", "DENSE", "DENSE_KIND", "TAG_INIT", "EXPLORE"):
        assert getattr(O, "STREAM_" + name) == macros["SGW_STREAM_" + name], name


def test_value_action_is_argmax_or_the_engines_uniform_draw():
    """oracle.value_action -- the checker of SGW_ACT_QF32 -- against its definition: np.argmax (first maximum; NaN counts as one)
    unless u32(EXPLORE, agent) < floor(epsilon * 2**32), then the action random_actions takes for that (env, turn, agent)
    (sorrel/models/pytorch/iqn.py:294-309 with the counter RNG in place of `random`)."""
    import numpy as np
    from oracle import gridstep_oracle as O

    spec = O.treasurehunt_spec(9, 9, 3, 2, seed=77)
    nact = len(spec.action_dy)
    rng = np.random.default_rng(0)
    assert O.value_action(spec, 0, 0, 1, 0, [1.0, 3.0, 3.0, 2.0]) == 1                 # first maximum
    assert O.value_action(spec, 0, 0, 1, 0, [1.0, np.nan, 9.0, np.nan]) == 1           # NaN counts as the maximum
    assert O.value_action(spec, 0, 0, 1, 0, [-np.inf] * nact) == 0
    explored = 0
    for env in range(40):
        for turn in range(1, 6):
            q = rng.standard_normal(nact).astype(np.float32)
            rnd = O.random_actions(spec, env, 3, turn)
            for a in range(spec.num_agents):
                assert O.value_action(spec, env, 3, turn, a, q, 0.0) == int(np.argmax(q))
                assert O.value_action(spec, env, 3, turn, a, q, 1.0) == int(rnd[a])
                u = int(O.rng_u32(spec.seed, env, 3, turn, O.STREAM_EXPLORE, a))
                want = int(rnd[a]) if u < O.prob_threshold(0.3) else int(np.argmax(q))
                assert O.value_action(spec, env, 3, turn, a, q, 0.3) == want
                explored += u < O.prob_threshold(0.3)
    assert 120 < explored < 240                                                         # ~0.3 of 600 draws


def test_library_exports_every_declared_symbol(built):
    lib = N.load()   # dlopen only (after torch, so one HIP runtime is mapped): no HIP call, works without a GPU
    for name in declared_symbols():
        assert hasattr(lib, name), f"libsgw.so does not export {name}"


def test_library_has_gfx950_code_object(built):
    data = open(N.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    assert b"step_kernel" in data


def test_config_struct_layout_matches_c(tmp_path):
    src = tmp_path / "layout.c"
    fields = [f[0] for f in N.SgwConfig._fields_]
    body = "".join(f'printf("{f} %zu\\n", offsetof(sgw_config, {f}));\n' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sgw.h"\nint main(){\n'
                   'printf("sizeof %zu\\n", sizeof(sgw_config));\n' + body + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(H.ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(out["sizeof"]) == C.sizeof(N.SgwConfig)
    for f in fields:
        assert int(out[f]) == getattr(N.SgwConfig, f).offset, f


def test_pure_host_helpers_agree_with_spec(built):
    from sorrel_amd.spec import treasurehunt_spec

    lib = N.load()
    for (h, w, a, r, want) in ((16, 16, 4, 2, 3476), (32, 32, 8, 3, 13592), (128, 128, 64, 5, 251984)):
        spec = treasurehunt_spec(h, w, a, r)
        cfg = spec.to_config(1)
        # SURVEY.md 8(d) per-env-step algorithmic bytes for C2 / C3 / C5
        assert lib.sgw_algorithmic_bytes_per_env_step(C.byref(cfg)) == want == spec.algorithmic_bytes_per_env_step()
        assert lib.sgw_grid_bytes_per_env(C.byref(cfg)) == 2 * h * w
        assert lib.sgw_obs_elems_per_env(C.byref(cfg)) == a * 6 * (2 * r + 1) ** 2


def test_engine_fails_loudly_without_gpu():
    import torch

    from sorrel_amd.engine import GridEngine
    from sorrel_amd.spec import treasurehunt_spec

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(N.SgwError):
        GridEngine(treasurehunt_spec(16, 16, 4, 2), 8, device="cuda")
    with pytest.raises(N.SgwError):
        GridEngine(treasurehunt_spec(16, 16, 4, 2), 8, device="cpu")


def test_missing_library_fails_loudly(monkeypatch):
    """No built extension -> SgwError naming the build command; nothing else is tried."""
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libsgw.so")
    with pytest.raises(N.SgwError, match="no CPU fallback"):
        N.load()

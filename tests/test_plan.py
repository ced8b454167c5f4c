"""sgw_plan on the CPU: the dispatcher's whole decision -- kernel family, lanes per env, LDS layout, staging, walk window, the
instances -- is a pure function of (config, options, CUs, LDS), enumerated here for every BASELINE config, the shapes measured in
profiles/ and both sides of every threshold, and pinned by tests/golden/plans.json: a plan that changes unannounced fails."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_plans_match_the_pinned_ones(built):
    import plan_cases

    with open(plan_cases.GOLDEN) as fh:
        want = json.load(fh)
    got = plan_cases.plans()
    assert sorted(got) == sorted(want), "cases added / removed: run `python tools/plan_cases.py --update` and commit the diff"
    changed = {k: {f: (want[k][f], got[k][f]) for f in got[k] if got[k][f] != want[k].get(f)} for k in got if got[k] != want[k]}
    assert not changed, ("the dispatcher's plan changed for these configurations (field: pinned -> now); if that is intended, run "
                         "`python tools/plan_cases.py --update` and commit the diff: " + json.dumps(changed, indent=1)[:4000])


def test_plan_is_pure_and_says_what_matters(built):
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec

    c3 = treasurehunt_spec(32, 32, 8, 3).to_config(65536, 0)
    a, b = N.plan(c3), N.plan(c3)
    assert a == b and a["family"] == N.FAMILY_WAVE and a["kernel"] == "step_fast<true, 2, 6, 3, 32, 32>" and a["whole_env_burst"] == 1
    assert a["lds_bytes"] == 4 * (512 + 2048 + 2352) and a["grid_blocks"] == 65536 // 4
    c5 = treasurehunt_spec(128, 128, 64, 5).to_config(2048, 0)
    p5 = N.plan(c5)
    assert p5["family"] == N.FAMILY_WORKGROUP and p5["threads"] == 512 and p5["walk_blocks"] == 3 * 256 and p5["walk_min_envs"] < 2048 <= p5["walk_max_envs"]
    assert N.plan(c5, num_cus=128)["walk_blocks"] == 3 * 128                       # a function of the device description it is given
    own = treasurehunt_spec(32, 33, 8, 3).to_config(65536, 0)
    assert N.plan(own)["kernel"] == "step_fast<true, 2, 6, 3, 32, 33>" and N.plan(own)["specialised"] == 1
    with N.options(jit=0):                                                       # hipRTC absent: the prebuilt run-time-shape instance
        p = N.plan(own)
        assert p["specialised"] == 0 and p["kernel"] == p["kernel_prebuilt"] == "step_fast<true, 2, 6, 0, 0, 0, false, false, true>"
    with pytest.raises(ValueError):
        N.set_option("no_such_key", 1)
    bad = treasurehunt_spec(32, 32, 8, 3).to_config(65536, 0)
    bad.vision_radius = 40
    with pytest.raises(ValueError):
        N.plan(bad)


def test_the_shipped_library_reads_one_environment_variable(built):
    """sgw_set_option replaced ~30 getenv hooks: the library source asks for SGW_DEBUG and nothing else."""
    import re

    text = ""
    for name in os.listdir(os.path.join(ROOT, "sorrel_amd", "csrc")):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(ROOT, "sorrel_amd", "csrc", name)) as fh:
                text += fh.read()
    calls = re.findall(r'getenv\("([A-Z_0-9]+)"\)', text)
    # SGW_DEBUG: the specialiser's log lines.  ROCM_PATH: which installation's hipRTC (jit.h).  No dispatcher knob is an environment variable.
    assert sorted(calls) == ["ROCM_PATH", "SGW_DEBUG"], calls
    # ... and one getenv by variable: where code objects are cached when the package directory is not the user's own (jit_default_cache_dir)
    assert re.findall(r'getenv\((\w+)\)', text) == ["var"] and '{"XDG_CACHE_HOME", "HOME"}' in text


def test_a_specialised_instance_compiles_without_a_device(built, tmp_path):
    """sgw_jit_compile: the library's embedded kernel sources + hipRTC produce the code object of a plan's kernel with no GPU in the
    machine (pre-filling a cache on a build host; tools/jit_regs.py).  The file carries the lowered name and an ELF; a second call is a
    cache hit; the instance uses no scratch."""
    import re
    import subprocess

    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec

    plan = N.plan(treasurehunt_spec(14, 18, 3, 2).to_config(1000, 0))
    assert plan["specialised"] == 1 and plan["kernel"].startswith("step_")
    with N.options(jit_cache_dir=str(tmp_path)):
        s0 = N.jit_stats()
        path = N.jit_compile(plan["kernel"])
        s1 = N.jit_stats()
        again = N.jit_compile(plan["kernel"])
        s2 = N.jit_stats()
    assert path == again and os.path.dirname(path) == str(tmp_path)
    assert s1["compiled"] == s0["compiled"] + 1 and s2["compiled"] == s1["compiled"] and s2["disk_hits"] == s1["disk_hits"] + 1
    lowered, code = N.jit_code_object(path)
    assert lowered.startswith("_Z") and code[:4] == b"\x7fELF"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if os.path.exists(readelf):
        obj = tmp_path / "k.hsaco"
        obj.write_bytes(code)
        notes = subprocess.run([readelf, "--notes", str(obj)], capture_output=True, text=True).stdout
        meta = next(b for b in notes.split("- .agpr_count") if lowered in b)
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta)[1]) == 0, "the specialised instance spills to scratch"
    with pytest.raises(N.SgwError):
        N.jit_compile("step_fast<true, 2, 6, 3, 32>")          # not an instance of the template: hipRTC's error comes back


def _canonical(name: str, lanes: int):
    """A plan's kernel name -> (template, full argument tuple) as `nm -C` spells the instance: trailing defaults filled in, the
    SGW_AGENT_RULE_* macros and the packed kernels' `G` resolved, `(turn loop)` = the MULTI argument."""
    import re

    multi = name.endswith(" (turn loop)")
    name = name.replace(" (turn loop)", "")
    m = re.fullmatch(r"(\w+)<(.*)>", name)
    assert m, name
    tmpl, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    rules = {"SGW_AGENT_RULE_MOVE": "0", "SGW_AGENT_RULE_TAG": "1", "SGW_AGENT_RULE_CLEANUP": "2", "SGW_MAX_AGENTS": str(__import__("sorrel_amd._native", fromlist=["x"]).MAX_AGENTS)}
    args = [rules.get(a, str(lanes) if a == "G" else a) for a in args]
    # (round 6: step_big has a ninth argument -- ROWS --, step_kernel a tenth -- the capacity of its per-agent LDS arrays -- and an eleventh -- ROWS)
    defaults = {"step_fast": ["?"] * 6 + ["false"] * 6, "step_big": ["?"] * 4 + ["false", "false", "false", "512", "false"],
                "step_kernel": ["?", "?", "0", "0", "0", "0", "0", "0", "false", "64", "false"]}.get(tmpl)
    if defaults:
        args += defaults[len(args):]
        if multi:
            args[{"step_kernel": 8, "step_big": 4}.get(tmpl, -1)] = "true"
    return tmpl, tuple(args)


def test_every_plan_without_specialised_instances_names_kernels_the_library_holds(built):
    """hipRTC absent (option jit = 0): whatever a plan of tools/plan_cases.py says will run -- whole turn, direct-store twin, rollout,
    walking variant, phase and row kernels -- is an instance compiled into libsgw.so (round 5 removed 26 turn-loop instances that
    spilled to scratch: nothing may still point at one)."""
    import re
    import subprocess

    import plan_cases
    from sorrel_amd import _native as N

    out = subprocess.run(["nm", "-C", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
    have = set()
    for line in out.splitlines():
        m = re.search(r"__device_stub__(\w+)<(.*)>\(", line.replace("(anonymous namespace)::", ""))
        if m:
            have.add((m.group(1), tuple(a.strip() for a in m.group(2).split(","))))
    assert len(have) > 100
    checked = 0
    for case, plan in plan_cases.plans().items():
        if plan["specialised"]:
            continue
        for key in ("kernel", "kernel_plain", "kernel_rollout", "kernel_walk", "kernel_phase", "kernel_observe_rows"):
            name = plan[key]
            if "<" not in name:                   # "-", "the step kernel"
                continue
            assert _canonical(name, plan["lanes_per_env"]) in have, (case, key, name)
            checked += 1
    assert checked > 100
    # ... and the count the round ended on (tools/regs.py lists them with registers / scratch): 177 before, 151 now
    assert len([h for h in have if h[0].startswith("step_")]) == 99      # (round 5: + step_fast_rows<2, 6, 3, 32, 32>; round 6: + six run-time-shape step_kernel<256, ..., 128> for > 64 agents)

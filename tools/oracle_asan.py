"""CPU sanitizer pass over the C oracle (make -C oracle asan): every world fixture and 40 random rule worlds
under AddressSanitizer + UBSan.  GPU sanitizers are not available on the pool; this covers the checker."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
from tests import helpers as H
import tests.helpers as TH
lib = C.CDLL("oracle/libgridstep_oracle_asan.so")
TH._ORACLE = lib
TH.oracle_lib = lambda: lib
n = 0
for name in [n for n in H.golden_names() if n != "stock_np_random"]:
    d, spec = H.load_golden(name)
    ws = H.world_spec(spec)
    for k, env_id in enumerate(int(e) for e in d["env_ids"]):
        co = H.COracle(ws, 1, first_env_id=env_id)
        epoch = int(d["epoch"]) if "epoch" in d else 0
        if name in H.INJECTED_FIXTURES:
            co.grid[0], co.pos[0], co.total[0] = d["grid0"][k], d["pos0"][k], 0.0
        else:
            co.reset(epoch)
        for t in range(d["obs"].shape[0]):
            acts = d["scripted"][t, k] if "scripted" in d else None
            co.step(epoch, t + 1, actions=acts, random_actions=acts is None)
            assert np.array_equal(co.grid[0], d["grid"][t, k]), (name, t)
        n += 1
for case in range(40):
    rng = np.random.default_rng(7000 + case)
    ws, g, pos = H.random_rule_world(rng)
    co = H.COracle(ws, 3)
    co.grid[...] = g; co.pos[...] = pos; co.total[...] = 0
    for t in range(1, 5):
        co.step(0, t, random_actions=True)
    co.observe(); co.metrics()
print("asan/ubsan run clean:", n, "fixture envs + 40 random rule worlds")

"""The configurations tests/test_plan.py enumerates sgw_plan for: every BASELINE config, the shapes measured in profiles/, both sides
of every batch-size / size threshold of the dispatcher, with and without specialised instances.  `python tools/plan_cases.py --update`
rewrites tests/golden/plans.json after a DELIBERATE change of the dispatcher (the diff of that file is the announcement)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "plans.json")


def cases():
    """(name, WorldSpec, num_envs, options)"""
    from sorrel_amd.spec import treasurehunt_spec
    from generic_tables_probe_worlds import move_world
    from tests import helpers as H

    def golden(name, **over):
        d, spec = H.load_golden(name)
        ws = H.world_spec(spec)
        for k, v in over.items():
            setattr(ws, k, v)
        if "num_agents" in over:
            ws.agent_type = [ws.agent_type[0]] * over["num_agents"]
        return ws

    th = treasurehunt_spec
    out = []
    # BASELINE.json configs 1-5 (4 = config 3 per GPU) and config 3's cache-defeating batch
    for name, spec, E in (("c1", th(10, 10, 2, 2), 1), ("c2", th(16, 16, 4, 2), 4096), ("c3", th(32, 32, 8, 3), 65536),
                          ("c3_524288", th(32, 32, 8, 3), 524288), ("c5_per_gpu", th(128, 128, 64, 5, dense_prob=0.25), 2048),
                          ("c5_8192", th(128, 128, 64, 5, dense_prob=0.25), 8192)):
        out.append((name, spec, E, {}))
    # profiles/r03_generic_tables.txt, r03_runtime_shapes.txt: worlds that are not the shipped examples
    for name, spec in (("own_32x32x2_C8", move_world(32, 32, 2, 8, 8, 3)), ("own_32x32x2_C5", move_world(32, 32, 2, 5, 8, 3)),
                       ("own_32x32x1_C4", move_world(32, 32, 1, 4, 8, 3)), ("own_32x32x3_C10", move_world(32, 32, 3, 10, 8, 3)),
                       ("own_24x24x2_C8_r4", move_world(24, 24, 2, 8, 6, 4)), ("own_40x40x2_C12_r2", move_world(40, 40, 2, 12, 8, 2)),
                       ("own_10x10x2_C5", move_world(10, 10, 2, 5, 2, 2)), ("own_16x16x1_C3_r3", move_world(16, 16, 1, 3, 4, 3)),
                       ("own_24x24x3_C7", move_world(24, 24, 3, 7, 4, 2)),
                       ("th_32x33", th(32, 33, 8, 3)), ("th_24x24", th(24, 24, 8, 3)), ("th_40x40", th(40, 40, 8, 3)), ("th_20x20_r4", th(20, 20, 4, 4)),
                       ("th_30x26_r5", th(30, 26, 7, 5)), ("th_21x21_default", th(21, 21, 2, 2)), ("th_10x10", th(10, 10, 2, 2))):
        out.append((name, spec, 65536, {}))
    # the examples as shipped (Tag 11x11, Cleanup 21x31x3) and their big variants (profiles/r03_misc_bench.txt, r03_big_rule_worlds*)
    out += [("tag_11x11", golden("tag_11x11_default"), 65536, {}), ("tag_32x32", golden("tag_9x9", height=32, width=32, num_agents=8, vision_radius=3), 65536, {}),
            ("tag_128x128_A64", golden("tag_9x9", height=128, width=128, num_agents=64, vision_radius=4), 2048, {}),
            ("tag_72x72_A16", golden("tag_9x9", height=72, width=72, num_agents=16, vision_radius=4), 8192, {}),
            ("tag_72x72_A16_small_batch", golden("tag_9x9", height=72, width=72, num_agents=16, vision_radius=4), 2048, {}),
            ("cleanup_21x31", golden("cleanup_21x31_default"), 65536, {}), ("cleanup_15x16", golden("cleanup_15x16"), 4096, {}),
            ("rgb_treasurehunt", golden("rgb_treasurehunt"), 65536, {})]
    # both sides of the thresholds
    for E in (4095, 4096, 12287, 12288):                                  # packing needs a batch that still fills the chip
        out.append((f"pack_16x16_A4_E{E}", th(16, 16, 4, 2), E, {}))
    for E in (4095, 4096, 16383, 16384):                                  # 4-8 KiB / 8-11 KiB worlds on the wave-per-env kernel
        out.append((f"mid_48x48_E{E}", th(48, 48, 8, 5), E, {}))
        out.append((f"mid_72x72_E{E}", th(72, 72, 8, 5), E, {}))
    for E in (1536, 1537, 1792, 1793, 2304, 2305):                        # step_big: walk window (1.5x .. 2.25x resident), staging above 1.75x
        out.append((f"c5_E{E}", th(128, 128, 64, 5, dense_prob=0.25), E, {}))
    # more than 64 agents (round 6): the ticket-ordered generic kernel whatever the size; 64 agents keep their kernels
    out += [("agents_65_of_128x128", th(128, 128, 65, 5), 2048, {}), ("agents_128_of_128x128", th(128, 128, 128, 5), 2048, {}),
            ("agents_80_of_40x40", th(40, 40, 80, 2), 4096, {}), ("agents_64_of_40x40", th(40, 40, 64, 2), 4096, {})]
    out += [("big_256_threads", th(100, 100, 8, 5), 8192, {}), ("big_512_threads", th(128, 128, 32, 5), 8192, {}),
            ("avv_100", th(24, 24, 4, 2), 65536, {}), ("avv_101", th(24, 24, 5, 2), 65536, {}), ("avv_200", th(30, 30, 8, 2), 65536, {}),
            ("cells_1024_pack", th(22, 23, 2, 2), 65536, {}), ("cells_1040_nopack", th(22, 24, 3, 3), 65536, {})]
    # the same worlds on the prebuilt instances (hipRTC absent), and the forced families the GPU tests use
    base = list(out)
    out += [(n + "__prebuilt", s, E, dict(o, jit=0)) for n, s, E, o in base]
    out += [("c3__force_generic", th(32, 32, 8, 3), 65536, {"force_generic": 1}), ("c3__group32", th(32, 32, 8, 3), 65536, {"group": 32}),
            ("c3__chunked", th(32, 32, 8, 3), 65536, {"burst": 2}), ("c5__no_walk", th(128, 128, 64, 5), 2048, {"big_walk": 0}),
            ("cleanup__generic", golden("cleanup_21x31_default"), 65536, {"fast_rules": 0})]
    return out


def plans():
    from sorrel_amd import _native as N

    out = {}
    for name, spec, E, opts in cases():
        cfg = spec.to_config(E, 0)
        cfg.grid_env_stride = (spec.layers * spec.height * spec.width + 15) // 16 * 16      # what GridEngine allocates (spec.alloc_grid)
        with N.options(**opts):
            out[name] = N.plan(cfg, 256, 160 * 1024)
    return out


if __name__ == "__main__":
    got = plans()
    if "--update" in sys.argv:
        with open(GOLDEN, "w") as fh:
            json.dump(got, fh, indent=1, sort_keys=True)
        print(f"wrote {len(got)} plans to {GOLDEN}")
    else:
        for k, v in got.items():
            print(f"{k:34s} {v['kernel']:64s} lanes={v['lanes_per_env']:3d} lds={v['lds_bytes']:6d} stage={v['obs_stage']}/{v['stage_agents']} walk={v['walk_blocks']}")

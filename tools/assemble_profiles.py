#!/usr/bin/env python3
"""gpurun_out/refresh/*.txt (tools/refresh_profiles.sh) -> profiles/r03_*.txt with a header saying what each file is."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, P = os.path.join(ROOT, "gpurun_out", "refresh"), os.path.join(ROOT, "profiles")


def body(f):
    return "\n".join(l for l in open(os.path.join(R, f)).read().splitlines() if "amdgpu.ids" not in l) + "\n"


FILES = {
    "r03_misc_bench.txt": ("misc_bench.txt",
        "# tools/bench_misc.py, round 3, final kernels (one MI355X box; boxes differ by ~5-10 %).  Round 2: profiles/r02_misc_bench_under_rocprof.txt, r02_big_rule_worlds.txt\n"
        "# changes behind the numbers: Tag -- only the 'it' agent looks at its neighbours, compile-time 32x32 wave-per-env instance, TagAgent.act on step_big;\n"
        "# Cleanup -- dword-wise ordered sweep shared by every kernel, quiet-dword skip, 3-bit packed counters, compile-time 21x31x3 instance, and (the big one) staged bursts on 128-byte lines;\n"
        "# small worlds -- the generic kernel's single-turn instances compiled without sgw_rollout's turn loop, a Philox key schedule per block instead of twenty keys held in SGPRs;\n"
        "# then the same Cleanup world at 65 536 envs, then the rule worlds above 4 KiB (MISC_ONLY=big)\n"),
    "r03_phased_path.txt": ("phased_path.txt",
        "# tools/phased_bench.py, round 3 (config 3 at 65 536 envs, then config 5's per-GPU share): a policy-driven turn in four protocols, engine only (no policy forward pass)\n"),
    "r03_cleanup_rollout_and_policy.txt": ("cleanup_rollout_and_policy.txt",
        "# tools/cleanup_rollout_bench.py (16 384 and 65 536 envs), round 3: the Cleanup example's shape turn by turn / sgw_rollout / policy-driven in the round-2 protocol (a window rendered per launch)\n"
        "# and the round-3 one (SGW_STEP_NO_MOVE + sgw_act); then tools/act_probe.py: GPU time per launch of the ten acts of a turn\n"),
    "r03_api_latency.txt": ("api_latency.txt",
        "# tools/latency_bench.py, round 3: policy turns through Environment.take_turn() with the patched-window protocol (sweep + every window once, sgw_act per agent; actions / rewards\n"
        "# written straight into replay rows); round 2 (1 + A launches, a window rendered per launch): profiles/r02_api_latency.txt\n"),
    "r03_runtime_shapes.txt": ("runtime_shapes.txt",
        "# tools/rt_shape_probe.py, round 3, final kernels: Treasurehunt-like worlds WITHOUT a compile-time map next to config 3 (first line), 65 536 envs, us per turn (sweep + moves + every window);\n"
        "# r = 2..5 run on step_fast<true, 2, 6, r, 0, 0, ..., STAGE> (compile-time window, run-time map); the last three lines: the packed small-world kernel.\n"
        "# The wave-per-env kernels are bound by the vector ALU as much as by HBM (profiles/r03_runtime_shapes_pmc.txt: 1 437 against 1 035 vector instructions per wave, 162 against 118 us, before the\n"
        "# compile-time-window instances)\n"),
    "r03_group_sweep.txt": ("group_sweep.txt",
        "# tools/group_sweep.py, round 3 final kernels: wave-per-env (SGW_GROUP=64) / 16 / 32 lanes per env / the generic kernel at 64 -- the data behind the packing rule in sgw_create (round 2: profiles/r02_group_sweep.txt)\n"),
    "r03_big_world_staging.txt": ("big_world_staging.txt",
        "# tools/big_stage_probe.py (then PROBE_WALK=1), round 3: step_big with its windows staged in LDS and written as line-aligned 16-byte streaming stores (default above ~1.75 rounds of workgroups)\n"
        "# against direct dword stores (SGW_BIG_STAGE=0), padded / unpadded LDS rows, and the walking variant's window; us per turn, a digest of obs + grid + totals (equal across variants),\n"
        "# the kernel launched, its LDS request and staging bytes per wave\n"),
    "r03_store_alignment_micro.txt": ("store_alignment_micro.txt",
        "# tools/micro/region_writer.hip, round 3: write-only streams shaped like Cleanup's observations (43 520 B per wave, 65 536 waves = 2.85 GB).  What costs: streaming (nt) stores that cover PART of\n"
        "# a 128-byte line (shifted by 16 / 32 / 64 bytes: ~4.4 / 4.4 / 4.9 TB/s against ~5.5), not who writes a region or in how many bursts; lane 0 on a line boundary (misalign 5 / 6) recovers it.\n"
        "# Behind: emit_chunk (step_fast.h), step_big's staged windows, rows_emit flat mode\n"),
    "r03_hbm_floor.txt": ("hbm_floor.txt",
        "# tools/hbm_floor.py, round 3: torch fill_ / copy_ rates on the same box -- the write-only ceiling falls with the size of the buffer (6.9 TB/s up to ~1.2 GB, 5.8 at Cleanup's 2.85 GB,\n"
        "# 5.6 at 5.7 GB = config 3 at 524 288 envs)\n"),
    "r03_cleanup_observe_probe.txt": ("cleanup_observe_probe.txt",
        "# tools/cleanup_observe_probe.py (SGW_STAGE_AGENTS = 1 / 3 / 5 / 10), round 3: Cleanup 21x31x3 at 65 536 envs, sgw_observe of every agent against the whole turn -- the window pipeline alone\n"
        "# takes as long as the turn (before the line-aligned bursts: observe 683 us / turn 680 us at 3 agents per burst; with the gather skipped 663, with the stores skipped 121: the emit was the bottleneck)\n"),
    "r03_generic_tables.txt": ("generic_tables.txt",
        "# tools/generic_tables_probe.py (then PROBE_SMALL=1), round 3: worlds whose (layers, channels) have no compile-time tables -- 3-bit packed counters for any one-hot table of <= 10 channels,\n"
        "# channel planes staged in groups of four; before: 32x32x2 C8 246.6 us, C5 235.0, 32x32x1 C4 182.7, 32x32x3 C10 309.0, 24x24x2 C8 r4 236.3, 40x40x2 C12 (packed then) 350.6\n"),
    "r03_mid_worlds.txt": ("mid_worlds.txt",
        "# tools/mid_world_probe.py, round 3: plain (th) and Tag worlds between 4 and 8 KiB per env, 2 048 ... 65 536 envs: step_big (SGW_FAST_8K=0) against the wave-per-env kernel (SGW_FAST_8K=1);\n"
        "# sgw_create takes the wave-per-env kernel from 4 096 envs on\n"),
    "r03_tag_group_probe.txt": ("tag_group_probe.txt",
        "# tools/tag_group_probe.py, round 3: Tag worlds up to 4 KiB, the dispatcher's choice / a wave per env (3-bit-counter Tag instance) / two envs per wave -- the data behind Tag's packing rule in sgw_create\n"),
}
for name, (src, hdr) in FILES.items():
    open(os.path.join(P, name), "w").write(hdr + body(src))
    print(name)

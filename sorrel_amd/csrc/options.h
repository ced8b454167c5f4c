// options.h -- host side of sgw.hip (included inside its anonymous namespace).
// Every knob of the dispatcher in ONE table, set through sgw_set_option (include/sgw.h).  The shipped library reads no
// environment variable for any of them (rounds 1-3 grew ~30 getenv hooks; tests and tools now say what they want).
// A NULL engine addresses the process-wide defaults that sgw_plan / sgw_create copy; an engine's own copy is frozen at
// sgw_create except for the keys marked `live`.
#pragma once

struct Options {
    // ---- specialisation
    int jit = 1;              // 1: compile the step kernels for this engine's own constants in-process (hipRTC), cached on disk;
                              // 0: the prebuilt instances only (also what is used when hipRTC is absent or a compile fails)
    int jit_cache = 1;        // keep compiled code objects on disk (jit_cache_dir)
    int jit_verbose = 0;      // one line per compile / cache hit on stderr
    int jit_own_rtc = 1;      // load /opt/rocm/lib/libhiprtc.so in a namespace of its own (the ROCm installation's compiler) rather than
                              // the hipRTC the process has already mapped (PyTorch's wheel bundles an older one); read once per process
    std::string jit_cache_dir;   // "" = <directory of libsgw.so>/jit_cache
    std::string jit_refuse;      // test hook: an instance whose template-id contains this text is refused as if it did not compile ("" = none)
    int burst = 0;            // wave-per-env kernels with a compile-time shape: 0 auto, 1 whole-env burst whenever legal, 2 chunked (STAGE) always
    // ---- which prebuilt instance
    int pack3 = 1;            // 3-bit packed counters for one-hot tables of <= 10 channels
    // ---- which kernel family
    int force_generic = 0;    // the LDS-resident generic kernel for everything
    int fast_rules = 1;       // layered rule sets on the wave-per-env RULES kernel
    int rules_11k = 0;        // ... up to 11 KiB whatever the batch (default: from 16 384 envs on)
    int fast_8k = -1;         // plain / Tag worlds of 4-11 KiB per env on the wave-per-env kernel: -1 by batch size, 0 never, 1 always
    int resolve_diag = 0;     // sgw_turn_resolve timing aid (resolve.h: ResolveArgs::diag)
    int force_big = 0;        // 1: the workgroup-per-env kernel (step_big) whatever the world's size (A/B: small batches of small worlds)
    int group = 0;            // lanes per env of the packed kernel: 0 auto, 16 / 32 force a packing, 64 forbids it
    int phase_kernel = -1;    // the byte-gather phase kernel: -1 worlds above 4 KiB, 0 never, 1 every plain-move world
    int phase_rows = 1;       // the row-load phase kernels (phase_rows / observe_rows)
    // ---- staging and occupancy
    int stage = 1;            // LDS staging of one-hot observations on the wave-per-env kernels
    int stage_agents = -1;    // agents per staged burst (-1 auto)
    int fast_wg_per_cu = 0;   // workgroup-per-CU cap of the wave-per-env kernels' large float32 writes (0 auto)
    int big_threads = 0;      // step_big workgroup: 0 auto, 256, 512
    int big_stage = -1;       // step_big window staging: -1 by batch size, 0 never, 1 always
    int big_wg_per_cu = 0;    // step_big workgroups per CU: 0 = what the code object admits, 1..3 = capped through the LDS request (A/B)
    int big_walk = 1;         // step_big<..., WALK>
    int big_walk_blocks = 0;  // ... this many workgroups whatever the batch (0 auto)
    int big_walk_share = 0;   // ... envs per workgroup assigned statically before the shared counter takes over (0 auto)
    // ---- live (also settable on an engine after sgw_create)
    int rows_mode = 0;        // sgw_observe_rows emit: 0 auto, 1 single floats, 2 float2 runs where legal, 3 aligned float4 runs
    int act_lanes = 0;        // sgw_act with 9..16 agents: lanes per env -- 0 auto, 8 (two agents per lane), 16 (one)
};

struct OptKey {
    const char* name;
    int Options::*field;
    int lo, hi;
    bool live;
};

const OptKey kOptKeys[] = {
    {"jit", &Options::jit, 0, 1, false},
    {"jit_cache", &Options::jit_cache, 0, 1, false},
    {"jit_verbose", &Options::jit_verbose, 0, 1, true},
    {"jit_own_rtc", &Options::jit_own_rtc, 0, 1, false},
    {"burst", &Options::burst, 0, 2, false},
    {"pack3", &Options::pack3, 0, 1, false},
    {"force_generic", &Options::force_generic, 0, 1, false},
    {"fast_rules", &Options::fast_rules, 0, 1, false},
    {"rules_11k", &Options::rules_11k, 0, 1, false},
    {"fast_8k", &Options::fast_8k, -1, 1, false},
    {"force_big", &Options::force_big, 0, 1, false},
    {"resolve_diag", &Options::resolve_diag, 0, 7, true},
    {"group", &Options::group, 0, 64, false},
    {"phase_kernel", &Options::phase_kernel, -1, 1, false},
    {"phase_rows", &Options::phase_rows, 0, 1, false},
    {"stage", &Options::stage, 0, 1, false},
    {"stage_agents", &Options::stage_agents, -1, SGW_MAX_AGENTS, false},
    {"fast_wg_per_cu", &Options::fast_wg_per_cu, 0, 8, false},
    {"big_threads", &Options::big_threads, 0, 512, false},
    {"big_stage", &Options::big_stage, -1, 1, false},
    {"big_wg_per_cu", &Options::big_wg_per_cu, 0, 8, false},
    {"big_walk", &Options::big_walk, 0, 1, false},
    {"big_walk_blocks", &Options::big_walk_blocks, 0, 1 << 20, false},
    {"big_walk_share", &Options::big_walk_share, 0, 1 << 20, false},
    {"rows_mode", &Options::rows_mode, 0, 3, true},
    {"act_lanes", &Options::act_lanes, 0, 16, true},
};

std::mutex g_opt_mu;
Options g_opts;   // the process-wide defaults (guarded by g_opt_mu)

// 0 ok; 1 unknown key; 2 bad value; 3 not settable on a live engine
int option_set(Options& o, const char* key, const char* value, bool live_only) {
    if (!key) {   // all keys back to their defaults
        if (live_only) return 3;
        o = Options();
        return 0;
    }
    if (!strcmp(key, "jit_cache_dir")) {
        if (live_only) return 3;
        o.jit_cache_dir = value ? value : "";
        return 0;
    }
    if (!strcmp(key, "jit_refuse")) {
        if (live_only) return 3;
        o.jit_refuse = value ? value : "";
        return 0;
    }
    for (const OptKey& k : kOptKeys) {
        if (strcmp(k.name, key)) continue;
        if (live_only && !k.live) return 3;
        if (!value || !*value) {   // back to the default
            o.*(k.field) = Options().*(k.field);
            return 0;
        }
        char* end = nullptr;
        const long v = strtol(value, &end, 10);
        if (end == value || *end || v < k.lo || v > k.hi) return 2;
        if (k.field == &Options::group && !(v == 0 || v == 16 || v == 32 || v == 64)) return 2;
        if (k.field == &Options::big_threads && !(v == 0 || v == 256 || v == 512)) return 2;
        o.*(k.field) = (int)v;
        return 0;
    }
    return 1;
}

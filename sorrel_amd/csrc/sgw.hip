// sgw.hip -- MI355X (gfx950 / CDNA4) batched gridworld step + observation engine.
//
// Hand-written HIP behind the C ABI of include/sgw.h.  One thread GROUP owns one environment for a whole take_turn
// (or, through sgw_rollout, for T turns): the env's grid (uint8 type ids, [L][H][W]) is staged once into LDS with 16-byte
// loads, the entity sweep and all sequential agent phases run against LDS, observation windows are gathered from LDS
// (lane = window cell) and the grid is written back once.  Integer / indexing work only: no MFMA; the bound is HBM
// bandwidth for the big shapes (observation stores dominate) and instruction issue for the small ones.
//
// Kernels, in file order:
//   step_kernel<G, ONEHOT, L, C, RULE, r, H, W>
//                                        every shape and rule; G = lanes per env: 256 (a workgroup per env, worlds above
//                                        4 KiB: the four waves take the agents in turn behind an LDS ticket, window bytes
//                                        captured before the act and stored after it), 64 (a wave per env), or 32 / 16 --
//                                        two / four SMALL envs share a wave and its instruction stream (what 10x10 ...
//                                        24x24 worlds of large batches run on; compile-time window for the examples as
//                                        shipped); all per-env state in the group's LDS slice; turn loop for sgw_rollout
//                                        built in
//   step_fast<ONEHOT, L, C, r, H, W, TAG, RULES, STAGE, MULTI>
//                                        a wave per env, worlds <= 4 KiB: register sweep, per-lane move inputs + scalar
//                                        move resolution, compile-time window geometry for the BASELINE shapes; one-hot
//                                        observations staged as bytes in LDS and emitted as one burst of streaming
//                                        16-byte stores (fixed shapes: whole env; STAGE: chunks of agents, any alignment);
//                                        MULTI = the turn loop of sgw_rollout
//   step_big<ONEHOT, L, C, r, MULTI, WALK>
//                                        a 512-thread workgroup per env, worlds above 4 KiB (config 5): padded LDS row
//                                        pitch, moves resolved in registers by wave 0 (only interfering agents are walked),
//                                        observations rendered by all waves from the post-move grid with later moves
//                                        undone in registers; WALK = resident workgroups walking the batch, the next
//                                        env's loads issued ahead of this env's observation stores
//   phase_kernel<ONEHOT>                 one policy-driven phase of a world above 4 KiB without staging the env
//   reset_kernel, random_actions_kernel, init_agent_state_kernel, reduce_stage1/2
// then the host side: validation, table building, kernel selection (sgw_create), the launchers.
//
// Semantics follow the reference Python step loop bit for bit; see include/sgw.h for the reference file:line each entry
// point replaces and oracle/gridstep_oracle.py for the line-by-line CPU restatement the kernels are tested against
// (the product never calls it).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/sgw.h"

namespace {

#include "common.h"
#include "step_generic.h"
#include "step_fast.h"
#include "step_big.h"
#include "phase.h"
#include "small_kernels.h"
#include "resolve.h"

// ---------------------------------------------------------------- host side
#include "options.h"
#include "jit.h"

thread_local char g_err[768] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(SGW_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int kEventPool = 4096;

// One launchable kernel of an engine: a prebuilt instance of the library, and / or the instance specialised for this engine
// that hipRTC compiles (at sgw_create for the whole-turn kernel, at first use for the others).
struct Kernel {
    const void* host = nullptr;     // prebuilt instance (nullptr: none that fits this engine's plan)
    const char* host_name = "-";
    std::string want;               // template-id of the specialised instance ("" : none wanted)
    hipFunction_t jit = nullptr;
    bool tried = false;             // the specialised instance has been asked for (and, if jit is still null, was refused)
    bool usable() const { return host != nullptr || !want.empty(); }
    const char* name() const { return (jit || (!host && !want.empty())) ? want.c_str() : host_name; }
};

}  // namespace

struct sgw_engine {
    sgw_config cfg;
    Options opt;          // the process-wide options as they were at sgw_create (live keys: sgw_set_option on the engine)
    Params base;          // everything except per-call fields
    DevTables* d_tab = nullptr;
    uint8_t* d_tmpl = nullptr;   // fill + border image of one env (reset)
    int* d_status = nullptr;
    double* d_part = nullptr;
    uint8_t* d_dcount = nullptr;     // sgw_turn_resolve, large batches: dirty rows per env of the pass, and where each env's part of the list begins
    uint32_t* d_doffsets = nullptr;
    TurnState* d_turn = nullptr;   // device-side turn state (sgw_turn_*): a whole policy turn as one capturable submission
    bool turn_rows = false;        // sgw_turn_bind gave replay rows
    int64_t turn_cap[SGW_MAX_AGENTS] = {};        // ... host mirror: rows of agent a's ring (0: none, or no states)
    int64_t turn_row_bytes[SGW_MAX_AGENTS] = {};  // ... bytes of one of its rows (E * row_elems * element size)
    const void* turn_states[SGW_MAX_AGENTS] = {};
    bool turn_rows_even = false;   // ... all of them 8-byte aligned with an even row stride (float2 copies)
    bool turn_rows_flat = false;   // ... all of them 16-byte aligned rows of exactly one window per env, E * N * 4 a multiple of 16 (flat second copies)
    int obs_format = SGW_OBS_F32;
    uint8_t* agent_state = nullptr;    // caller-owned, bound with sgw_bind_agent_state
    uint8_t* state_at_pov = nullptr;
    uint8_t* agent_dir = nullptr;      // caller-owned, bound with sgw_bind_agent_dir
    int tail_kind = SGW_TAIL_NONE, tail_len = 0;   // sgw_bind_row_tail
    const float* tail_table = nullptr;
    int wpe = 1;          // waves per env
    int group = 64;       // generic step kernel: threads per env (16 / 32: several envs share a wave)
    bool onehot = true;
    bool rgb16 = false;   // integer appearance tables behind clip / 255: step_fast's I16 instances
    int plain_tab_bytes = 0;   // ... whose direct-store variant is the float64 kernel with its own, larger table area
    bool fast = false;    // step_fast specialisation applies
    bool big = false;     // step_big (workgroup per env, pipelined agents) applies
    bool jit = false;     // the plan counts on instances specialised for this engine (hipRTC)
    bool whole_env_burst = false;   // step_fast with a compile-time shape: the whole env's windows leave in one burst
    Kernel k_step;        // a whole turn (sgw_step of all agents, sgw_observe)
    Kernel k_plain;       // STAGE kernels: the direct-store variant for calls that cannot be staged
    Kernel k_multi;       // sgw_rollout's turns in one launch
    Kernel k_walk;        // step_big<..., WALK>: resident workgroups walking the batch
    Kernel k_rows;        // phase_rows<L, NW, R>: a policy-driven phase with a lane per window row (one-hot, plain moves)
    Kernel k_obs_rows;    // observe_rows<L, NW, R>: a range of agents, per-agent destinations
    Kernel k_sweep_rows;  // step_fast_rows<L, C, R, H, W>: the sweep + every agent's window into per-agent destinations, one launch (step_big<..., ROWS> for worlds above 4 KiB)
    Kernel k_sweep_rows_tail;          // ... step_fast_rows<..., TAIL = true>: the same with the bound row tail behind every window (resolved by sgw_bind_row_tail)
    bool sweep_rows_chunked = false;   // ... it is a step_fast_rowsx instance (a chunk-staging kernel: any env_stride, row tails)
    int walk_blocks = 0;                            // how many workgroups of the walking kernel the chip holds at once
    int64_t walk_min_envs = 0, walk_max_envs = 0;  // batches above min and up to max take it (multiples of what the plain kernel holds at once)
    int64_t big_stage_min_envs = 0;                // step_big stages its windows for batches above this
    int stage_agents = 0;      // agents per staged chunk (STAGE kernels)
    bool phase_ok = false;     // the phase kernel applies (plain moves)
    int rows_wpb = 0;          // windows per 256-thread workgroup of observe_rows
    int rows_epb = 0;          // envs per 256-thread workgroup of it
    size_t rows_lds = 0;
    bool multi_turn = false;   // the step kernel in use runs sgw_rollout's turns in one launch
    void (*reset_fn)(const Params) = nullptr;
    size_t lds_bytes = 0;       // reset / generic step
    size_t step_lds_bytes = 0;  // step kernel actually launched
    bool fast_rules = false;   // the RULES variant of step_fast applies
    int step_env_lds = 0;
    int obs_stage = 0;     // bytes of LDS observation staging per wave (step_fast, one-hot)
    int big_stage = 0;     // ... per wave of step_big (0: direct stores)
    int big_threads = kBigThreads;   // threads per workgroup of the step_big instance in use (the rollout instance: always kBigThreads)
    int fast_tab_bytes = 0;
    int big_tab_bytes = 0;   // step_big: only the counter words of the channels in use
    int grid_blocks = 1;
    int fast_wg_cap = 5;   // step_fast workgroups per CU when writing large float32 observations of a large batch (0: no cap)
    bool fast_wg_cap_forced = false;   // option fast_wg_per_cu given: that value for the staged path too (default there: 6)
    int wg_per_cu = 0;     // sgw_set_wg_per_cu: 0 = the automatic rule above, 1..8 = forced, -1 = never capped
    uint32_t auto_max_turns = 0;       // sgw_set_auto_reset
    double* episode_return = nullptr;  // caller-owned
    int reset_blocks = 1;
    int num_cus = 256;
    size_t lds_cap = 65536;  // sharedMemPerBlock of the device
    int dev = 0;
    std::string arch = "gfx950";
    DevTables h_tab;         // host copy of the tables (plan_engine builds it; sgw_create uploads it)
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev0, ev1;
    int ev_used = 0;
    double ms_acc = 0.0;
    int64_t launches = 0;
    std::vector<float> series;   // per-launch durations since the last sgw_get_step_times_ms
    int64_t series_dropped = 0;  // launches beyond kSeriesCap since the last read: in the sum, not in the series
};

namespace {

uint64_t prob_threshold(double pr) {
    const double t = std::floor(pr * 4294967296.0);
    if (!(t > 0.0)) return 0;
    if (t >= 4294967296.0) return 4294967296ull;
    return (uint64_t)t;
}

int validate(const sgw_config* c) {
    if (!c) return fail(SGW_EINVAL, "config is NULL");
    if (c->height < 3 || c->width < 3 || c->height > SGW_MAX_DIM || c->width > SGW_MAX_DIM)
        return fail(SGW_EINVAL, "height/width must be in [3, %d] (got %dx%d)", SGW_MAX_DIM, c->height, c->width);
    if (c->layers < 1 || c->layers > SGW_MAX_LAYERS)
        return fail(SGW_EINVAL, "layers must be in [1, %d] (got %d)", SGW_MAX_LAYERS, c->layers);
    if (c->num_agents < 1 || c->num_agents > SGW_MAX_AGENTS)
        return fail(SGW_EINVAL, "num_agents must be in [1, %d] (got %d)", SGW_MAX_AGENTS, c->num_agents);
    if (c->num_agents > (c->height - 2) * (c->width - 2))
        return fail(SGW_EINVAL, "more agents (%d) than interior cells", c->num_agents);
    if (c->vision_radius < 0 || c->vision_radius > (std::min(c->height, c->width) - 1) / 2)
        return fail(SGW_EINVAL, "vision_radius %d invalid: visual_field needs r <= (min(H,W)-1)//2 = %d",
                    c->vision_radius, (std::min(c->height, c->width) - 1) / 2);
    if (c->num_types < 1 || c->num_types > SGW_MAX_TYPES)
        return fail(SGW_EINVAL, "num_types must be in [1, %d] (got %d)", SGW_MAX_TYPES, c->num_types);
    if (c->num_channels < 1 || c->num_channels > SGW_MAX_CHANNELS)
        return fail(SGW_EINVAL, "num_channels must be in [1, %d] (got %d)", SGW_MAX_CHANNELS, c->num_channels);
    if (c->num_actions < 1 || c->num_actions > SGW_MAX_ACTIONS)
        return fail(SGW_EINVAL, "num_actions must be in [1, %d] (got %d)", SGW_MAX_ACTIONS, c->num_actions);
    for (int i = 0; i < c->num_actions; ++i)
        if (c->action_dy[i] < -1 || c->action_dy[i] > 1 || c->action_dx[i] < -1 || c->action_dx[i] > 1)
            return fail(SGW_EINVAL, "action %d: (dy, dx) must be in {-1,0,1}", i);
    if (c->agent_layer < 0 || c->agent_layer >= c->layers) return fail(SGW_EINVAL, "agent_layer out of range");
    if (c->default_type < 0 || c->default_type >= c->num_types) return fail(SGW_EINVAL, "default_type out of range");
    if (c->fill_type < 0 || c->fill_type >= c->num_types) return fail(SGW_EINVAL, "fill_type out of range");
    for (int a = 0; a < c->num_agents; ++a) {
        if (c->agent_type[a] >= c->num_types) return fail(SGW_EINVAL, "agent_type[%d] out of range", a);
        if (c->type_rule[c->agent_type[a]] != SGW_RULE_NONE)
            return fail(SGW_EINVAL, "agent types are skipped by the sweep and must have SGW_RULE_NONE");
    }
    for (int t = 0; t < c->num_types; ++t) {
        if (c->type_rule[t] == SGW_RULE_NONE) continue;
        if (c->type_rule[t] == SGW_RULE_BECOME_IF) {
            if (c->rule_layer[t] >= c->layers) return fail(SGW_EINVAL, "type %d: rule_layer out of range", t);
            if (c->rule_become[t] >= c->num_types) return fail(SGW_EINVAL, "type %d: rule_become out of range", t);
            continue;
        }
        if (c->type_rule[t] != SGW_RULE_SPAWN)
            return fail(SGW_EINVAL, "type %d: unsupported transition rule %d", t, (int)c->type_rule[t]);
        if (c->spawn_count[t] < 1 || c->spawn_count[t] > SGW_MAX_CHOICES)
            return fail(SGW_EINVAL, "type %d: spawn_count must be in [1, %d]", t, SGW_MAX_CHOICES);
        for (int k = 0; k < c->spawn_count[t]; ++k)
            if (c->spawn_choice[t][k] >= c->num_types) return fail(SGW_EINVAL, "type %d: spawn choice out of range", t);
        if (!(c->spawn_prob[t] >= 0.0 && c->spawn_prob[t] <= 1.0))
            return fail(SGW_EINVAL, "type %d: spawn_prob must be in [0, 1]", t);
    }
    for (int z = 0; z < c->layers; ++z) {
        if (c->layer_fill_type[z] >= c->num_types) return fail(SGW_EINVAL, "layer_fill_type[%d] out of range", z);
        if (c->layer_border_type[z] != SGW_NO_BORDER && c->layer_border_type[z] >= c->num_types)
            return fail(SGW_EINVAL, "layer_border_type[%d] out of range", z);
    }
    if (c->dense_count > SGW_MAX_CHOICES) return fail(SGW_EINVAL, "dense_count too large");
    for (int k = 0; k < c->dense_count; ++k)
        if (c->dense_choice[k] >= c->num_types) return fail(SGW_EINVAL, "dense choice out of range");
    if (!(c->dense_prob >= 0.0 && c->dense_prob <= 1.0)) return fail(SGW_EINVAL, "dense_prob must be in [0, 1]");
    if (c->agent_rule != SGW_AGENT_RULE_MOVE && c->agent_rule != SGW_AGENT_RULE_TAG && c->agent_rule != SGW_AGENT_RULE_CLEANUP)
        return fail(SGW_EINVAL, "unknown agent_rule %d", (int)c->agent_rule);
    if (c->agent_rule == SGW_AGENT_RULE_CLEANUP) {
        if (c->beam_radius < 0 || 3 * c->beam_radius > 64) return fail(SGW_EINVAL, "beam_radius must be in [0, 21]");
        if (c->clean_beam_type >= c->num_types || c->zap_beam_type >= c->num_types)
            return fail(SGW_EINVAL, "beam types out of range");
        for (int i = 0; i < c->num_actions; ++i)
            if (c->action_kind[i] > SGW_ACTION_ZAP) return fail(SGW_EINVAL, "action %d: unknown action kind", i);
    }
    if (c->agent_rule == SGW_AGENT_RULE_TAG) {
        if (c->tag_it_type >= c->num_types || c->tag_notit_type >= c->num_types || c->tag_it_type == c->tag_notit_type)
            return fail(SGW_EINVAL, "tag_it_type / tag_notit_type must be two distinct registered types");
        if (c->type_passable[c->tag_it_type] || c->type_passable[c->tag_notit_type])
            return fail(SGW_EINVAL, "tag agent types must be impassable");
    }
    if (c->obs_post != SGW_OBS_POST_NONE && c->obs_post != SGW_OBS_POST_CLIP255_DIV255)
        return fail(SGW_EINVAL, "unknown obs_post %d", c->obs_post);
    if (c->grid_env_stride != 0 && c->grid_env_stride < (int64_t)c->layers * c->height * c->width)
        return fail(SGW_EINVAL, "grid_env_stride is smaller than one env");
    if (c->num_envs < 1) return fail(SGW_EINVAL, "num_envs must be >= 1");
    if (c->first_env_id + (uint64_t)c->num_envs > 4294967296ull)
        return fail(SGW_EINVAL, "global env ids must fit 32 bits");
    return SGW_OK;
}

using StepFn = void (*)(const Params);
using RowsFn = void (*)(const Params, const RowPtrs);

// ---- prebuilt instances (the path when hipRTC is absent; also what a specialised instance falls back to) -----------
#define PICK(...)                                             \
    do {                                                      \
        *name = #__VA_ARGS__;                                 \
        return reinterpret_cast<const void*>(static_cast<StepFn>(__VA_ARGS__)); \
    } while (0)

#define PICK2(...)                                            \
    do {                                                      \
        *name = #__VA_ARGS__;                                 \
        return reinterpret_cast<const void*>(static_cast<RowsFn>(__VA_ARGS__)); \
    } while (0)

// step_kernel<G, ONEHOT, L, C, RULE, r, H, W, MULTI>: the single-turn instance, or (multi) the one with sgw_rollout's turn loop.
// A compile-time radius for the examples as shipped (Tag 11x11 / 9x9 windows 116-119 -> 107-108 us, Treasurehunt 5x5 51.0 -> 46.2 us at
// 65 536 envs); everything else about a user's own world comes from the specialised instance (jit.h).
#define PICK_SK(G_, OH, L_, C_, RULE_, R_, H_, W_, NAME)                          \
    do {                                                                          \
        *name = multi ? NAME " (turn loop)" : NAME;                               \
        return multi ? reinterpret_cast<const void*>(static_cast<RowsFn>(step_kernel<G_, OH, L_, C_, RULE_, R_, H_, W_, true>)) \
                     : reinterpret_cast<const void*>(static_cast<RowsFn>(step_kernel<G_, OH, L_, C_, RULE_, R_, H_, W_, false>)); \
    } while (0)
// (round 5) The library holds the turn-loop (sgw_rollout) instance of the packed / wave-per-env kernels for plain movers only, plus the Tag
// example as shipped: every run-time-shape turn-loop instance of the Tag / Cleanup rules and every one of the workgroup-per-env form
// (G = 256) spilled registers to scratch (12-100 bytes per lane; tools/regs.py), and with hipRTC the normal path they were fallbacks of
// fallbacks.  Without one, sgw_rollout is a loop of single-turn launches (the specialised instance, where hipRTC is there, has none of that).
#define PICK_SK1(G_, OH, L_, C_, RULE_, R_, H_, W_, NAME)                         \
    do {                                                                          \
        *name = multi ? "-" : NAME;                                               \
        return multi ? nullptr : reinterpret_cast<const void*>(static_cast<RowsFn>(step_kernel<G_, OH, L_, C_, RULE_, R_, H_, W_, false>)); \
    } while (0)
template <int G>
const void* pick_step_g(bool onehot, int L, int C, int rule, int r, int H, int W, bool multi, const char** name) {
    constexpr int kMove = SGW_AGENT_RULE_MOVE, kTag = SGW_AGENT_RULE_TAG, kCleanup = SGW_AGENT_RULE_CLEANUP;
    if (rule == SGW_AGENT_RULE_CLEANUP) {
        if (onehot) PICK_SK1(G, true, 0, 0, kCleanup, 0, 0, 0, "step_kernel<G, true, 0, 0, SGW_AGENT_RULE_CLEANUP>");
        PICK_SK1(G, false, 0, 0, kCleanup, 0, 0, 0, "step_kernel<G, false, 0, 0, SGW_AGENT_RULE_CLEANUP>");
    }
    if (rule == SGW_AGENT_RULE_TAG) {
        if constexpr (G == 32) {
            if (onehot && L == 1 && C == 4 && r == 4 && H == 11 && W == 11) PICK_SK(32, true, 1, 4, kTag, 4, 11, 11, "step_kernel<32, true, 1, 4, SGW_AGENT_RULE_TAG, 4, 11, 11>");   // the Tag example as shipped
            if (onehot && L == 1 && C == 4 && r == 4) PICK_SK1(32, true, 1, 4, kTag, 4, 0, 0, "step_kernel<32, true, 1, 4, SGW_AGENT_RULE_TAG, 4>");
            if (onehot && L == 1 && C == 4 && r == 3) PICK_SK1(32, true, 1, 4, kTag, 3, 0, 0, "step_kernel<32, true, 1, 4, SGW_AGENT_RULE_TAG, 3>");
        }
        if (onehot && L == 1 && C == 4) PICK_SK1(G, true, 1, 4, kTag, 0, 0, 0, "step_kernel<G, true, 1, 4, SGW_AGENT_RULE_TAG>");   // the Tag example's tables
        if (onehot) PICK_SK1(G, true, 0, 0, kTag, 0, 0, 0, "step_kernel<G, true, 0, 0, SGW_AGENT_RULE_TAG>");
        PICK_SK1(G, false, 0, 0, kTag, 0, 0, 0, "step_kernel<G, false, 0, 0, SGW_AGENT_RULE_TAG>");
    }
    if constexpr (G == 256) {      // a workgroup per env: single-turn instances only
        if (onehot && L == 2 && C == 6) PICK_SK1(G, true, 2, 6, kMove, 0, 0, 0, "step_kernel<G, true, 2, 6>");
        if (onehot) PICK_SK1(G, true, 0, 0, kMove, 0, 0, 0, "step_kernel<G, true>");
        PICK_SK1(G, false, 0, 0, kMove, 0, 0, 0, "step_kernel<G, false>");
    } else {
        if constexpr (G == 16)
            if (onehot && L == 2 && C == 6 && r == 2) PICK_SK(16, true, 2, 6, kMove, 2, 0, 0, "step_kernel<16, true, 2, 6, SGW_AGENT_RULE_MOVE, 2>");   // the Treasurehunt example's 5x5 windows
        if (onehot && L == 2 && C == 6) PICK_SK(G, true, 2, 6, kMove, 0, 0, 0, "step_kernel<G, true, 2, 6>");                             // Treasurehunt-shaped tables
        if (onehot) PICK_SK(G, true, 0, 0, kMove, 0, 0, 0, "step_kernel<G, true>");
        PICK_SK(G, false, 0, 0, kMove, 0, 0, 0, "step_kernel<G, false>");
    }
}
#undef PICK_SK
#undef PICK_SK1

// more than 64 agents (round 6): the workgroup-per-env instances with 128-entry per-agent arrays; prebuilt with run-time shapes only (single-turn: a rollout
// is a loop of launches without hipRTC)
const void* pick_step_many(bool onehot, int rule, bool multi, const char** name) {
    if (multi) return nullptr;
    if (rule == SGW_AGENT_RULE_CLEANUP) {
        if (onehot) PICK2(step_kernel<256, true, 0, 0, SGW_AGENT_RULE_CLEANUP, 0, 0, 0, false, SGW_MAX_AGENTS>);
        PICK2(step_kernel<256, false, 0, 0, SGW_AGENT_RULE_CLEANUP, 0, 0, 0, false, SGW_MAX_AGENTS>);
    }
    if (rule == SGW_AGENT_RULE_TAG) {
        if (onehot) PICK2(step_kernel<256, true, 0, 0, SGW_AGENT_RULE_TAG, 0, 0, 0, false, SGW_MAX_AGENTS>);
        PICK2(step_kernel<256, false, 0, 0, SGW_AGENT_RULE_TAG, 0, 0, 0, false, SGW_MAX_AGENTS>);
    }
    if (onehot) PICK2(step_kernel<256, true, 0, 0, SGW_AGENT_RULE_MOVE, 0, 0, 0, false, SGW_MAX_AGENTS>);
    PICK2(step_kernel<256, false, 0, 0, SGW_AGENT_RULE_MOVE, 0, 0, 0, false, SGW_MAX_AGENTS>);
}
const void* pick_step(const Options& o, int group, bool onehot, int L, int C, int rule, int r, int H, int W, bool multi, const char** name) {
    if (group == 16) return pick_step_g<16>(onehot, L, C, rule, r, H, W, multi, name);
    if (group == 32) return pick_step_g<32>(onehot, L, C, rule, r, H, W, multi, name);
    if (group == 64) return pick_step_g<64>(onehot, L, C, rule, r, H, W, multi, name);
    return pick_step_g<256>(onehot, L, C, rule, r, H, W, multi, name);
}
// the MULTI (turn-loop) instantiations of step_fast that sgw_rollout launches; nullptr: no such variant, the rollout is
// a loop of single-turn launches
const void* pick_fast_multi(const Options& o, bool onehot, int L, int C, int r, int H, int W, bool tag, bool rules, bool stage, const char** name) {
    const bool p3 = onehot && rules && C <= 10 && L <= 7 && o.pack3;
    if (p3 && stage && L == 3 && C == 9) PICK(step_fast<true, 3, 9, 0, 0, 0, false, true, true, true, true>);   // Cleanup
    if (p3 && stage) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true, true, true, true>);    // layered rule sets
    if (onehot && rules && stage && L == 3 && C == 9) PICK(step_fast<true, 3, 9, 0, 0, 0, false, true, true, true>);
    if (onehot && rules && stage) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true, true, true>);
    if (!onehot || tag || rules || L != 2 || C != 6) return nullptr;
    if (r == 3 && H == 32 && W == 32) PICK(step_fast<true, 2, 6, 3, 32, 32, false, false, false, true>);
    if (r == 2 && H == 16 && W == 16) PICK(step_fast<true, 2, 6, 2, 16, 16, false, false, false, true>);
    if (stage) PICK(step_fast<true, 2, 6, 0, 0, 0, false, false, true, true>);
    return nullptr;
}

StepFn pick_reset(int wpe) { return wpe == 1 ? reset_kernel<1> : reset_kernel<4>; }

const void* pick_big(bool onehot, int L, int C, int r, bool tag, int threads, const char** name) {
    if (tag) {   // TagAgent.act on the workgroup-per-env kernel (moves in registers, the "it" token walked by wave 0)
        if (threads == 256) {
            if (onehot && L == 1 && C == 4 && r == 4) PICK2(step_big<true, 1, 4, 4, false, false, true, 256>);
            if (onehot) PICK2(step_big<true, 0, 0, 0, false, false, true, 256>);
        }
        if (onehot && L == 1 && C == 4 && r == 4) PICK2(step_big<true, 1, 4, 4, false, false, true>);   // the Tag example's tables and 9x9 window
        if (onehot) PICK2(step_big<true, 0, 0, 0, false, false, true>);
        PICK2(step_big<false, 0, 0, 0, false, false, true>);
    }
    if (!onehot) PICK2(step_big<false, 0, 0, 0>);
    if (threads == 256) {   // up to 32 agents: four waves per workgroup
        if (L == 2 && C == 6 && r == 5) PICK2(step_big<true, 2, 6, 5, false, false, false, 256>);
        PICK2(step_big<true, 0, 0, 0, false, false, false, 256>);
    }
    if (L == 2 && C == 6 && r == 5) PICK2(step_big<true, 2, 6, 5>);   // BASELINE config 5
    PICK2(step_big<true, 0, 0, 0>);
}

// which of pick_big's choices run 256 threads (the others: kBigThreads): worlds whose windows are little work for eight waves --
// agents x window cells up to 2 048 (round 3, 8 192 envs, us at 512 -> 256 threads: 90x90x2 A16 r3 123 -> 98, 100x100x2 A8 r5 106 -> 89,
// Tag 128x128 A32 r3 143 -> 119, Tag 160x160 A16 r4 141 -> 135, 128x128x2 A16 r3 194 -> 200; but 128x128x2 A32 r5 211 -> 243, config 5 352 -> 394)
int big_threads_for(const Options& o, bool onehot, int num_agents, int window_cells) {
    if (o.big_threads == 256 || o.big_threads == 512) return onehot ? o.big_threads : kBigThreads;   // A/B and test hook
    return (onehot && num_agents * window_cells <= 2048) ? 256 : kBigThreads;
}

bool fixed_fast_shape(int L, int C, int r, int H, int W, bool tag) {   // = the compile-time-shape instances of pick_fast
    if (tag) return L == 1 && C == 4 && r == 3 && H == 32 && W == 32;
    return L == 2 && C == 6 && ((r == 3 && H == 32 && W == 32) || (r == 2 && H == 16 && W == 16));
}

const void* pick_big_multi(bool onehot, int L, int C, int r, const char** name) {
    if (!onehot) PICK2(step_big<false, 0, 0, 0, true>);
    if (L == 2 && C == 6 && r == 5) PICK2(step_big<true, 2, 6, 5, true>);
    *name = "-";      // (round 5: the run-time-table turn-loop instance used 28 bytes of scratch per lane; without hipRTC such a world's rollout
    return nullptr;   // is a loop of single-turn launches)
}

const void* pick_big_walk(bool onehot, int L, int C, int r, int threads, const char** name) {
    if (!onehot) PICK2(step_big<false, 0, 0, 0, false, true>);
    if (threads == 256) {
        if (L == 2 && C == 6 && r == 5) PICK2(step_big<true, 2, 6, 5, false, true, false, 256>);
        PICK2(step_big<true, 0, 0, 0, false, true, false, 256>);
    }
    if (L == 2 && C == 6 && r == 5) PICK2(step_big<true, 2, 6, 5, false, true>);
    PICK2(step_big<true, 0, 0, 0, false, true>);
}

// phase_rows instances: layers x counter words (channels / 4) x vision radius.  Shapes outside the table are compiled on demand
// (jit.h); without hipRTC they keep the staging kernels (worlds <= 4 KiB) or phase_kernel (above).
const void* pick_rows(int L, int NW, int r, const char** name, const void** obs_fn, const char** obs_name) {
#define ROWS_CASE(l, n, rr)                           \
    if (L == l && NW == n && r == rr) {               \
        *obs_fn = reinterpret_cast<const void*>(static_cast<RowsFn>(observe_rows<l, n, rr>)); \
        *obs_name = "observe_rows<" #l ", " #n ", " #rr ">"; \
        PICK(phase_rows<l, n, rr>);                   \
    }
    ROWS_CASE(2, 2, 3);   // BASELINE configs 3 / 4
    ROWS_CASE(2, 2, 2);   // BASELINE config 2, the Treasurehunt example
    ROWS_CASE(2, 2, 5);   // BASELINE config 5
#undef ROWS_CASE
    return nullptr;
}

const void* pick_fast(const Options& o, bool onehot, bool rgb16, int L, int C, int r, int H, int W, bool tag, bool rules, bool stage, const char** name) {
    if (rgb16 && stage && !rules && C == 3) {   // integer colour tables behind clip / 255 (the reference's RGBObservationSpec): 16-bit counters, result table
        if (tag) PICK(step_fast<true, 0, 3, 0, 0, 0, true, false, true, false, false, true>);
        PICK(step_fast<true, 0, 3, 0, 0, 0, false, false, true, false, false, true>);
    }
    if (rules) {
        // one-hot tables of <= 10 channels: 3-bit packed counters (ONE table word per cell and layer instead of ceil(C / 4))
        const bool p3 = onehot && C <= 10 && L <= 7 && o.pack3;
        if (p3 && L == 3 && C == 9 && stage && r == 5 && H == 21 && W == 31)
            PICK(step_fast<true, 3, 9, 5, 21, 31, false, true, true, false, true>);   // Cleanup as shipped (21x31x3 map, 11x11 windows)
        if (p3 && stage) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true, true, false, true>);
        if (p3) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true, false, false, true>);
        if (onehot && stage) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true, true>);
        if (onehot) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true>);
        PICK(step_fast<false, 0, 0, 0, 0, 0, false, true>);
    }
    if (tag) {
        if (onehot && L == 1 && C == 4 && r == 3 && H == 32 && W == 32) PICK(step_fast<true, 1, 4, 3, 32, 32, true>);   // Tag on the headline's map
        const bool p3t = onehot && stage && C <= 10 && L <= 7 && o.pack3;   // 3-bit packed counters
        if (p3t) PICK(step_fast<true, 0, 0, 0, 0, 0, true, false, true, false, true>);
        if (onehot && stage) PICK(step_fast<true, 0, 0, 0, 0, 0, true, false, true>);
        if (onehot) PICK(step_fast<true, 0, 0, 0, 0, 0, true>);
        PICK(step_fast<false, 0, 0, 0, 0, 0, true>);
    }
    if (!onehot) PICK(step_fast<false, 0, 0, 0, 0, 0>);
    if (L == 2 && C == 6 && r == 3 && H == 32 && W == 32) PICK(step_fast<true, 2, 6, 3, 32, 32>);   // BASELINE configs 3/4 (headline)
    if (L == 2 && C == 6 && r == 2 && H == 16 && W == 16) PICK(step_fast<true, 2, 6, 2, 16, 16>);   // BASELINE config 2
    if (L == 2 && C == 6) {   // treasurehunt-shaped, any size
        if (stage) PICK(step_fast<true, 2, 6, 0, 0, 0, false, false, true>);
        PICK(step_fast<true, 2, 6, 0, 0, 0>);
    }
    // any other one-hot table of <= 10 channels: 3-bit packed counters (ONE table word per cell and layer instead of four, ten
    // guarded channel planes instead of sixteen)
    const bool p3 = onehot && stage && C <= 10 && L <= 7 && o.pack3;
    if (p3) PICK(step_fast<true, 0, 0, 0, 0, 0, false, false, true, false, true>);
    if (stage) PICK(step_fast<true, 0, 0, 0, 0, 0, false, false, true>);
    PICK(step_fast<true, 0, 0, 0, 0, 0>);
}

// ---- template-ids of the specialised instances (spelled like the PICK names: trailing default arguments dropped, so an instance the
// library already holds is recognised and not compiled again)
std::string join_args(const char* tmpl, std::vector<std::string> a, size_t keep, const char* drop) {
    while (a.size() > keep && a.back() == drop) a.pop_back();
    std::string s = std::string(tmpl) + "<";
    for (size_t i = 0; i < a.size(); ++i) s += (i ? ", " : "") + a[i];
    return s + ">";
}
const char* tf(bool b) { return b ? "true" : "false"; }
std::string fast_rows_id(int L, int C, int r, int H, int W, bool tag = false, bool tail = false) {
    return "step_fast_rows<" + std::to_string(L) + ", " + std::to_string(C) + ", " + std::to_string(r) + ", " + std::to_string(H) + ", " + std::to_string(W) +
           (tail ? (tag ? ", true, true>" : ", false, true>") : (tag ? ", true>" : ">"));
}
// the ROWX twin of a chunk-staging instance `name` ("step_fast<true, ..., STAGE = true, ...>") for this engine's constants; "" if `name` is not one
std::string fast_rowsx_id_like(const char* name, int L, int C, int r, int H, int W) {
    std::vector<std::string> a;
    const char* s = name ? strchr(name, '<') : nullptr;
    if (!s) return "";
    std::string cur;
    for (++s; *s && *s != '>'; ++s) {
        if (*s == ',') { a.push_back(cur); cur.clear(); }
        else if (*s != ' ') cur += *s;
    }
    a.push_back(cur);
    while (a.size() < 12) a.push_back("false");
    if (a[0] != "true" || a[8] != "true" || a[9] == "true" || a[11] == "true") return "";     // one-hot, STAGE, single-turn, not the 16-bit colour instance
    return "step_fast_rowsx<" + std::to_string(L) + ", " + std::to_string(C) + ", " + std::to_string(r) + ", " + std::to_string(H) + ", " + std::to_string(W) + ", " +
           a[6] + ", " + a[7] + ", " + a[10] + ">";
}
std::string fast_id(bool onehot, int L, int C, int r, int H, int W, bool tag, bool rules, bool stage, bool multi, bool p3, bool i16) {
    return join_args("step_fast", {tf(onehot), std::to_string(L), std::to_string(C), std::to_string(r), std::to_string(H), std::to_string(W),
                                   tf(tag), tf(rules), tf(stage), tf(multi), tf(p3), tf(i16)}, 6, "false");
}
// the prebuilt choice `name` ("step_fast<...>") with its numeric arguments replaced by this engine's own (and MULTI / STAGE as asked)
std::string fast_id_like(const char* name, int L, int C, int r, int H, int W, int stage /* -1 keep */, int multi /* -1 keep */) {
    std::vector<std::string> a;
    const char* s = strchr(name, '<');
    if (!s) return "";
    std::string cur;
    for (++s; *s && *s != '>'; ++s) {
        if (*s == ',') { a.push_back(cur); cur.clear(); }
        else if (*s != ' ') cur += *s;
    }
    a.push_back(cur);
    while (a.size() < 12) a.push_back("false");
    const bool i16 = a[11] == "true";
    return fast_id(a[0] == "true", L, C, r, i16 ? 0 : H, i16 ? 0 : W, a[6] == "true", a[7] == "true", stage < 0 ? a[8] == "true" : stage != 0,
                   multi < 0 ? a[9] == "true" : multi != 0, a[10] == "true", i16);
}
std::string generic_id(int G, bool onehot, int L, int C, int rule, int r, int H, int W, bool multi, bool many = false, bool rows = false) {
    if (rows)        // (the ROWS instance: every argument spelled)
        return join_args("step_kernel", {std::to_string(G), tf(onehot), std::to_string(L), std::to_string(C), std::to_string(rule), std::to_string(r),
                                         std::to_string(H), std::to_string(W), "false", std::to_string(many ? SGW_MAX_AGENTS : 64), "true"}, 11, "");
    if (many)        // (128-entry per-agent arrays: every argument spelled)
        return join_args("step_kernel", {std::to_string(G), tf(onehot), std::to_string(L), std::to_string(C), std::to_string(rule), std::to_string(r),
                                         std::to_string(H), std::to_string(W), tf(multi), std::to_string(SGW_MAX_AGENTS)}, 10, "");
    return join_args("step_kernel", {std::to_string(G), tf(onehot), std::to_string(L), std::to_string(C), std::to_string(rule), std::to_string(r),
                                     std::to_string(H), std::to_string(W), tf(multi)}, 8, "false");
}
std::string big_id(bool onehot, int L, int C, int r, bool multi, bool walk, bool tag, int threads, bool rows = false) {
    std::vector<std::string> a = {tf(onehot), std::to_string(L), std::to_string(C), std::to_string(r), tf(multi), tf(walk), tf(tag), std::to_string(threads)};
    if (rows) { a.push_back("true"); return join_args("step_big", a, 9, ""); }      // (the ROWS instance: every argument spelled)
    if (threads == kBigThreads) a.pop_back();
    return join_args("step_big", a, 4, threads == kBigThreads ? "false" : "");
}
std::string rows_id(const char* tmpl, int L, int NW, int r) {
    return std::string(tmpl) + "<" + std::to_string(L) + ", " + std::to_string(NW) + ", " + std::to_string(r) + ">";
}

int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr size_t kSeriesCap = (size_t)1 << 20;

// workgroups of `threads` threads and `lds` dynamic bytes a CU holds at once when the kernel is compiled for `waves_per_simd` (its
// __launch_bounds__): what hipOccupancyMaxActiveBlocksPerMultiprocessor answers, as pure arithmetic (sgw_plan makes no HIP call)
int resident_per_cu(int threads, size_t lds, int waves_per_simd) {
    const int by_waves = std::max(1, (waves_per_simd * 4) / std::max(1, threads / kWave));
    const int by_lds = lds ? (int)(kLdsPerCu / (((lds + 1023) & ~(size_t)1023))) : by_waves;
    return std::max(1, std::min(by_waves, by_lds));
}

// Everything sgw_create decides, as a function of (config, options, CUs, LDS per workgroup) alone: kernel family, lanes per env, LDS
// layout, staging, walk window, the instances to launch.  No HIP call (sgw_plan runs it without a device; tests/test_plan.py
// enumerates it).  `jit`: specialised instances may be counted on.
int plan_engine(sgw_engine* e, bool jit) {
    const sgw_config& c = e->cfg;
    const Options& o = e->opt;
    e->jit = jit;
    e->k_step = e->k_plain = e->k_multi = e->k_walk = e->k_rows = e->k_obs_rows = e->k_sweep_rows = e->k_sweep_rows_tail = Kernel();
    e->stage_agents = 0;
    e->walk_blocks = 0;
    e->walk_min_envs = e->walk_max_envs = e->big_stage_min_envs = 0;
    e->big_stage = 0;
    e->whole_env_burst = false;

    // ---- tables
    DevTables& h = e->h_tab;
    memset(&h, 0, sizeof(h));
    bool onehot = true;
    for (int t = 0; t < c.num_types; ++t) {
        int ones = 0, ch = -1;
        bool other = false;
        for (int k = 0; k < c.num_channels; ++k) {
            const double v = c.appearance[t][k];
            h.appearance[t][k] = v;
            if (v == 1.0) { ++ones; ch = k; }
            else if (v != 0.0) other = true;
        }
        if (other || ones > 1) onehot = false;
        if (ones == 1 && !other) h.delta[ch >> 2][t] = 1u << (8 * (ch & 3));
        if (ones == 1 && !other && ch < 10) h.delta3[t] = 1u << (3 * ch);
        h.value[t] = c.type_value[t];
        h.thr_lo[t] = (uint32_t)(prob_threshold(c.spawn_prob[t]) & 0xFFFFFFFFull);
        h.spawn_count[t] = c.spawn_count[t];
        memcpy(h.spawn_choice[t], c.spawn_choice[t], SGW_MAX_CHOICES);
    }
    for (int t = 0; t < c.num_types; ++t) {
        h.rule[t] = c.type_rule[t];
        h.rule_layer[t] = c.rule_layer[t];
        h.rule_become[t] = c.rule_become[t];
        h.rule_mask[t] = c.rule_mask[t];
    }
    memcpy(h.agent_type, c.agent_type, SGW_MAX_AGENTS);
    memcpy(h.dense_choice, c.dense_choice, SGW_MAX_CHOICES);
    memcpy(h.layer_fill, c.layer_fill_type, 8);
    memcpy(h.layer_border, c.layer_border_type, 8);
    if (c.obs_post != SGW_OBS_POST_NONE) onehot = false;   // post-processing lives on the general float64 path
    e->onehot = onehot;
    // ... except the reference's RGBObservationSpec as it builds its own maps (uint8 colours, clip / 255): integer tables of <= 4
    // channels take the byte-staging window pipeline with 16-bit counters and a table of the 256 possible results (step_fast.h, I16)
    bool rgb16 = c.obs_post == SGW_OBS_POST_CLIP255_DIV255 && c.num_channels <= 4 && c.layers <= 7;
    for (int t = 0; t < c.num_types && rgb16; ++t)
        for (int k = 0; k < c.num_channels; ++k) {
            const double v = c.appearance[t][k];
            if (!(v >= 0.0 && v <= 9362.0 && v == std::floor(v))) rgb16 = false;
        }
    if (rgb16) {
        for (int t = 0; t < c.num_types; ++t)
            for (int k = 0; k < c.num_channels; ++k) h.delta16[k >> 1][t] |= (uint32_t)c.appearance[t][k] << (16 * (k & 1));
        for (int k = 0; k < 256; ++k) h.post_lut[k] = (float)(std::fmin(std::fmax((double)k, 0.0), 255.0) / 255.0);   // = obs_finish of an integer sum
    }
    e->rgb16 = rgb16;

    // ---- static launch parameters
    Params& p = e->base;
    memset(&p, 0, sizeof(p));
    p.H = c.height; p.W = c.width; p.L = c.layers; p.A = c.num_agents; p.r = c.vision_radius;
    p.V = 2 * c.vision_radius + 1; p.VV = p.V * p.V; p.C = c.num_channels; p.T = c.num_types;
    p.nact = c.num_actions; p.zA = c.agent_layer;
    p.cells = c.layers * c.height * c.width;
    p.cells_pad = (p.cells + 15) & ~15;
    p.env_stride = c.grid_env_stride > 0 ? c.grid_env_stride : p.cells;
    p.env_lds = p.cells_pad + agent_lds_bytes(c.num_agents > 64 ? SGW_MAX_AGENTS : 64);      // (the generic kernel's per-agent LDS arrays: its AC template argument)
    p.tab_bytes = onehot ? kTabFastBytes : (int)sizeof(DevTables);
    p.default_type = (uint32_t)c.default_type;
    p.fill_type = (uint32_t)c.fill_type;
    for (int t = 0; t < c.num_types; ++t) {
        if (c.type_rule[t] == SGW_RULE_SPAWN) {
            p.spawn_mask |= 1u << t;
            if (prob_threshold(c.spawn_prob[t]) >= 4294967296ull) p.thr_full_mask |= 1u << t;
        }
        if (c.type_passable[t]) p.pass_mask |= 1u << t;
    }
    for (int a = 0; a < c.num_actions; ++a) {
        p.dy_pack |= (uint32_t)(c.action_dy[a] + 1) << (2 * a);
        p.dx_pack |= (uint32_t)(c.action_dx[a] + 1) << (2 * a);
    }
    for (int q = 0; q < 4; ++q) p.fill_delta[q] = h.delta[q][c.fill_type];
    p.fill_delta3 = h.delta3[c.fill_type];
    p.fill_delta16[0] = h.delta16[0][c.fill_type];
    p.fill_delta16[1] = h.delta16[1][c.fill_type];
    int nspawn = 0;
    p.spawn_pat = 0xFFFFFFFFu;   // matches no valid type id
    for (int t = 0; t < c.num_types; ++t) {
        if (c.type_rule[t] != SGW_RULE_SPAWN) continue;
        ++nspawn;
        p.spawn_pat = 0x01010101u * (uint32_t)t;
        const uint64_t thr = prob_threshold(c.spawn_prob[t]);
        p.spawn_thr = (uint32_t)(thr & 0xFFFFFFFFull);
        p.spawn_full = thr >= 4294967296ull ? 1u : 0u;
        p.spawn_n = c.spawn_count[t];
        p.choice_lo = p.choice_hi = 0;
        for (int k = 0; k < c.spawn_count[t]; ++k) {
            if (k < 4) p.choice_lo |= (uint32_t)c.spawn_choice[t][k] << (8 * k);
            else p.choice_hi |= (uint32_t)c.spawn_choice[t][k] << (8 * (k - 4));
        }
    }
    p.single_spawner = nspawn <= 1 ? 1 : 0;
    p.onehot = onehot ? 1 : 0;
    p.nturns = 1;
    p.obs_A = c.num_agents;      // observations go to the [E][A][C][V][V] tensor unless a call says otherwise
    p.obs_a0 = 0;
    p.seed_lo = (uint32_t)c.seed;
    p.seed_hi = (uint32_t)(c.seed >> 32);
    p.first_env = (uint32_t)c.first_env_id;
    p.E = c.num_envs;
    p.dense_thr = prob_threshold(c.dense_prob);
    p.dense_count = c.dense_count;
    p.obs_post = c.obs_post;
    p.agent_rule = c.agent_rule;
    p.tag_it = c.tag_it_type;
    p.tag_notit = c.tag_notit_type;
    p.tag_reward = c.tag_reward;
    p.agent_mask = 0;
    for (int a = 0; a < c.num_agents; ++a) p.agent_mask |= 1u << (c.agent_type[a] & 31u);
    if (c.agent_rule == SGW_AGENT_RULE_TAG) p.agent_mask |= (1u << (c.tag_it_type & 31u)) | (1u << (c.tag_notit_type & 31u));
    p.has_become = 0;
    p.become_mask = 0;
    for (int t = 0; t < c.num_types; ++t)
        if (c.type_rule[t] == SGW_RULE_BECOME_IF) {
            p.has_become = 1;
            p.become_mask |= 1u << t;
        }
    p.quiet0 = p.quiet1 = 0xFFFFFFFFu;         // (four pad bytes: never cells)
    for (int z = 0; z < c.layers; ++z) {
        const uint32_t t = c.layer_fill_type[z];
        if (t >= (uint32_t)c.num_types || c.type_rule[t] != SGW_RULE_NONE) continue;
        const uint32_t w = 0x01010101u * t;
        if (p.quiet0 == 0xFFFFFFFFu || p.quiet0 == w) p.quiet0 = w;
        else if (p.quiet1 == 0xFFFFFFFFu || p.quiet1 == w) p.quiet1 = w;
    }
    p.kind_pack = 0;
    for (int a = 0; a < c.num_actions; ++a) p.kind_pack |= (uint32_t)(c.action_kind[a] & 3u) << (2 * a);
    p.beam_radius = c.beam_radius;
    p.clean_beam = c.clean_beam_type;
    p.zap_beam = c.zap_beam_type;
    p.beam_block_mask = c.beam_block_mask;
    p.total_factor = c.reward_total_factor > 0 ? c.reward_total_factor : 1;

    // ---- group geometry: one wave per env while a slice stays small, else a workgroup per env
    const bool plain_move = c.agent_rule == SGW_AGENT_RULE_MOVE;   // step_big implements MovingAgent.act only
    const bool tagk = c.agent_rule == SGW_AGENT_RULE_TAG;
    const bool simple_rules = !p.has_become && c.agent_rule != SGW_AGENT_RULE_CLEANUP;   // else: generic kernel
    const bool vec16 = (p.env_stride & 15) == 0 && p.env_stride >= p.cells_pad;   // 16-byte loads/stores per env are legal
    // Layered rule sets (BECOME_IF, Cleanup) stay on the wave-per-env RULES kernel up to 8 KiB per env: above 4 KiB their
    // alternative is the ticket-ordered workgroup-per-env generic kernel, where every act is a hand-off between waves
    // (Cleanup 48x48x3, 4 096 envs: 95 us there against 74.5 here); up to 11 KiB from 16 384 envs on, as for the plain worlds below
    // (Cleanup 56x64x3 at 4 096 / 16 384 / 65 536 envs 97 / 394 / 1 499 us on the generic kernel, 107 / 308 / 1 229 here)
    const int rules_units = (c.num_envs >= 16384 || o.rules_11k) ? kMaxUnitsPlain : kMaxUnitsRules;
    bool rules_8k = !simple_rules && !tagk && vec16 && p.cells_pad > 4096 && (p.cells_pad >> 4) <= 64 * rules_units && p.VV <= 128;
    if (!o.fast_rules || o.force_generic) rules_8k = false;
    // Plain and Tag worlds between 4 and 8 KiB per env: a LARGE batch of them also runs a wave per env (step_big spends a 512-thread
    // workgroup and three barriers on an env; per env that is about twice the time of the wave-per-env kernel, which pays only when the
    // batch is too small to fill the chip with waves).  tools/mid_world_probe.py, us per turn at 2 048 / 4 096 / 8 192 / 65 536 envs,
    // workgroup per env -> wave per env: 48x48x2 A8 r5 23 / 40 / 73 / 640 -> 22 / 31 / 56 / 373; 64x64x2 A16 r3 33 / 57 / 107 / 858 ->
    // 28 / 39 / 69 / 488; 50x50x2 A8 r3 26 / 46 / 84 / 686 -> 19 / 25 / 41 / 272; Tag 72x72 A16 r4 23 / 41 / 74 / 727 -> 25 / 34 / 61 / 447;
    // Tag 90x90 A12 r3 27 / 48 / 90 / 741 -> 20 / 25 / 48 / 347.  Option fast_8k = 0 / 1: never / whatever the batch.
    const bool fast_8k_ok = simple_rules && (onehot || (e->rgb16 && c.num_channels == 3)) && vec16 && nspawn <= 1 && p.cells_pad > 4096 && (p.cells_pad >> 4) <= 64 * kMaxUnitsPlain && p.VV <= 128;
    // (between 8 and 11 KiB -- three workgroups per CU -- from 16 384 envs on: 72x72x2 A8 r5 at 4 096 / 16 384 / 32 768 envs 47 / 166 / 429 ->
    // 55 / 156 / 341 us, Tag 100x100 A16 r4 43 / 219 / 429 -> 49 / 140 / 306; above that two workgroups per CU no longer pay: 90x90x2 555 -> 640)
    bool fast_8k = fast_8k_ok && c.num_envs >= (p.cells_pad <= 8192 ? 4096 : 16384);
    if (o.fast_8k == 0) fast_8k = false;
    if (o.fast_8k == 1) fast_8k = fast_8k_ok;
    if (o.force_generic) fast_8k = false;
    e->wpe = (p.cells_pad <= 4096 || rules_8k || fast_8k) ? 1 : 4;
    if (o.force_big && simple_rules && !o.force_generic) e->wpe = 4;
    // More than 64 agents (round 6): the wave- and workgroup-per-env kernels keep an agent per LANE of one wave; such worlds run on the generic kernel
    // with a workgroup per env (an agent phase = the work of one wave behind the LDS ticket, per-agent state in LDS arrays of SGW_MAX_AGENTS), any size
    const bool many_agents = c.num_agents > 64;
    if (many_agents) e->wpe = 4;
    const int epb = kBlock / (e->wpe * kWave);
    e->lds_bytes = (size_t)p.tab_bytes + (size_t)epb * p.env_lds;
    e->fast = e->wpe == 1 && vec16 && (p.cells_pad >> 4) <= 64 * (fast_8k ? kMaxUnitsPlain : kMaxUnits) && nspawn <= 1 && p.VV <= 128 && simple_rules;   // MovingAgent.act and TagAgent.act
    // the layered rule set on the wave-per-env kernel (RULES variant): any spawners, BECOME_IF rules, Cleanup or plain agents
    e->fast_rules = !e->fast && e->wpe == 1 && vec16 && (p.cells_pad >> 4) <= 64 * rules_units && p.VV <= 128 &&
                    !tagk && (p.cells_pad <= 4096 || rules_8k);
    if (!o.fast_rules) e->fast_rules = false;   // test hook: generic kernel instead
    e->fast = e->fast || e->fast_rules;
    if (many_agents) e->fast = e->fast_rules = false;
    // fast kernel: wave-private LDS = [one-hot counter words | appearance table][grid]
    // (the integer-table RGB instances exist for three channels, plain or Tag movers, worlds <= 4 KiB)
    const bool rgb16_fast = e->rgb16 && e->fast && !e->fast_rules && c.num_channels == 3;
    const bool bytes_ok = onehot || rgb16_fast;        // the byte-staging window pipeline applies
    e->fast_tab_bytes = bytes_ok ? 4 * SGW_MAX_TYPES * 4 : SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8;
    bool agents_impassable = true;
    for (int a = 0; a < c.num_agents; ++a) agents_impassable = agents_impassable && !c.type_passable[c.agent_type[a]];
    const bool tag_move = tagk;      // TagAgent.act moves like MovingAgent.act; step_big<..., TAG> walks the "it" token
    e->big = e->wpe == 4 && vec16 && nspawn <= 1 && p.VV <= 128 && agents_impassable && (plain_move || tag_move) && simple_rules && !many_agents;
    e->big_threads = kBigThreads;
    if (e->big) e->big_threads = big_threads_for(o, onehot, c.num_agents, p.VV);
    bool stage_kernel = false;   // a STAGE kernel (bursts of agents) applies
    const int ob_elems = c.num_agents * c.num_channels * p.VV;
    // a compile-time-shape instance of step_fast whose whole env leaves in ONE burst (the headline's way): a prebuilt one, or -- with
    // specialised instances -- any one-hot plain / Tag world of <= 4 KiB whose windows are a multiple of 4 elements and <= 4 KiB of bytes
    bool fixed_shape = !e->fast_rules && fixed_fast_shape(c.layers, c.num_channels, c.vision_radius, c.height, c.width, tagk);
    if (jit && !fixed_shape && e->fast && !e->fast_rules && onehot && p.cells_pad <= 4096 && (ob_elems & 3) == 0 && ob_elems <= 4096 && o.burst != 2) {
        // ... while the wave's LDS (tables + grid + the env's window bytes) still lets SIX workgroups share a CU; beyond that the
        // chunked bursts keep the occupancy (option burst = 1: whenever legal).  (profiles/r04_jit_probe.txt, whole / chunks: 32x32x2 C8
        // 143.6 / 148.0 us, 30x30 r4 174.3 / 183.0, 40x40 163.3 / 165.6 -- six per CU; 32x32x3 C10, five per CU: 191.2 / 182.3)
        const size_t per_wave = (size_t)e->fast_tab_bytes + p.cells_pad + ((ob_elems + 15) & ~15);
        fixed_shape = o.burst == 1 || per_wave * 4 + 1024 <= kLdsPerCu / 6;
    }
    if (jit && (o.burst == 2 || o.stage_agents >= 0)) fixed_shape = false;   // (a forced burst size asks for the chunked emit)
    {   // LDS staging of one-hot observations
        const int per_agent = c.num_channels * p.VV;
        e->obs_stage = 0;
        if (e->fast && onehot && fixed_shape) {   // (never an RGB world: six channels)
            // whole envs of a multiple of 4 elements, at most 4 KiB of byte counts
            if ((ob_elems & 3) == 0 && ob_elems <= 4096) e->obs_stage = (ob_elems + 15) & ~15;
        } else if (e->fast && bytes_ok) {
            // run-time shapes: as many agents per burst as fit the wave's share of LDS at full occupancy (8 workgroups
            // of 4 waves per CU, 1 KiB granules: 5 120 bytes per wave); if not even one agent fits, at 5 workgroups per CU
            const int base = e->fast_tab_bytes + (e->fast_rules ? kRuleLds : 0) + p.cells_pad;
            // The RULES kernels (layered rule sets: big windows over three layers) stage for FIVE workgroups per CU: longer bursts and
            // fewer half-written observation streams open at once beat the extra waves, as for the occupancy cap of the plain
            // kernels -- Cleanup 21x31x3 at 65 536 envs, agents per burst 1 / 2 / 3 / 4 / 5 / 10: 666 / 695 / 643-680 / 640 / 643 / 850 us
            // (16 384 envs: 201 / 193 / 185-190 / 188 / 181 / 240).
            // instances with a run-time channel count write their planes in groups of four: up to three planes of slack behind a chunk
            const bool static_channels = jit || (e->fast_rules ? (c.layers == 3 && c.num_channels == 9 && c.vision_radius == 5 && c.height == 21 && c.width == 31 && o.pack3)
                                                                 : (!tagk && c.layers == 2 && c.num_channels == 6));
            const int slack = 48 + (static_channels ? 0 : 3 * p.VV);
            int budget = e->fast_rules ? (int)((kLdsPerCu / 5 - 1024) / 4) - base - slack : (int)(kLdsPerCu / 8 / 4) - base - slack;
            for (int wg = 4; (e->fast_rules || fast_8k) && budget < per_agent && wg >= 2; --wg)      // big envs: fewer workgroups per CU until a window fits
                budget = (int)((kLdsPerCu / wg - 1024) / 4) - base - slack;
            if (budget < per_agent) budget = (int)((kLdsPerCu / 5 - 1024) / 4) - base - slack;
            int apc = budget >= per_agent ? std::min(c.num_agents, budget / per_agent) : 0;
            if (o.stage_agents >= 0) apc = std::min(c.num_agents, o.stage_agents);   // A/B hook
            if (apc > 0) {
                e->stage_agents = apc;
                e->obs_stage = (apc * per_agent + slack - 48 + 31 + 15) & ~15;   // + 31: the chunk's offset from a 128-byte line of global memory (step_fast.h: emit_chunk)
                stage_kernel = true;
            }
        }
        if (!o.stage) { e->obs_stage = 0; stage_kernel = false; e->stage_agents = 0; }   // test / tuning hook
    }
    if (o.force_generic) e->fast = e->big = e->fast_rules = false;   // test hook: the generic kernel on shapes the specialised ones would take
    // Small worlds: two or four envs per wave on the LDS-resident generic kernel (step_kernel<16 / 32>).  A wave-per-env
    // kernel spends most of a small world's life on per-env work that keeps few lanes busy (a 21x21x2 world: 29 of 64
    // lanes in the sweep, 25 in the 5x5 gather, one in the moves), and at ~700 instructions per env it is bound by
    // instruction issue, not memory; packed, that stream is shared.  Rule from tools/group_sweep.py (65 536 envs, us per
    // step, wave-per-env / 16 lanes / 32 lanes per env -- profiles/r02_group_sweep.txt):
    //   10x10 A2 r2 88/29/42   16x16 A4 r2 81/46/55   21x21 A2 r2 90/46/55   21x21 A8 r2 132/118/105
    //   24x24 A4 r3 134/137/124   32x32 A2 r2 93/96/78   28x28 A8 r3 167/243/223   32x32 A8 r3 118/282/206
    //   Tag 11x11 A5 r4 171/155/111   Tag 32x32 A8 r3 194/156/140   Cleanup 21x31x3 A10 r5 694/1383/1246
    // i.e. pack while the observation work per env (A * V * V window cells) is small, and only for batches that still
    // fill the chip twice over once packed (a small batch is latency-bound: config 2, 4 096 envs, 11 us wave-per-env
    // against 16-19 us packed).  Option group = 16 / 32 forces a packing, 64 forbids it.
    e->group = e->wpe * kWave;
    if (e->wpe == 1) {
        const int64_t avv = (int64_t)c.num_agents * p.VV;
        // (round 3: the batch a packing needs, re-measured on the single-turn instances -- us per step at 1 024 / 4 096 / 8 192 / 16 384 /
        // 32 768 envs, wave per env | 32 lanes | 16 lanes: 16x16 A4 r2 7.6 / 11.1 / 15.9 / 26.1 / 45.4 | 8.6 / 10.2 / 12.6 / 19.3 / 31.2 |
        // 10.5 / 11.6 / 13.0 / 17.4 / 28.1; 10x10 A2 r2 7.2 / 9.2 / 12.8 / 21.1 / 36.8 | 7.2 / 8.2 / 10.1 / 15.1 / 24.1 | 7.5 / 8.0 / 9.1 /
        // 11.2 / 18.0: two envs per wave from 4 096 envs on, four from 12 288)
        auto enough = [&](int G) { return c.num_agents <= G && (int64_t)c.num_envs * G / kWave >= (G == 16 ? 3072 : 2048); };
        int g = 0;
        // (Tag on a map with a compile-time-shape wave-per-env instance stays there: 32x32 / 8 agents 117 us against 143 packed)
        // (round 3, tools/tag_group_probe.py, two envs per wave / wave per env: 11x11 A5 r4 99 / 131 us, 16x16 A4 r3 51 / 104, 20x20 A5 r4 116 / 127,
        // 24x24 A6 r3 80 / 125, 28x28 A6 r3 96 / 115, but 30x30 A6 r4 183 / 128, 32x32 A8 r4 217 / 148, 40x40 A8 r3 163 / 135, 48x48 A10 r4 323 / 180, 64x64 A8 r3 221 / 183
        // (wave-per-env: the 3-bit-counter Tag instance): pack while map bytes + 2 x window cells of all agents stay below 1 500)
        if (tagk)
            g = (enough(32) && p.cells_pad + 2 * avv < 1500 && !(e->fast && fixed_fast_shape(c.layers, c.num_channels, c.vision_radius, c.height, c.width, true))) ? 32 : 0;
        else if (c.agent_rule == SGW_AGENT_RULE_MOVE && !p.has_become) {
            // (round 3, profiles/r03_group_sweep.txt -- the wave-per-env kernels have gained more than the packed ones since the rule
            // was set: 24x24 A4 r3 98 / 157 / 122 us, 32x32 A8 r2 136 / 193 / 152, 32x32 A4 r3 90 / 197 / 144, while 21x21 A8 r2
            // 122 / 132 / 105 and 32x32 A2 r2 94 / 104 / 80 still pack: two envs per wave only while the windows OR the map are small)
            if (avv <= 100 && p.cells_pad <= 1024 && enough(16)) g = 16;
            else if (avv <= 200 && (avv <= 100 || p.cells_pad <= 1024) && enough(32)) g = 32;
        }
        if (o.group) g = o.group;
        const bool fits = (g == 16 || g == 32) && c.num_agents <= g &&
                          (c.agent_rule != SGW_AGENT_RULE_CLEANUP || 3 * c.beam_radius <= g);
        if (fits) {
            e->group = g;
            e->fast = e->fast_rules = false;
        }
    }
    if (!e->fast) { e->obs_stage = 0; stage_kernel = false; e->stage_agents = 0; }
    e->rgb16 = rgb16_fast && e->fast && stage_kernel;      // the I16 instances are STAGE kernels: no staging area, no integer path
    if (!onehot && !e->rgb16) {
        e->fast_tab_bytes = SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8;
        e->obs_stage = 0;
        stage_kernel = false;
        e->stage_agents = 0;
    }
    e->whole_env_burst = e->fast && onehot && fixed_shape && e->obs_stage > 0;
    // what the float64 kernel (calls the STAGE kernel cannot serve: agent ranges, OBS_NEXT, unaligned tensors) needs instead
    e->plain_tab_bytes = e->rgb16 ? SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8 : 0;
    const int epb_step = (e->fast || e->big) ? epb : kBlock / e->group;   // envs per workgroup of the step kernel
    e->step_env_lds = e->fast ? e->fast_tab_bytes + (e->fast_rules ? kRuleLds : 0) + p.cells_pad + e->obs_stage : p.env_lds;
    e->step_lds_bytes = e->fast ? (size_t)epb * e->step_env_lds + (e->rgb16 ? 1024 : 0) : (size_t)p.tab_bytes + (size_t)epb_step * e->step_env_lds;   // (+ the I16 result table)
    p.big_pitch = c.width;
    if (e->big) {
        // LDS of a workgroup: [counter words of the channels in use | appearance table][agent arrays][grid image][staging].
        // Staging of the one-hot windows (a wave's window leaves as line-aligned 16-byte streaming stores, step_big.h phase
        // R): on for instances with compile-time tables (config 5: 434 -> 347-372 us per turn at 8 192 envs); an instance with run-time
        // tables pays more for the byte staging than the stores give back (64x64 / 16 agents / 7x7 windows: 229 -> 270 us) and is
        // compiled without it.  Option big_stage = 0: never.
        // Padded rows (W + 16: the ~3 rows a 32-lane group of the window gather touches fall on disjoint banks) where a row
        // is whole 16-byte units -- unless the padding costs a workgroup per CU (LDS is handed out in 1 KiB granules): with
        // the staging, config 5's image fits four times into a CU only unpadded (39 936 bytes: exactly), and a fourth workgroup is
        // worth more than the conflict-free gather (1 280 envs: 55 us at four per CU, 71 at three).
        const bool static_tables = onehot && (jit || (tagk ? (c.layers == 1 && c.num_channels == 4 && c.vision_radius == 4)
                                                           : (c.layers == 2 && c.num_channels == 6 && c.vision_radius == 5)));   // = pick_big's compile-time instances
        e->big_tab_bytes = static_tables ? ((c.num_channels + 3) / 4) * SGW_MAX_TYPES * 4 : e->fast_tab_bytes;   // (the run-time instance adds all four counter words)
        const size_t fixed = (size_t)e->big_tab_bytes + big_agent_lds(tagk);
        bool stage_on = static_tables && !tagk;   // (the Tag example's 9x9x4 windows are ten lines each: staged 54 / 79 us, direct 47 / 74, 128x128 at 2 048 envs / 72x72 at 8 192)
        if (o.big_stage == 0) stage_on = false;
        e->big_stage = stage_on ? (c.num_channels * p.VV + 31 + 3) & ~3 : 0;
        const size_t stage_all = (size_t)(e->big_threads / 64) * e->big_stage;
        auto per_cu = [&](size_t bytes) { return std::min<size_t>(4, kLdsPerCu / (((bytes + 1023) & ~(size_t)1023) + 1024)); };   // (a workgroup's request must stay 1 KiB below its share)
        const size_t plain_img = (size_t)p.cells_pad, padded_img = (size_t)c.layers * c.height * (c.width + 16);
        const bool can_pad = (c.width & 15) == 0 && (p.cells & 15) == 0;
        if (can_pad && per_cu(fixed + padded_img + stage_all) >= per_cu(fixed + plain_img + stage_all)) p.big_pitch = c.width + 16;
        e->step_lds_bytes = fixed + (p.big_pitch == c.width ? plain_img : padded_img);
        p.big_stage = p.big_stage_off = 0;
        if (e->big_stage) {
            p.big_stage_off = (int)e->step_lds_bytes;        // (a multiple of 16: every piece before it is)
            e->step_lds_bytes += stage_all;
        }
    }
    const size_t lds_max = 160 * 1024;
    if (e->lds_bytes > lds_max)
        return fail(SGW_EINVAL, "world of %d bytes per env does not fit the %zu-byte LDS-resident path", p.cells, lds_max);

    // ---- the instances
    const int L = c.layers, C = c.num_channels, r = c.vision_radius, H = c.height, W = c.width;
    const bool static_map = p.cells_pad <= 4096;   // step_fast: the whole grid in NU <= 4 register units per lane
    if (e->fast) {
        e->k_plain.host = pick_fast(o, e->onehot, false, L, C, r, H, W, tagk, e->fast_rules, false, &e->k_plain.host_name);
        e->k_step.host = pick_fast(o, e->onehot, e->rgb16, L, C, r, H, W, tagk, e->fast_rules, stage_kernel, &e->k_step.host_name);
        e->k_multi.host = pick_fast_multi(o, e->onehot, L, C, r, H, W, tagk, e->fast_rules, stage_kernel, &e->k_multi.host_name);
        if (jit) {
            const int jh = static_map ? H : 0, jw = static_map ? W : 0;
            if (e->whole_env_burst) {
                e->k_step.want = fast_id(true, L, C, r, jh, jw, tagk, false, false, false, false, false);
                e->k_multi.want = tagk ? "" : fast_id(true, L, C, r, jh, jw, false, false, false, true, false, false);
                if (!fixed_fast_shape(L, C, r, H, W, tagk)) e->k_step.host = e->k_multi.host = nullptr;   // (no prebuilt twin stages a whole env of this shape)
                e->k_plain = Kernel();                                                                     // the same instance serves agent ranges with direct stores
            } else {
                e->k_step.want = fast_id_like(e->k_step.host_name, L, C, r, jh, jw, -1, -1);
                e->k_plain.want = fast_id_like(e->k_plain.host_name, L, C, r, jh, jw, -1, -1);
                if (e->k_multi.host) e->k_multi.want = fast_id_like(e->k_multi.host_name, L, C, r, jh, jw, -1, -1);
                else if (stage_kernel && e->onehot && !tagk && !e->rgb16) e->k_multi.want = fast_id_like(e->k_step.host_name, L, C, r, jh, jw, -1, 1);
            }
        } else if (e->whole_env_burst) {
            e->k_plain = Kernel();
        }
        // the policy turn's first launch into per-agent rows (sgw_sweep_observe_rows): plain movers whose env leaves as one burst
        if (e->whole_env_burst && static_map && !e->fast_rules && ((C * (2 * r + 1) * (2 * r + 1)) & 1) == 0) {      // (round 6: Tag movers too)
            if (L == 2 && C == 6 && r == 3 && H == 32 && W == 32 && !tagk) {
                e->k_sweep_rows.host = reinterpret_cast<const void*>(&step_fast_rows<2, 6, 3, 32, 32>);
                e->k_sweep_rows.host_name = "step_fast_rows<2, 6, 3, 32, 32>";
            }
            if (jit) e->k_sweep_rows.want = fast_rows_id(L, C, r, H, W, tagk);
            if (jit) e->k_sweep_rows_tail.want = fast_rows_id(L, C, r, H, W, tagk, true);       // (compiled when a tail is bound: sgw_bind_row_tail)
        }
        // ... and on the chunk-staging instances (layered rule sets, Tag, run-time maps): the ROWX twin, specialised only (round 6)
        e->sweep_rows_chunked = false;
        if (jit && !e->whole_env_burst && stage_kernel && e->onehot && !e->rgb16 && e->stage_agents >= 1 && e->k_step.host_name) {
            const int jh = static_map ? H : 0, jw = static_map ? W : 0;
            e->k_sweep_rows.want = fast_rowsx_id_like(e->k_step.host_name, L, C, r, jh, jw);
            e->sweep_rows_chunked = !e->k_sweep_rows.want.empty();
        }
    } else if (e->big) {
        e->k_step.host = pick_big(e->onehot, L, C, r, tag_move, e->big_threads, &e->k_step.host_name);
        if (!tag_move) e->k_multi.host = pick_big_multi(e->onehot, L, C, r, &e->k_multi.host_name);
        if (!tag_move && ((p.cells + 15) >> 4) <= 4 * e->big_threads)   // the prefetch holds one 4-unit round per thread
            e->k_walk.host = pick_big_walk(e->onehot, L, C, r, e->big_threads, &e->k_walk.host_name);
        if (jit) {
            e->k_sweep_rows.want = big_id(e->onehot, L, C, r, false, false, tag_move, e->big_threads, true);   // (round 6: the fused sweep + rows launch; specialised only)
            e->k_step.want = big_id(e->onehot, L, C, r, false, false, tag_move, e->big_threads);
            if (!tag_move) e->k_multi.want = big_id(e->onehot, L, C, r, true, false, false, kBigThreads);   // (whether or not the library holds a twin)
            if (e->k_walk.host) e->k_walk.want = big_id(e->onehot, L, C, r, false, true, false, e->big_threads);
        }
    } else {
        if (many_agents) {      // (group == 256: set above)
            e->k_step.host = pick_step_many(e->onehot, c.agent_rule, false, &e->k_step.host_name);
            e->k_multi.host = nullptr;
        } else {
            e->k_step.host = pick_step(o, e->group, e->onehot, L, C, c.agent_rule, r, H, W, false, &e->k_step.host_name);
            e->k_multi.host = pick_step(o, e->group, e->onehot, L, C, c.agent_rule, r, H, W, true, &e->k_multi.host_name);   // the generic kernel's instance with the turn loop
        }
        if (jit) {
            e->k_step.want = generic_id(e->group, e->onehot, L, C, c.agent_rule, r, H, W, false, many_agents);
            e->k_multi.want = generic_id(e->group, e->onehot, L, C, c.agent_rule, r, H, W, true, many_agents);
            e->k_sweep_rows.want = generic_id(e->group, e->onehot, L, C, c.agent_rule, r, H, W, false, many_agents, true);   // (round 6: the fused sweep + rows launch)
        }
    }
    if (!o.big_walk) e->k_walk = Kernel();   // A/B hook
    for (Kernel* k : {&e->k_step, &e->k_plain, &e->k_multi, &e->k_walk, &e->k_sweep_rows})
        if (k->host && k->want == k->host_name) k->want.clear();   // the library already holds exactly this instance
    e->multi_turn = e->k_multi.usable();   // kernels with sgw_rollout's turn loop
    e->reset_fn = pick_reset(e->wpe);
    // Worlds above 4 KiB only: there, gathering one window from global memory beats staging 32 KiB through LDS (config 5:
    // 14.9 against 24.4 us per phase launch); a 2 KiB env is staged with four coalesced 16-byte loads per lane and the
    // byte gather from global is the slower way (config 3: 62.9 against 46.1 us).  Option phase_kernel = 0 / 1 forces.
    e->phase_ok = plain_move && e->wpe == 4;
    if (o.phase_kernel >= 0) e->phase_ok = plain_move && o.phase_kernel == 1;
    e->rows_epb = e->rows_wpb = 0;
    e->rows_lds = 0;
    if (e->onehot && p.cells >= 8 && o.phase_rows) {
        // (observe_rows renders windows whatever the agents do when they act: every agent rule; phase_rows moves plain movers only)
        const int NW = (C + 3) / 4;
        e->k_rows.host = pick_rows(L, NW, r, &e->k_rows.host_name, &e->k_obs_rows.host, &e->k_obs_rows.host_name);
        if (jit && !e->k_rows.host && r >= 1 && r <= 7) {
            e->k_rows.want = rows_id("phase_rows", L, NW, r);
            e->k_obs_rows.want = rows_id("observe_rows", L, NW, r);
        }
        if (!plain_move) e->k_rows = Kernel();
        if (e->k_obs_rows.usable()) {
            const int V = 2 * r + 1;
            e->rows_epb = e->rows_wpb = 4 * (64 / (V <= 4 ? 4 : (V <= 8 ? 8 : 16)));
            // per wave: counter words, the value table, the staging bytes of the windows it carries
            e->rows_lds = (size_t)4 * (NW * 34 * 4 + SGW_MAX_TYPES * 8 + (((e->rows_epb / 4) * C * V * V + 15) & ~15));
        }
    }
    p.stage_agents = e->stage_agents;
    e->fast_wg_cap = 5;
    e->fast_wg_cap_forced = false;
    if (o.fast_wg_per_cu > 0) { e->fast_wg_cap = o.fast_wg_per_cu; e->fast_wg_cap_forced = true; }   // tuning hook
    e->grid_blocks = (int)ceil_div(p.E, (e->fast || e->big) ? epb : epb_step);   // every step kernel: one env per group, the dispatcher balances
    if (e->k_walk.usable()) {
        // workgroups a CU holds at once: the walking variant is compiled for SGW_WALK_WAVES waves per SIMD (76 VGPRs: three 512-thread
        // workgroups per CU), the plain kernel for 6 (the hardware admits a fourth workgroup while the request stays 1 KiB below a quarter
        // of the CU's LDS -- 1 280 envs of config 5: 55 us there, 71 above)
        const size_t walk_lds = e->step_lds_bytes - (size_t)(e->big_threads / 64) * e->big_stage;   // (the walking variant stores directly: no staging area)
        const int per_cu = resident_per_cu(e->big_threads, walk_lds, SGW_WALK_WAVES);
        const int plain_per_cu = resident_per_cu(e->big_threads, e->step_lds_bytes, SGW_BIG_WAVES);
        // Engaged for batches of 1.5x to 3x what the plain kernel holds at once (one env per workgroup, four
        // workgroups per CU at config 5 = 1 024 envs), measured on config 5's shape, same box, us per launch, walking
        // against plain: 1 280 envs 53 / 55, 1 536 74-77 / 70-72, 2 048 88-96 / 109-118, 3 072 161-183 / 174-178,
        // 4 096 206 / 224, 8 192 511 / 436.  Fewer walking workgroups are resident (76 VGPRs: three per CU), they
        // run in lockstep and each env's prefetch waits for the previous env's stores, so over many rounds the
        // dispatcher's four per CU win; over two or three rounds the hidden drain does.  Also measured at 2 048 envs:
        // 683 workgroups (three envs each, evenly) 100 us, 512 100 us, 1 024 / 1 365 (oversubscribed) 93-107 us, a
        // 64-VGPR build (four per CU, six spilled registers) 95-98 us, staggered starts 96-101 us.
        e->walk_blocks = per_cu * e->num_cus;
        // Round 3, with the staged windows (config 5, us per launch, plain direct / walking direct / plain staged): 1 024 envs
        // 47 / - / 59, 1 280 54 / 54 / 69, 1 536 70 / 76 / 78, 2 048 105 / 91 / 97, 2 560 132 / 128 / 117, 3 072 155 / 168 / 141,
        // 4 096 216 / - / 174-181, 8 192 415 / - / 347: the window is 1.5x to 2.25x now, staging takes over above it.
        e->walk_min_envs = (int64_t)plain_per_cu * e->num_cus * 3 / 2;
        e->walk_max_envs = e->big_stage ? (int64_t)plain_per_cu * e->num_cus * 9 / 4 : (int64_t)plain_per_cu * e->num_cus * 3;
        if (o.big_walk_blocks > 0) {   // tuning / test hook: this many workgroups, whatever the batch
            e->walk_blocks = o.big_walk_blocks;
            e->walk_min_envs = e->walk_blocks;
            e->walk_max_envs = INT64_MAX;
        }
    }
    // staged windows pay once the batch is a few rounds of workgroups (a single round is latency-bound, and the staging adds
    // an LDS round trip per window): above 1.75x what the chip holds at once (see the table above)
    if (e->big) {
        const int64_t by_lds = (int64_t)(kLdsPerCu / (((e->step_lds_bytes + 1023) & ~(size_t)1023) + 1024));
        const int64_t resident = std::max<int64_t>(1, std::min<int64_t>(2048 / e->big_threads, by_lds));   // workgroups a CU holds at once
        e->big_stage_min_envs = resident * e->num_cus * 7 / 4;
        if (o.big_stage == 1) e->big_stage_min_envs = 0;   // test hook: staged whatever the batch
    }
    e->reset_blocks = (int)ceil_div(p.E, epb);   // one env per group and launch
    return SGW_OK;
}

// The specialised instance of `k`, compiled / loaded on first use.  A refusal leaves the prebuilt twin in charge (or, if the plan
// has none, is an error the caller reports).
int resolve_kernel(sgw_engine* e, Kernel& k) {
    if (k.jit || k.want.empty() || k.tried) return (k.jit || k.host) ? SGW_OK : fail(SGW_EHIP, "no instance of %s is available", k.want.c_str());
    k.tried = true;
    std::string err;
    k.jit = jit_get(k.want, e->opt, e->arch.c_str(), e->dev, &err);
    if (!k.jit && !k.host) return fail(SGW_EHIP, "specialising %s failed and the library holds no prebuilt twin: %s", k.want.c_str(), err.c_str());
    if (k.jit && e->step_lds_bytes > 65536)   // (what hipFuncSetAttribute does for the prebuilt instances)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k.jit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(e->lds_bytes, e->step_lds_bytes) + 16);
    return SGW_OK;
}

int launch_kernel(sgw_engine* e, Kernel& k, unsigned blocks, unsigned threads, size_t lds, hipStream_t s, Params& p, RowPtrs* rp) {
    if (!k.jit && !k.want.empty() && !k.tried)
        if (int rc = resolve_kernel(e, k)) return rc;
    static RowPtrs no_rows{};                 // (kernels that take the row pointers read them only when Params says so: rows_on, ...)
    void* args[2] = {&p, rp ? rp : &no_rows};
    if (k.jit) HIP_TRY(hipModuleLaunchKernel(k.jit, blocks, 1, 1, threads, 1, 1, (unsigned)lds, s, args, nullptr));
    else if (k.host) HIP_TRY(hipLaunchKernel(k.host, dim3(blocks), dim3(threads), args, lds, s));
    else return fail(SGW_EHIP, "no kernel to launch");
    return SGW_OK;
}

// Waits for the recorded event pairs and folds them into the running sum and the per-launch series.
int time_drain(sgw_engine* e) {
    if (e->ev_used == 0) return SGW_OK;
    for (int i = 0; i < e->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(e->ev1[i]));   // each pair on its own: timed launches may have gone to different streams
        HIP_TRY(hipEventElapsedTime(&ms, e->ev0[i], e->ev1[i]));
        e->ms_acc += ms;
        if (e->series.size() < kSeriesCap) e->series.push_back(ms);
        else e->series_dropped++;
    }
    e->ev_used = 0;
    return SGW_OK;
}

int time_begin(sgw_engine* e, hipStream_t s) {
    if (!e->timing) return SGW_OK;
    if (e->ev_used == kEventPool)   // the pool wraps: the one place a launch call waits (include/sgw.h, Conventions)
        if (int rc = time_drain(e)) return rc;
    HIP_TRY(hipEventRecord(e->ev0[e->ev_used], s));
    return SGW_OK;
}

int time_end(sgw_engine* e, hipStream_t s) {
    if (!e->timing) return SGW_OK;
    HIP_TRY(hipEventRecord(e->ev1[e->ev_used], s));
    e->ev_used++;
    e->launches++;
    return SGW_OK;
}

}  // namespace

extern "C" {

const char* sgw_last_error(void) { return g_err; }
#ifdef SGW_STAMPS
int sgw_debug_stamps(unsigned long long* out) {   // diagnostic builds only; not part of the ABI: [kStampEnvs][8] of the last launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * kStampEnvs) == hipSuccess ? 0 : -1;
}
#endif

const char* sgw_version(void) { return "sgw 0.2 (gfx950)"; }

int64_t sgw_obs_elems_per_env(const sgw_config* c) {
    const int64_t V = 2 * c->vision_radius + 1;
    return (int64_t)c->num_agents * c->num_channels * V * V;
}
int64_t sgw_grid_bytes_per_env(const sgw_config* c) { return (int64_t)c->layers * c->height * c->width; }
int64_t sgw_algorithmic_bytes_per_env_step(const sgw_config* c) {
    const int64_t V = 2 * c->vision_radius + 1;
    // grid read+write, per agent obs f32 store + action + reward + pos load/store, total f64 rw
    return 2 * sgw_grid_bytes_per_env(c) + (int64_t)c->num_agents * (c->num_channels * V * V * 4 + 1 + 4 + 4) + 16;
}

int sgw_set_option(sgw_engine* e, const char* key, const char* value) {
    int rc;
    if (e) {
        rc = option_set(e->opt, key, value, true);
    } else {
        std::lock_guard<std::mutex> lock(g_opt_mu);
        rc = option_set(g_opts, key, value, false);
    }
    if (rc == 1) return fail(SGW_EINVAL, "sgw_set_option: unknown key '%s'", key ? key : "(null)");
    if (rc == 2) return fail(SGW_EINVAL, "sgw_set_option: value '%s' is out of range for '%s'", value ? value : "(null)", key);
    if (rc == 3) return fail(SGW_EINVAL, "sgw_set_option: '%s' shapes the plan of an engine: set it (with a NULL engine) before sgw_create", key ? key : "(all keys)");
    return SGW_OK;
}

static int sgw_debug(void) {   // the ONE environment variable the shipped library reads: SGW_DEBUG=1 turns on the specialiser's log lines
    static const int v = [] { const char* f = getenv("SGW_DEBUG"); return (f && f[0] == '1') ? 1 : 0; }();
    return v;
}

// what plan_engine decided, for sgw_plan / sgw_launch_info
static void describe_plan(const sgw_engine* e, sgw_plan_info* out) {
    memset(out, 0, sizeof(*out));
    out->family = e->fast ? SGW_FAMILY_WAVE : (e->big ? SGW_FAMILY_WORKGROUP : SGW_FAMILY_GENERIC);
    out->lanes_per_env = (e->fast || e->big) ? (e->big ? e->big_threads : kWave) : e->group;
    out->threads = e->big ? e->big_threads : kBlock;
    out->grid_blocks = e->grid_blocks;
    out->lds_bytes = (int64_t)e->step_lds_bytes;
    out->env_lds = e->step_env_lds;
    out->obs_stage = e->obs_stage;
    out->stage_agents = e->stage_agents;
    out->whole_env_burst = e->whole_env_burst ? 1 : 0;
    out->big_stage = e->big_stage;
    out->big_pitch = e->base.big_pitch;
    out->onehot = e->onehot ? 1 : 0;
    out->rgb16 = e->rgb16 ? 1 : 0;
    out->rules = e->fast_rules ? 1 : 0;
    out->specialised = e->jit ? 1 : 0;
    out->phase_kernel = e->phase_ok ? 1 : 0;
    out->rollout_in_one_launch = e->multi_turn ? 1 : 0;
    out->walk_blocks = e->walk_blocks;
    out->walk_min_envs = e->walk_min_envs;
    out->walk_max_envs = e->walk_max_envs;
    out->big_stage_min_envs = e->big_stage_min_envs;
    auto put = [](char* dst, size_t cap, const Kernel& k, bool prebuilt) {
        const char* s = prebuilt ? (k.host ? k.host_name : "-") : (k.want.empty() ? (k.host ? k.host_name : "-") : k.want.c_str());
        snprintf(dst, cap, "%s", s);
    };
    put(out->kernel, sizeof(out->kernel), e->k_step, false);
    put(out->kernel_prebuilt, sizeof(out->kernel_prebuilt), e->k_step, true);
    put(out->kernel_plain, sizeof(out->kernel_plain), e->k_plain, false);
    put(out->kernel_rollout, sizeof(out->kernel_rollout), e->k_multi, false);
    put(out->kernel_walk, sizeof(out->kernel_walk), e->k_walk, false);
    put(out->kernel_phase, sizeof(out->kernel_phase), e->k_rows, false);
    if (!e->k_rows.usable()) snprintf(out->kernel_phase, sizeof(out->kernel_phase), "%s", e->phase_ok ? (e->onehot ? "phase_kernel<true>" : "phase_kernel<false>") : "the step kernel");
    put(out->kernel_observe_rows, sizeof(out->kernel_observe_rows), e->k_obs_rows, false);
}

int sgw_plan(const sgw_config* cfg, int32_t num_cus, int64_t lds_per_workgroup, sgw_plan_info* out) {
    if (!out) return fail(SGW_EINVAL, "sgw_plan: out is NULL");
    if (int rc = validate(cfg)) return rc;
    sgw_engine* e = new (std::nothrow) sgw_engine();
    if (!e) return fail(SGW_ENOMEM, "out of host memory");
    e->cfg = *cfg;
    {
        std::lock_guard<std::mutex> lock(g_opt_mu);
        e->opt = g_opts;
    }
    e->num_cus = num_cus > 0 ? num_cus : 256;
    e->lds_cap = lds_per_workgroup > 0 ? (size_t)lds_per_workgroup : 65536;
    // (whether hipRTC can be loaded is a property of the machine, not of the plan: sgw_plan answers for a machine that has it
    // unless the option says otherwise; sgw_create re-plans with jit = false when a compile is refused)
    const int rc = plan_engine(e, e->opt.jit != 0 && SGW_JIT_SOURCES);
    if (rc == SGW_OK) describe_plan(e, out);
    delete e;
    return rc;
}

int sgw_jit_compile(const char* instance, const char* arch, char* path_out, int64_t capacity) {
    if (!instance || !arch) return fail(SGW_EINVAL, "sgw_jit_compile: NULL argument");
    Options o;
    {
        std::lock_guard<std::mutex> lock(g_opt_mu);
        o = g_opts;
    }
    std::lock_guard<std::mutex> lock(g_jit_mu);
    std::string lowered, code, path, err;
    bool from_disk = false;
    const auto t0 = std::chrono::steady_clock::now();
    if (!jit_build(instance, o, arch, &lowered, &code, &path, &from_disk, &err)) return fail(SGW_EHIP, "sgw_jit_compile: %s", err.c_str());
    if (!from_disk) {
        g_jit_stats.compiled++;
        g_jit_stats.compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } else {
        g_jit_stats.disk_hits++;
    }
    if (path_out && capacity > 0) snprintf(path_out, (size_t)capacity, "%s", path.c_str());
    return SGW_OK;
}

int sgw_jit_stats(double* out6) {
    if (!out6) return fail(SGW_EINVAL, "sgw_jit_stats: NULL argument");
    std::lock_guard<std::mutex> lock(g_jit_mu);
    out6[0] = (double)g_jit_stats.compiled; out6[1] = (double)g_jit_stats.disk_hits; out6[2] = (double)g_jit_stats.mem_hits;
    out6[3] = (double)g_jit_stats.failed; out6[4] = g_jit_stats.compile_ms; out6[5] = g_jit_stats.load_ms;
    return SGW_OK;
}

int sgw_create(const sgw_config* cfg, sgw_engine** out) {
    if (!out) return fail(SGW_EINVAL, "out is NULL");
    *out = nullptr;
    if (int rc = validate(cfg)) return rc;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));

    sgw_engine* e = new (std::nothrow) sgw_engine();
    if (!e) return fail(SGW_ENOMEM, "out of host memory");
    e->cfg = *cfg;
    {
        std::lock_guard<std::mutex> lock(g_opt_mu);
        e->opt = g_opts;
    }
    if (sgw_debug()) e->opt.jit_verbose = 1;
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->lds_cap = prop.sharedMemPerBlock > 0 ? prop.sharedMemPerBlock : 65536;
    e->dev = dev;
    e->arch = prop.gcnArchName[0] ? prop.gcnArchName : "gfx950";

    // the plan; EVERY specialised instance it counts on is compiled / loaded now, and a refusal of any of them (no hipRTC, no embedded
    // sources, a compile error, a code object that does not load) re-plans for the prebuilt instances.  (Until round 5 only the whole-turn
    // kernel was resolved here and the others at first use: a late refusal then left a prebuilt run-time-shape twin running under a plan
    // laid out for compile-time shapes -- no slack behind the staged chunk for the planes a run-time channel count writes in groups of
    // four -- and sgw_capabilities promised row kernels that existed only specialised.)
    bool jit = e->opt.jit != 0 && SGW_JIT_SOURCES;
    for (;;) {
        if (int rc = plan_engine(e, jit)) { delete e; return rc; }
        if (!jit) break;
        bool ok = true;
        {   // (the ones that have to be compiled: in one program, jit.h)
            std::vector<std::string> names;
            for (Kernel* k : {&e->k_step, &e->k_plain, &e->k_multi, &e->k_walk, &e->k_rows, &e->k_obs_rows, &e->k_sweep_rows})
                if (!k->want.empty()) names.push_back(k->want);
            jit_prefetch(names, e->opt, e->arch.c_str(), e->dev);
        }
        for (Kernel* k : {&e->k_step, &e->k_plain, &e->k_multi, &e->k_walk, &e->k_rows, &e->k_obs_rows, &e->k_sweep_rows}) {
            if (k->want.empty()) continue;
            std::string err;
            k->tried = true;
            k->jit = jit_get(k->want, e->opt, e->arch.c_str(), e->dev, &err);
            if (!k->jit) { ok = false; break; }
        }
        if (ok) break;
        jit = false;
    }
    const Params& pc = e->base;
    (void)pc;

    hipError_t err = hipMalloc(&e->d_tab, sizeof(DevTables));
    if (err == hipSuccess) err = hipMemcpy(e->d_tab, &e->h_tab, sizeof(DevTables), hipMemcpyHostToDevice);
    {   // the reset image: per layer the fill type, the border type around it; bytes past the last cell are not cells
        const sgw_config& c = e->cfg;
        std::vector<uint8_t> img((size_t)e->base.cells_pad, (uint8_t)0xFF);
        for (int z = 0; z < c.layers; ++z)
            for (int y = 0; y < c.height; ++y)
                for (int x = 0; x < c.width; ++x) {
                    const bool edge = y == 0 || y == c.height - 1 || x == 0 || x == c.width - 1;
                    const uint8_t b = c.layer_border_type[z];
                    img[((size_t)z * c.height + y) * c.width + x] = (b != SGW_NO_BORDER && edge) ? b : c.layer_fill_type[z];
                }
        if (err == hipSuccess) err = hipMalloc(&e->d_tmpl, img.size());
        if (err == hipSuccess) err = hipMemcpy(e->d_tmpl, img.data(), img.size(), hipMemcpyHostToDevice);
    }
    if (err == hipSuccess) err = hipMalloc(&e->d_status, 4 * sizeof(int));
    if (err == hipSuccess) err = hipMemset(e->d_status, 0, 4 * sizeof(int));
    if (err == hipSuccess) err = hipMalloc(&e->d_part, 2 * kRedBlocks * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&e->d_turn, sizeof(TurnState));
    if (err == hipSuccess) err = hipMemset(e->d_turn, 0, sizeof(TurnState));
    if (e->cfg.num_envs >= 8192) {     // sgw_turn_resolve's scan-built dirty list (large batches): allocated HERE, not inside a stream-ordered (capturable) call
        const size_t nb = (size_t)ceil_div(e->cfg.num_envs, 256);
        if (err == hipSuccess) err = hipMalloc(&e->d_dcount, (size_t)e->cfg.num_envs);
        if (err == hipSuccess) err = hipMalloc(&e->d_doffsets, 2 * nb * sizeof(uint32_t));      // block sums | block offsets
        if (err == hipSuccess) err = hipMemset(e->d_doffsets, 0, 2 * nb * sizeof(uint32_t));
    }
    if (err != hipSuccess) {
        sgw_destroy(e);
        return fail(SGW_EHIP, "device allocation failed: %s", hipGetErrorString(err));
    }
    e->base.tab = e->d_tab;
    e->base.tmpl = e->d_tmpl;
    e->base.status = e->d_status;

    if (std::max(e->lds_bytes, e->step_lds_bytes) > std::min<size_t>(e->lds_cap, 65536)) {
        err = hipSuccess;
        for (Kernel* k : {&e->k_step, &e->k_plain, &e->k_multi, &e->k_walk})
            if (err == hipSuccess && k->host)
                err = hipFuncSetAttribute(k->host, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds_bytes + 16);   // (+ the walking variant's hand-over word)
        for (Kernel* k : {&e->k_step, &e->k_plain, &e->k_multi, &e->k_walk})
            if (err == hipSuccess && k->jit)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k->jit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds_bytes + 16);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(e->reset_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
        if (err != hipSuccess) {
            sgw_destroy(e);
            return fail(SGW_EHIP, "cannot reserve %zu bytes of LDS: %s", e->lds_bytes, hipGetErrorString(err));
        }
    }
    *out = e;
    return SGW_OK;
}

void sgw_destroy(sgw_engine* e) {
    if (!e) return;
    for (hipEvent_t ev : e->ev0) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev1) (void)hipEventDestroy(ev);
    if (e->d_tab) (void)hipFree(e->d_tab);
    if (e->d_tmpl) (void)hipFree(e->d_tmpl);
    if (e->d_status) (void)hipFree(e->d_status);
    if (e->d_part) (void)hipFree(e->d_part);
    if (e->d_dcount) (void)hipFree(e->d_dcount);
    if (e->d_doffsets) (void)hipFree(e->d_doffsets);
    if (e->d_turn) (void)hipFree(e->d_turn);
    delete e;
}

static int launch_reset(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, double* total_reward, uint32_t epoch, void* stream) {
    const sgw_config& c = e->cfg;
    const int b = c.layer_border_type[c.agent_layer];
    if (b == SGW_NO_BORDER || c.type_passable[b])
        return fail(SGW_EINVAL, "sgw_reset: the agent layer needs an impassable border type (the reference has no bounds check in move)");
    if (epoch >= (1u << 28)) return fail(SGW_EINVAL, "epoch must be < 2^28");
    Params p = e->base;
    p.grid = grid; p.pos = agent_pos; p.total = total_reward; p.epoch = epoch;
    p.agent_state = e->agent_state;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(e->reset_fn, dim3(e->reset_blocks), dim3(kBlock), e->lds_bytes, s, p);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_reset(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, double* total_reward, uint32_t epoch, void* stream) {
    if (!e || !grid || !agent_pos || !total_reward) return fail(SGW_EINVAL, "sgw_reset: NULL argument");
    return launch_reset(e, grid, agent_pos, total_reward, epoch, stream);
}

// The dynamic-LDS request of a step launch, and the workgroup-per-CU cap it encodes (0 = none).  The cap is the one
// launch-time lever on occupancy; the policy (sgw_set_wg_per_cu: 0 = the automatic rule, 1..8 forced, -1 never):
// five workgroups per CU for whole-turn float32 observation writes of 8 KiB or more per env in large batches, where fewer
// concurrent waves mean fewer half-written lines open in HBM --
//  - the unstaged path (Cleanup, 21x31x3 at 65 536 envs: 893 -> 801 us);
//  - the staged path only when the grids of the batch no longer fit the caches, and SIX per CU there since its bursts sit on
//    128-byte lines (round 3, config 3's shape, us per launch at 8 / 7 / 6 / 5 / 4 per CU: 524 288 envs 1 321 / 1 061 / 1 082 / 1 167 / 1 347,
//    262 144 envs 640 / 598 / 566 / 595 / 679; before the aligned bursts five was best: 262 144 envs 662 -> 578 us, 524 288: 1375 -> 1146 us).  While they do fit (configs 3/4: 65 536 envs, 134 MB) the staged emit with its streaming
//    full-line stores is fastest at full occupancy (124 us at 8 and 7 per CU, 126 at 6, 131 at 5; 131 072 envs: 248 vs 281).
// The uint8 format, small batches and the shapes with small windows, which are latency-bound (Tag 11x11, 6.5 KB per env:
// 164 us at full occupancy, 192 us capped), are not capped.
static size_t step_lds_request(const sgw_engine* e, const Params& p, int* cap_out) {
    size_t lds = e->step_lds_bytes;
    int cap = 0;
    if (e->fast && e->wg_per_cu > 0) cap = e->wg_per_cu;
    else if (e->fast && e->wg_per_cu == 0 && e->fast_wg_cap > 0 && p.obs && !(p.flags & SGW_STEP_NO_OBS) && !p.obs_u8 &&
             (!p.obs_stage || (size_t)p.E * (size_t)p.env_stride > kCacheResidentGrid) && p.a1 == p.A && p.a0 == 0 &&
             (size_t)p.A * p.C * p.VV * 4 >= 8192 && p.E >= (int64_t)e->num_cus * 32 * 2)
        cap = (p.obs_stage && !e->fast_wg_cap_forced) ? 6 : e->fast_wg_cap;   // (staged, line-aligned bursts: six -- see above)
    if (cap > 0) lds = std::max(lds, (size_t)(kLdsPerCu / cap - 1024) & ~(size_t)511);   // 1 KiB below the share: LDS is handed out in 1 KiB granules
    if (cap_out) *cap_out = cap;
    return lds;
}

// A/B hook (option big_wg_per_cu): fewer step_big workgroups per CU than the code object admits, through the LDS request
static size_t big_cap_lds(const sgw_engine* e, size_t lds) {
    if (!e->big || e->opt.big_wg_per_cu <= 0) return lds;
    const size_t want = (size_t)(kLdsPerCu / e->opt.big_wg_per_cu - 1024) & ~(size_t)511;
    return (want > lds && want <= 65536) ? want : lds;
}

static int launch_step(sgw_engine* e, Params& p, hipStream_t s, RowPtrs* sweep_rows = nullptr) {
    p.agent_state = e->agent_state;
    p.state_at_pov = e->state_at_pov;
    p.agent_dir = e->agent_dir;
    if (p.agent_rule == SGW_AGENT_RULE_CLEANUP && p.do_move && !p.agent_dir)
        return fail(SGW_EINVAL, "SGW_AGENT_RULE_CLEANUP needs sgw_bind_agent_dir");
    p.obs_u8 = e->obs_format == SGW_OBS_U8 ? 1 : 0;
    if (p.agent_rule == SGW_AGENT_RULE_TAG && p.do_move && !p.agent_state)
        return fail(SGW_EINVAL, "SGW_AGENT_RULE_TAG needs sgw_bind_agent_state");
    if (int rc = time_begin(e, s)) return rc;
    p.env_lds = e->step_env_lds;
    if (e->fast || e->big) p.tab_bytes = e->big ? e->big_tab_bytes : e->fast_tab_bytes;
    p.obs_stage = (e->fast && p.obs && (reinterpret_cast<uintptr_t>(p.obs) & 15) == 0) ? e->obs_stage : 0;
    if (sweep_rows) p.obs_stage = e->obs_stage;   // (step_fast_rows: the staged windows leave per agent, whatever `obs` is)
    if (p.spawn_mask == 0 && !p.has_become) p.flags &= ~SGW_STEP_SWEEP;   // nothing transitions
    int cap = 0;
    size_t lds = step_lds_request(e, p, &cap);
    // step_big: the walking variant keeps the direct stores (measured faster there), and so does a launch whose observation
    // pointer is not 16-byte aligned; such a launch does not ask for the staging area either
    const bool walk = e->big && p.nturns == 1 && !sweep_rows && e->k_walk.usable() && p.E > e->walk_min_envs && p.E <= e->walk_max_envs;
    p.big_stage = (e->big && ((p.obs && (reinterpret_cast<uintptr_t>(p.obs) & 15) == 0) || (sweep_rows && !p.obs_u8)) && !walk && p.E > e->big_stage_min_envs) ? e->big_stage : 0;
    if (walk) {     // the walking workgroups: a static share each, the rest off a counter (step_big.h)
        p.walk_ctr = reinterpret_cast<uint32_t*>(e->d_status) + 1;
        p.walk_static = e->opt.big_walk_share > 0 ? e->opt.big_walk_share : (int)std::max<int64_t>(1, p.E / e->walk_blocks);
    }
    if (e->big && p.nturns > 1 && e->big_threads != kBigThreads) p.big_stage = 0;   // (the rollout instance runs kBigThreads: the staging area is sized for this engine's waves)
    if (e->big && e->big_stage && !p.big_stage) lds -= (size_t)(e->big_threads / 64) * e->big_stage;
    if (walk) { p.walk_word = (int)lds; lds += 16; }   // (behind the grid image: the walking variant has no staging area there)
    lds = big_cap_lds(e, lds);
    // A policy-driven phase (at most one agent moves, at most one window is rendered, no sweep, plain moves) of a one-hot
    // world whose (layers, channels, radius) has a phase_rows instance: a lane per window row, no staging, any world size.
    // (the phase kernels take the acting agent's action from the tensor: a phase whose action is drawn on the device -- SGW_STEP_RANDOM_ACTIONS,
    // an agent with a RandomModel among agents that step one by one -- stays on the step kernel, which draws it)
    if (sweep_rows && e->fast) {
        if (p.obs_stage <= 0 || p.a0 != 0 || p.a1 != p.A || (p.flags & SGW_STEP_NO_OBS))      // (step_fast_rows has no other way to emit than its staged burst)
            return fail(SGW_EINVAL, "sgw_sweep_observe_rows: this engine does not stage its windows");
        if (e->sweep_rows_chunked) p.stage_agents = 1;                                           // (ROWX: a chunk = one agent = one row)
        Kernel& kr = (p.tail_kind != SGW_TAIL_NONE && !e->sweep_rows_chunked) ? e->k_sweep_rows_tail : e->k_sweep_rows;
        if (int rc = launch_kernel(e, kr, (unsigned)e->grid_blocks, kBlock, lds, s, p, sweep_rows)) return rc;
        return time_end(e, s);
    }
    const bool one_phase = p.nturns == 1 && !(p.flags & (SGW_STEP_SWEEP | SGW_STEP_RANDOM_ACTIONS)) && p.a1 - p.a0 <= 1 && (p.do_move || p.a1 - p.a0 == 1);
    if (e->k_rows.usable() && one_phase && !p.obs_u8) {
        // one window per env: contiguous for all envs only in the packed destination ([E][C][V][V])
        const int64_t N = (int64_t)p.C * p.VV;
        const uintptr_t dst = reinterpret_cast<uintptr_t>(p.obs);
        p.rows_mode = (p.obs_A == 1 && (dst & 15) == 0) ? kRowsFlat : kRowsRun;
        (void)N;
        if (int rc = launch_kernel(e, e->k_rows, (unsigned)ceil_div(p.E, e->rows_epb), kBlock, e->rows_lds, s, p, nullptr)) return rc;
        return time_end(e, s);
    }
    // ... otherwise, for worlds above 4 KiB: the byte-gather phase kernel (option phase_kernel)
    if (e->phase_ok && one_phase) {
        Params q = p;
        q.env_lds = (e->onehot ? 4 * SGW_MAX_TYPES * 4 : SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8) + SGW_MAX_TYPES * 8;   // + the value table
        hipLaunchKernelGGL(e->onehot ? phase_kernel<true> : phase_kernel<false>, dim3((unsigned)ceil_div(p.E, 4)), dim3(kBlock),
                           (size_t)4 * q.env_lds, s, q);
        HIP_TRY(hipGetLastError());
        return time_end(e, s);
    }
    // a STAGE kernel has no direct-store path: calls it cannot serve take the plain variant
    Kernel* k = &e->k_step;
    if (e->fast && e->stage_agents > 0 && e->k_plain.usable() &&
        (p.obs_stage == 0 || p.a0 != 0 || p.a1 != p.A || p.obs_next || (p.flags & SGW_STEP_NO_OBS))) {
        k = &e->k_plain;
        p.obs_stage = 0;   // (a compile-time-shape plain instance would otherwise stage a whole env into a chunk-sized area)
        if (e->rgb16) {    // the float64 kernel: its own table area, no staging, no result table
            p.tab_bytes = e->plain_tab_bytes;
            p.env_lds = e->plain_tab_bytes + p.cells_pad;
            lds = (size_t)(kBlock / kWave) * p.env_lds;
        }
    }
    if (p.nturns > 1) k = &e->k_multi;   // sgw_rollout made sure it exists and the call qualifies
    int blocks = e->grid_blocks;
    if (walk) {   // two to three rounds of the plain kernel
        k = &e->k_walk;
        blocks = e->walk_blocks;
    }
    if (sweep_rows) k = &e->k_sweep_rows;     // (step_big<..., ROWS> / step_kernel<..., ROWS>: the plain single-turn variant with the row pointers as its second argument; `walk` is off above)
    if (int rc = launch_kernel(e, *k, (unsigned)blocks, e->big ? (p.nturns > 1 ? kBigThreads : e->big_threads) : kBlock, lds, s, p, sweep_rows)) return rc;
    return time_end(e, s);
}

int sgw_observe(sgw_engine* e, const uint8_t* grid, const uint8_t* agent_pos, float* obs, int32_t agent_begin,
                int32_t agent_end, void* stream) {
    if (!e || !grid || !agent_pos || !obs) return fail(SGW_EINVAL, "sgw_observe: NULL argument");
    if (agent_begin < 0 || agent_end > e->cfg.num_agents || agent_begin > agent_end)
        return fail(SGW_EINVAL, "sgw_observe: agent range [%d, %d) invalid", agent_begin, agent_end);
    Params p = e->base;
    p.grid = const_cast<uint8_t*>(grid); p.pos = const_cast<uint8_t*>(agent_pos); p.obs = obs;
    p.a0 = agent_begin; p.a1 = agent_end; p.flags = 0; p.do_move = 0;
    return launch_step(e, p, static_cast<hipStream_t>(stream));
}

int sgw_observe_full(sgw_engine* e, const uint8_t* grid, void* out, void* stream) {
    if (!e || !grid || !out) return fail(SGW_EINVAL, "sgw_observe_full: NULL argument");
    Params p = e->base;
    p.grid = const_cast<uint8_t*>(grid);
    p.obs_u8 = e->obs_format == SGW_OBS_U8 ? 1 : 0;
    const int64_t n = p.E * (int64_t)p.H * p.W;
    const int blocks = (int)std::min<int64_t>(ceil_div(n, kBlock), (int64_t)e->num_cus * 16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = time_begin(e, s)) return rc;
    hipLaunchKernelGGL(observe_full_kernel, dim3(blocks), dim3(kBlock), 0, s, p, out);
    HIP_TRY(hipGetLastError());
    return time_end(e, s);
}

int sgw_step(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* obs, float* rewards,
             double* total_reward, uint32_t epoch, uint32_t turn, int32_t agent_begin, int32_t agent_end,
             uint32_t flags, void* stream) {
    if (!e || !grid || !agent_pos || !actions || !rewards || !total_reward)
        return fail(SGW_EINVAL, "sgw_step: NULL argument");
    if (!obs && (flags & SGW_STEP_OBS_NEXT)) return fail(SGW_EINVAL, "sgw_step: SGW_STEP_OBS_NEXT needs obs");
    if (!obs && !(flags & SGW_STEP_NO_OBS)) return fail(SGW_EINVAL, "sgw_step: obs is NULL without SGW_STEP_NO_OBS");
    if (agent_begin < 0 || agent_end > e->cfg.num_agents || agent_begin > agent_end)
        return fail(SGW_EINVAL, "sgw_step: agent range [%d, %d) invalid", agent_begin, agent_end);
    if (epoch >= (1u << 28)) return fail(SGW_EINVAL, "epoch must be < 2^28");
    Params p = e->base;
    p.grid = grid; p.pos = agent_pos; p.actions = actions; p.obs = obs; p.rewards = rewards; p.total = total_reward;
    p.epoch = epoch; p.turn = turn; p.a0 = agent_begin; p.a1 = agent_end; p.flags = flags; p.do_move = 1;
    if (flags & SGW_STEP_OBS_AGENT_MAJOR) {
        if (!e->big || agent_begin != 0 || agent_end != e->cfg.num_agents || (flags & (SGW_STEP_OBS_NEXT | SGW_STEP_NO_OBS)) || !obs)
            return fail(SGW_EINVAL, "sgw_step: SGW_STEP_OBS_AGENT_MAJOR is for whole-turn calls with observations on engines with SGW_CAP_OBS_AGENT_MAJOR");
        p.obs_ag = (int64_t)e->cfg.num_envs * e->base.C * e->base.VV;
        p.flags &= ~SGW_STEP_OBS_AGENT_MAJOR;
    }
    if (flags & SGW_STEP_NO_MOVE) {    // sweep + windows, nobody acts
        if (flags & (SGW_STEP_RANDOM_ACTIONS | SGW_STEP_OBS_NEXT | SGW_STEP_OBS_NEXT_PACKED))
            return fail(SGW_EINVAL, "sgw_step: SGW_STEP_NO_MOVE does not combine with RANDOM_ACTIONS / OBS_NEXT");
        p.do_move = 0;
        p.flags &= ~SGW_STEP_NO_MOVE;
        return launch_step(e, p, static_cast<hipStream_t>(stream));
    }
    if (flags & SGW_STEP_OBS_NEXT) {   // the stepped agents' own observations are not written
        p.obs_next = 1;
        p.flags |= SGW_STEP_NO_OBS;
        if (flags & SGW_STEP_OBS_NEXT_PACKED) {   // `obs` holds one window per env: agent_end's
            p.obs_A = 1;
            p.obs_a0 = agent_end;
        }
    } else if (flags & SGW_STEP_OBS_NEXT_PACKED) {
        return fail(SGW_EINVAL, "sgw_step: SGW_STEP_OBS_NEXT_PACKED qualifies SGW_STEP_OBS_NEXT");
    }
    if (int rc = launch_step(e, p, static_cast<hipStream_t>(stream))) return rc;
    if (e->auto_max_turns && turn == e->auto_max_turns && agent_end == e->cfg.num_agents) {
        // end of the epoch: keep the returns, then create_world + populate_environment for the next one
        if (epoch + 1 >= (1u << 28)) return fail(SGW_EINVAL, "epoch must be < 2^28");
        if (e->episode_return)
            HIP_TRY(hipMemcpyAsync(e->episode_return, total_reward, sizeof(double) * (size_t)e->cfg.num_envs,
                                   hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
        return launch_reset(e, grid, agent_pos, total_reward, epoch + 1, stream);
    }
    return SGW_OK;
}

int sgw_rollout(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* obs, float* rewards,
                double* total_reward, uint32_t epoch, uint32_t first_turn, uint32_t num_turns, int64_t obs_turn_stride,
                int64_t actions_turn_stride, int64_t rewards_turn_stride, uint32_t flags, void* stream) {
    if (!e || !grid || !agent_pos || !actions || !rewards || !total_reward)
        return fail(SGW_EINVAL, "sgw_rollout: NULL argument");
    if (!obs && !(flags & SGW_STEP_NO_OBS)) return fail(SGW_EINVAL, "sgw_rollout: obs is NULL without SGW_STEP_NO_OBS");
    if (flags & SGW_STEP_OBS_NEXT) return fail(SGW_EINVAL, "sgw_rollout: SGW_STEP_OBS_NEXT is a per-agent flag of sgw_step");
    if (obs_turn_stride < 0 || actions_turn_stride < 0 || rewards_turn_stride < 0)
        return fail(SGW_EINVAL, "sgw_rollout: negative turn stride");
    const int A = e->cfg.num_agents;
    if (e->multi_turn && !e->k_multi.jit && !e->k_multi.host) {   // the turn-loop instance exists only specialised: get it now, or loop over single turns
        if (resolve_kernel(e, e->k_multi) != SGW_OK) e->multi_turn = false;
    }
    uint32_t done = 0;
    while (done < num_turns) {
        const uint32_t turn = first_turn + done;
        // turns of one launch: up to the end of the epoch if auto-reset is armed; one per launch on kernels without a turn loop
        uint32_t n = num_turns - done;
        if (e->auto_max_turns && turn <= e->auto_max_turns) n = std::min(n, e->auto_max_turns - turn + 1);
        if (!e->multi_turn) n = 1;
        // the wave-per-env MULTI kernels stage their observations: every turn's slot must keep the 16-byte alignment
        if (e->fast && (!obs || (flags & SGW_STEP_NO_OBS) || (reinterpret_cast<uintptr_t>(obs) & 15) || (obs_turn_stride & 3) || e->obs_stage == 0)) n = 1;
        if (epoch >= (1u << 28)) return fail(SGW_EINVAL, "epoch must be < 2^28");
        Params p = e->base;
        p.grid = grid; p.pos = agent_pos; p.total = total_reward;
        p.actions = actions + (int64_t)done * actions_turn_stride;
        p.obs = obs ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(obs) + (int64_t)done * obs_turn_stride * (e->obs_format == SGW_OBS_U8 ? 1 : 4)) : nullptr;
        p.rewards = rewards + (int64_t)done * rewards_turn_stride;
        p.epoch = epoch; p.turn = turn; p.a0 = 0; p.a1 = A; p.flags = flags; p.do_move = 1;
        p.nturns = n; p.ts_obs = obs_turn_stride; p.ts_act = actions_turn_stride; p.ts_rew = rewards_turn_stride;
        if (int rc = launch_step(e, p, static_cast<hipStream_t>(stream))) return rc;
        done += n;
        if (e->auto_max_turns && first_turn + done - 1 == e->auto_max_turns) {
            // end of the epoch: keep the returns, reset for the next one; the caller's turn counter restarts at 1
            if (epoch + 1 >= (1u << 28)) return fail(SGW_EINVAL, "epoch must be < 2^28");
            if (e->episode_return)
                HIP_TRY(hipMemcpyAsync(e->episode_return, total_reward, sizeof(double) * (size_t)e->cfg.num_envs,
                                       hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
            if (int rc = launch_reset(e, grid, agent_pos, total_reward, epoch + 1, stream)) return rc;
            epoch += 1;
            first_turn = 1 - done;     // turn = first_turn + done continues at 1 (unsigned wrap-around is intended)
        }
    }
    return SGW_OK;
}

int sgw_capabilities(sgw_engine* e) {
    if (!e) return 0;
    int caps = 0;
    if (e->k_obs_rows.usable() && e->obs_format == SGW_OBS_F32) caps |= SGW_CAP_OBSERVE_ROWS;
    caps |= SGW_CAP_ACT;      // MovingAgent.act, TagAgent.act and CleanupAgent.act all have an sgw_act instance
    {   // sgw_turn_resolve: plain movers with impassable agent types, float32 windows
        bool ok = e->cfg.agent_rule == SGW_AGENT_RULE_MOVE && e->obs_format == SGW_OBS_F32 && e->cfg.num_agents <= 64;   // (the resolve kernel keeps an agent per lane)
        for (int a = 0; a < e->cfg.num_agents; ++a) ok = ok && !e->cfg.type_passable[e->cfg.agent_type[a]];
        if (ok) caps |= SGW_CAP_RESOLVE;
    }
    if (e->big) caps |= SGW_CAP_OBS_AGENT_MAJOR;
    // (round 6: also step_big<..., ROWS> and the chunk-staging twins, which write the bound row tail themselves; the whole-env instance has a TAIL twin)
    if (e->k_sweep_rows.usable() && e->obs_format == SGW_OBS_F32 &&
        (e->tail_kind == SGW_TAIL_NONE || !e->fast || e->sweep_rows_chunked || e->k_sweep_rows_tail.jit)) caps |= SGW_CAP_SWEEP_ROWS;
    return caps;
}

static int fill_rows(const sgw_engine* e, void* const* rows, int64_t env_stride, int a0, int a1, bool need_all, RowPtrs* rp, const char* who) {
    const sgw_config& c = e->cfg;
    const int64_t V = 2 * c.vision_radius + 1;
    if (!rows) return fail(SGW_EINVAL, "%s: rows is NULL", who);
    if (env_stride < (int64_t)c.num_channels * V * V) return fail(SGW_EINVAL, "%s: env_stride is smaller than one window", who);
    if (e->tail_kind != SGW_TAIL_NONE && env_stride < (int64_t)c.num_channels * V * V + e->tail_len)
        return fail(SGW_EINVAL, "%s: env_stride is smaller than one window + the bound row tail (%d elements)", who, e->tail_len);
    memset(rp, 0, sizeof(*rp));
    for (int a = a0; a < a1; ++a) {
        if (!rows[a] && need_all) return fail(SGW_EINVAL, "%s: rows[%d] is NULL", who, a);
        if (e->obs_format == SGW_OBS_F32 && (reinterpret_cast<uintptr_t>(rows[a]) & 3u))
            return fail(SGW_EINVAL, "%s: rows[%d] is not 4-byte aligned", who, a);
        rp->p[a] = rows[a];
    }
    rp->stride = env_stride;
    return SGW_OK;
}

static int observe_rows_impl(sgw_engine* e, const uint8_t* grid, const uint8_t* agent_pos, void* const* rows, int64_t env_stride,
                             int32_t agent_begin, int32_t agent_end, const TurnState* ts, void* stream) {
    if (!e || !grid || !agent_pos) return fail(SGW_EINVAL, "sgw_observe_rows: NULL argument");
    if (agent_begin < 0 || agent_end > e->cfg.num_agents || agent_begin >= agent_end)
        return fail(SGW_EINVAL, "sgw_observe_rows: agent range [%d, %d) invalid", agent_begin, agent_end);
    if (!(sgw_capabilities(e) & SGW_CAP_OBSERVE_ROWS))
        return fail(SGW_EINVAL, "sgw_observe_rows: no row-load instance for this world (one-hot float32 windows of plain movers only; "
                                "see sgw_capabilities) -- use sgw_observe");
    RowPtrs rp;
    if (int rc = fill_rows(e, rows, env_stride, agent_begin, agent_end, true, &rp, "sgw_observe_rows")) return rc;
    if (ts && e->turn_rows) {   // a recorded turn: the windows also go to the replay rows of the turn in flight
        rp.ts = ts;
        rp.dual = 1;
        rp.rows_mode2 = e->turn_rows_flat ? kRowsFlat : kRowsRun;
    }
    Params p = e->base;
    p.grid = const_cast<uint8_t*>(grid); p.pos = const_cast<uint8_t*>(agent_pos);
    p.a0 = agent_begin; p.a1 = agent_end; p.flags = 0; p.do_move = 0;
    p.agent_state = e->agent_state;
    p.tail_kind = e->tail_kind; p.tail_len = e->tail_len; p.tail_table = e->tail_table;
    if (p.tail_kind == SGW_TAIL_AGENT_IS_IT && !p.agent_state) return fail(SGW_EINVAL, "SGW_TAIL_AGENT_IS_IT needs sgw_bind_agent_state");
    {   // how the windows leave (phase.h, rows_emit)
        const int64_t N = (int64_t)p.C * p.VV;
        const int nA = agent_end - agent_begin;
        bool al16 = true, al8 = true, slots = env_stride == (int64_t)nA * N;     // slots: rows[a] = rows[a0] + (a - a0) * N, i.e. one [E][nA][N] tensor
        for (int a = agent_begin; a < agent_end; ++a) {
            const uintptr_t q = reinterpret_cast<uintptr_t>(rows[a]);
            al16 = al16 && (q & 15) == 0;
            al8 = al8 && (q & 7) == 0;
            slots = slots && q == reinterpret_cast<uintptr_t>(rows[agent_begin]) + (uintptr_t)(a - agent_begin) * N * 4;
        }
        p.rows_by_agent = 0;
        const bool pair_ok = al8 && (N & 1) == 0 && (env_stride & 1) == 0;
        if (slots && (reinterpret_cast<uintptr_t>(rows[agent_begin]) & 15) == 0) p.rows_mode = kRowsFlat;
        else if (env_stride == N && al16) { p.rows_mode = kRowsFlat; p.rows_by_agent = 1; }
        else p.rows_mode = kRowsRun;      // per window: aligned float4 runs, the ends element by element
        const int m = e->opt.rows_mode;   // test hook: force another emit (1 = single floats, 2 = float2 runs where legal, 3 = aligned runs)
        if (m == kRowsSingle || m == kRowsRun || (m == kRowsPair && pair_ok)) { p.rows_mode = m; p.rows_by_agent = 0; }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = time_begin(e, s)) return rc;
    const int wpw = e->rows_wpb / 4;                                   // windows per wave
    const int64_t waves = p.rows_by_agent ? ceil_div(p.E, wpw) * (agent_end - agent_begin) : ceil_div(p.E * (agent_end - agent_begin), wpw);
    if (rp.dual && rp.rows_mode2 == kRowsFlat && !p.rows_by_agent) rp.rows_mode2 = kRowsRun;   // (flat second copies need a wave = consecutive envs of ONE agent)
    if (int rc = launch_kernel(e, e->k_obs_rows, (unsigned)ceil_div(waves, 4), kBlock, e->rows_lds, s, p, &rp)) return rc;
    return time_end(e, s);
}

int sgw_observe_rows(sgw_engine* e, const uint8_t* grid, const uint8_t* agent_pos, void* const* rows, int64_t env_stride,
                     int32_t agent_begin, int32_t agent_end, void* stream) {
    return observe_rows_impl(e, grid, agent_pos, rows, env_stride, agent_begin, agent_end, nullptr, stream);
}

int sgw_sweep_observe_rows(sgw_engine* e, uint8_t* grid, const uint8_t* agent_pos, void* const* rows, int64_t env_stride, uint32_t epoch,
                           uint32_t turn, uint32_t flags, void* stream) {
    if (!e || !grid || !agent_pos) return fail(SGW_EINVAL, "sgw_sweep_observe_rows: NULL argument");
    if (!(sgw_capabilities(e) & SGW_CAP_SWEEP_ROWS))
        return fail(SGW_EINVAL, "sgw_sweep_observe_rows: no fused instance for this world (see sgw_capabilities: SGW_CAP_SWEEP_ROWS) -- "
                                "sgw_step(sweep only) + sgw_observe_rows do the same in two launches");
    if (flags & ~(uint32_t)SGW_STEP_SWEEP) return fail(SGW_EINVAL, "sgw_sweep_observe_rows: flags may hold SGW_STEP_SWEEP only");
    if (epoch >= (1u << 28)) return fail(SGW_EINVAL, "epoch must be < 2^28");
    const int A = e->cfg.num_agents;
    if (env_stride != (int64_t)e->base.C * e->base.VV + e->tail_len)
        return fail(SGW_EINVAL, "sgw_sweep_observe_rows: env_stride must be exactly one window (C * V * V elements) + the bound row tail (%d)", e->tail_len);
    RowPtrs rp;
    if (int rc = fill_rows(e, rows, env_stride, 0, A, true, &rp, "sgw_sweep_observe_rows")) return rc;
    Params p = e->base;
    p.grid = grid; p.pos = const_cast<uint8_t*>(agent_pos);
    p.actions = nullptr; p.obs = nullptr; p.rewards = nullptr; p.total = nullptr;
    p.epoch = epoch; p.turn = turn; p.a0 = 0; p.a1 = A; p.flags = flags & SGW_STEP_SWEEP; p.do_move = 0;
    p.tail_kind = e->tail_kind; p.tail_len = e->tail_len; p.tail_table = e->tail_table;
    if (p.tail_kind == SGW_TAIL_AGENT_IS_IT && !e->agent_state) return fail(SGW_EINVAL, "SGW_TAIL_AGENT_IS_IT needs sgw_bind_agent_state");
    return launch_step(e, p, static_cast<hipStream_t>(stream), &rp);
}

static int act_impl(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, void* const* rows, int64_t env_stride,
                    float* rewards, double* total_reward, int32_t agent, const void* agent_action, int32_t action_kind,
                    float* reward_row, int64_t* action_row, const TurnState* ts, void* stream, int dual = 0) {
    if (!e || !grid || !agent_pos || !actions || !rewards || !total_reward) return fail(SGW_EINVAL, "sgw_act: NULL argument");
    if (agent < 0 || agent >= e->cfg.num_agents) return fail(SGW_EINVAL, "sgw_act: agent %d out of range", agent);
    if (e->cfg.agent_rule == SGW_AGENT_RULE_TAG && !e->agent_state) return fail(SGW_EINVAL, "SGW_AGENT_RULE_TAG needs sgw_bind_agent_state");
    if (e->cfg.agent_rule == SGW_AGENT_RULE_CLEANUP && !e->agent_dir) return fail(SGW_EINVAL, "SGW_AGENT_RULE_CLEANUP needs sgw_bind_agent_dir");
    RowPtrs rp;
    memset(&rp, 0, sizeof(rp));
    if (rows)
        if (int rc = fill_rows(e, rows, env_stride, agent + 1, e->cfg.num_agents, false, &rp, "sgw_act")) return rc;
    if (agent_action && action_kind != SGW_ACT_U8 && action_kind != SGW_ACT_I32 && action_kind != SGW_ACT_I64 && action_kind != SGW_ACT_QF32)
        return fail(SGW_EINVAL, "sgw_act: unknown action_kind %d", action_kind);
    if (agent_action && action_kind == SGW_ACT_QF32 && (reinterpret_cast<uintptr_t>(agent_action) & 3))
        return fail(SGW_EINVAL, "sgw_act: SGW_ACT_QF32 action values must be 4-byte aligned");
    rp.agent_action = agent_action; rp.action_kind = action_kind; rp.reward_row = reward_row; rp.action_row = action_row;
    rp.ts = ts;
    rp.ets = (agent_action && action_kind == SGW_ACT_QF32) ? e->d_turn : nullptr;
    rp.dual = (dual && ts && e->turn_rows) ? 1 : 0;
    Params p = e->base;
    p.grid = grid; p.pos = agent_pos; p.actions = actions; p.rewards = rewards; p.total = total_reward;
    p.a0 = agent; p.a1 = agent + 1; p.flags = SGW_STEP_NO_OBS; p.do_move = 1;
    p.agent_state = e->agent_state;
    p.state_at_pov = e->state_at_pov;
    p.agent_dir = e->agent_dir;
    p.obs_u8 = e->obs_format == SGW_OBS_U8 ? 1 : 0;
    p.tail_kind = rows ? e->tail_kind : SGW_TAIL_NONE; p.tail_len = e->tail_len; p.tail_table = e->tail_table;
    if (p.tail_kind != SGW_TAIL_NONE && env_stride < (int64_t)e->base.C * e->base.VV + e->tail_len) p.tail_kind = SGW_TAIL_NONE;   // (windows in the observation tensor: no room for tails)
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = time_begin(e, s)) return rc;
    const int A = e->cfg.num_agents;
    const bool wide = A > 8 && A <= 16 && (e->opt.act_lanes == 16 || (e->opt.act_lanes == 0 && p.E <= 32768));   // (two agents per lane pay off once the waves outnumber the SIMDs' slots: profiles/r04_act_probe.txt)
    const int G = A <= 16 ? (wide ? 16 : 8) : (A <= 32 ? 32 : 64);      // lanes per env; 9..16 agents: two per lane
    const unsigned blocks = (unsigned)ceil_div(p.E, 4 * (64 / G));
#define ACT_PICK(R, OH) (A <= 8 ? act_patch<8, 1, R, OH> : (A <= 16 ? (wide ? act_patch<16, 1, R, OH> : act_patch<8, 2, R, OH>) : (A <= 32 ? act_patch<32, 1, R, OH> : (A <= 64 ? act_patch<64, 1, R, OH> : act_patch<64, 2, R, OH>))))
#define ACT_RULE(OH) (e->cfg.agent_rule == SGW_AGENT_RULE_TAG ? ACT_PICK(SGW_AGENT_RULE_TAG, OH) \
                      : e->cfg.agent_rule == SGW_AGENT_RULE_CLEANUP ? ACT_PICK(SGW_AGENT_RULE_CLEANUP, OH) : ACT_PICK(SGW_AGENT_RULE_MOVE, OH))
    RowsFn fn = e->onehot ? ACT_RULE(true) : ACT_RULE(false);
#undef ACT_RULE
#undef ACT_PICK
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(kBlock), 0, s, p, rp);
    HIP_TRY(hipGetLastError());
    return time_end(e, s);
}

int sgw_act(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, void* const* rows, int64_t env_stride,
            float* rewards, double* total_reward, int32_t agent, const void* agent_action, int32_t action_kind,
            float* reward_row, int64_t* action_row, void* stream) {
    return act_impl(e, grid, agent_pos, actions, rows, env_stride, rewards, total_reward, agent, agent_action, action_kind, reward_row,
                    action_row, nullptr, stream);
}

// ---- a whole policy turn as ONE submission (include/sgw.h: sgw_turn_*)
int sgw_turn_bind(sgw_engine* e, const sgw_turn_rows* rows) {
    if (!e) return fail(SGW_EINVAL, "sgw_turn_bind: NULL engine");
    // Only the ring fields (row ... row_elems: one contiguous range of TurnState) are written; epoch, turn and the exploration
    // thresholds are the stream-ordered kernels' (sgw_turn_set / sgw_turn_epsilon / sgw_turn_end) and are never read back or rewritten
    // here.  The call blocks: everything submitted before it -- on ANY stream, a recorded turn's side stream included -- has finished
    // when the rings change (until round 5 a read-modify-write of the whole struct on the null stream could overtake kernels in
    // flight on a non-blocking stream and write stale counters back).
    TurnState h;
    memset(&h, 0, sizeof(h));
    HIP_TRY(hipDeviceSynchronize());
    const int A = e->cfg.num_agents;
    const int64_t N = (int64_t)e->base.C * e->base.VV;
    for (int a = 0; a < SGW_MAX_AGENTS; ++a) {
        h.row[a] = h.cap[a] = h.step[a] = h.row_elems[a] = 0;
        h.states[a] = nullptr; h.rewards[a] = nullptr; h.actions[a] = nullptr; h.dones[a] = nullptr;
    }
    e->turn_rows = false;
    for (int a = 0; a < SGW_MAX_AGENTS; ++a) { e->turn_cap[a] = 0; e->turn_row_bytes[a] = 0; e->turn_states[a] = nullptr; }
    if (rows) {
        for (int a = 0; a < A; ++a) {
            if (rows->capacity[a] <= 0) continue;
            if (rows->row[a] < 0 || rows->row[a] >= rows->capacity[a] || rows->step[a] < 1)
                return fail(SGW_EINVAL, "sgw_turn_bind: agent %d: row must be in [0, capacity) and step >= 1", a);
            if (rows->states[a] && rows->row_elems[a] < N) return fail(SGW_EINVAL, "sgw_turn_bind: agent %d: row_elems is smaller than one window", a);
            const int esz = e->obs_format == SGW_OBS_U8 ? 1 : 4;
            if (rows->states[a] && (reinterpret_cast<uintptr_t>(rows->states[a]) % esz)) return fail(SGW_EINVAL, "sgw_turn_bind: agent %d: states is misaligned", a);
            h.row[a] = rows->row[a]; h.cap[a] = rows->capacity[a]; h.step[a] = rows->step[a]; h.row_elems[a] = rows->row_elems[a];
            h.states[a] = rows->states[a]; h.rewards[a] = rows->rewards[a]; h.actions[a] = rows->actions[a];
            h.dones[a] = rows->states[a] ? rows->dones[a] : nullptr;   // (zeroed by the window copy)
            if (rows->states[a]) {
                e->turn_cap[a] = rows->capacity[a];
                e->turn_row_bytes[a] = (int64_t)e->cfg.num_envs * rows->row_elems[a] * esz;
                e->turn_states[a] = rows->states[a];
            }
            if (!e->turn_rows) e->turn_rows_even = e->turn_rows_flat = true;
            e->turn_rows = true;
            if (rows->states[a] && ((reinterpret_cast<uintptr_t>(rows->states[a]) & 15) || rows->row_elems[a] != N || (((int64_t)e->cfg.num_envs * N * 4) & 15)))
                e->turn_rows_flat = false;
            if (rows->states[a] && ((reinterpret_cast<uintptr_t>(rows->states[a]) & 7) || (rows->row_elems[a] & 1))) e->turn_rows_even = false;
        }
    }
    constexpr size_t r0 = offsetof(TurnState, row), r1 = offsetof(TurnState, eps_thr);
    HIP_TRY(hipMemcpy(reinterpret_cast<char*>(e->d_turn) + r0, reinterpret_cast<const char*>(&h) + r0, r1 - r0, hipMemcpyHostToDevice));
    return SGW_OK;
}

int sgw_turn_set(sgw_engine* e, uint32_t epoch, uint32_t turn, void* stream) {
    if (!e) return fail(SGW_EINVAL, "sgw_turn_set: NULL engine");
    if (epoch >= (1u << 28)) return fail(SGW_EINVAL, "epoch must be < 2^28");
    hipLaunchKernelGGL(turn_set_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), e->d_turn, epoch, turn);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_turn_epsilon(sgw_engine* e, int32_t agent, double epsilon, void* stream) {
    if (!e) return fail(SGW_EINVAL, "sgw_turn_epsilon: NULL engine");
    if (agent < -1 || agent >= e->cfg.num_agents) return fail(SGW_EINVAL, "sgw_turn_epsilon: agent %d out of range", agent);
    if (!(epsilon >= 0.0 && epsilon <= 1.0)) return fail(SGW_EINVAL, "sgw_turn_epsilon: epsilon must be in [0, 1]");
    const double t = std::floor(epsilon * 4294967296.0);
    const uint64_t thr = t >= 4294967296.0 ? 4294967296ull : (uint64_t)t;
    hipLaunchKernelGGL(turn_epsilon_kernel, dim3(1), dim3(SGW_MAX_AGENTS), 0, static_cast<hipStream_t>(stream), e->d_turn, agent, e->cfg.num_agents, thr);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_turn_begin(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* obs, float* rewards, double* total_reward,
                   uint32_t flags, void* stream) {
    if (!e || !grid || !agent_pos || !actions || !obs || !rewards || !total_reward) return fail(SGW_EINVAL, "sgw_turn_begin: NULL argument");
    if (flags & ~(SGW_STEP_SWEEP)) return fail(SGW_EINVAL, "sgw_turn_begin: only SGW_STEP_SWEEP may be set");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Params p = e->base;
    p.grid = grid; p.pos = agent_pos; p.actions = actions; p.obs = obs; p.rewards = rewards; p.total = total_reward;
    p.ts = e->d_turn;      // the turn in flight = the device's count of completed turns + 1 (Environment.take_turn: self.turn += 1)
    p.a0 = 0; p.a1 = e->cfg.num_agents; p.flags = flags; p.do_move = 0;
    return launch_step(e, p, s);
}

int sgw_turn_act(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, void* obs, float* rewards, double* total_reward,
                 int32_t agent, const void* agent_action, int32_t action_kind, void* stream) {
    if (!e || !obs) return fail(SGW_EINVAL, "sgw_turn_act: NULL argument");
    const int A = e->cfg.num_agents;
    const int64_t N = (int64_t)e->base.C * e->base.VV;
    const int esz = e->obs_format == SGW_OBS_U8 ? 1 : 4;
    void* rows[SGW_MAX_AGENTS];
    for (int a = 0; a < A; ++a) rows[a] = static_cast<uint8_t*>(obs) + (int64_t)a * N * esz;   // slot a of the [E][A][C][V][V] tensor
    return act_impl(e, grid, agent_pos, actions, rows, (int64_t)A * N, rewards, total_reward, agent, agent_action, action_kind, nullptr, nullptr,
                    e->d_turn, stream);
}

int sgw_turn_begin_rows(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* rewards, double* total_reward,
                        void* const* rows, int64_t env_stride, uint32_t flags, void* stream) {
    if (!e || !grid || !agent_pos || !actions || !rewards || !total_reward || !rows) return fail(SGW_EINVAL, "sgw_turn_begin_rows: NULL argument");
    if (flags & ~(SGW_STEP_SWEEP)) return fail(SGW_EINVAL, "sgw_turn_begin_rows: only SGW_STEP_SWEEP may be set");
    if (!(sgw_capabilities(e) & SGW_CAP_OBSERVE_ROWS)) return fail(SGW_EINVAL, "sgw_turn_begin_rows needs SGW_CAP_OBSERVE_ROWS (one-hot float32 windows); use sgw_turn_begin");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // both steps in one launch -- on the whole-env instance, whose emit also writes the recorded turn's second copy (the ring rows by the device's
    // row count); step_big, the chunk-staging and the generic instances (round 6) render into the rows alone, so a recorded turn keeps the two launches there
    if ((sgw_capabilities(e) & SGW_CAP_SWEEP_ROWS) && e->fast && !e->sweep_rows_chunked && e->tail_kind == SGW_TAIL_NONE &&
        env_stride == (int64_t)e->base.C * e->base.VV) {
        RowPtrs rp;
        if (int rc = fill_rows(e, rows, env_stride, 0, e->cfg.num_agents, true, &rp, "sgw_turn_begin_rows")) return rc;
        rp.ts = e->d_turn;
        rp.dual = e->turn_rows ? 1 : 0;
        Params p = e->base;
        p.grid = grid; p.pos = agent_pos; p.actions = nullptr; p.obs = nullptr; p.rewards = nullptr; p.total = nullptr;
        p.ts = e->d_turn;
        p.a0 = 0; p.a1 = e->cfg.num_agents; p.flags = flags & SGW_STEP_SWEEP; p.do_move = 0;
        return launch_step(e, p, s, &rp);
    }
    if (flags & SGW_STEP_SWEEP) {     // the entity sweep alone, at the device's turn
        Params p = e->base;
        p.grid = grid; p.pos = agent_pos; p.actions = actions; p.obs = nullptr; p.rewards = rewards; p.total = total_reward;
        p.ts = e->d_turn;
        p.a0 = 0; p.a1 = 0; p.flags = SGW_STEP_SWEEP | SGW_STEP_NO_OBS; p.do_move = 1;
        if (int rc = launch_step(e, p, s)) return rc;
    }
    return observe_rows_impl(e, grid, agent_pos, rows, env_stride, 0, e->cfg.num_agents, e->d_turn, stream);
}

int sgw_turn_act_rows(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, void* const* rows, int64_t env_stride, float* rewards,
                      double* total_reward, int32_t agent, const void* agent_action, int32_t action_kind, void* stream) {
    if (!e || !rows) return fail(SGW_EINVAL, "sgw_turn_act_rows: NULL argument");
    return act_impl(e, grid, agent_pos, actions, rows, env_stride, rewards, total_reward, agent, agent_action, action_kind, nullptr, nullptr,
                    e->d_turn, stream, 1);
}

int sgw_turn_resolve(sgw_engine* e, uint8_t* grid, uint8_t* agent_pos, uint8_t* actions, float* rows, int64_t row_elems, float* rewards,
                     double* total_reward, uint8_t* scratch, int64_t* dirty_list, uint32_t* counters, const int64_t* new_actions,
                     int64_t n_new, float* reward_rows, int64_t* action_rows, int32_t pass, void* stream) {
    if (!e || !grid || !agent_pos || !actions || !rows || !rewards || !total_reward || !scratch)
        return fail(SGW_EINVAL, "sgw_turn_resolve: NULL argument");
    const sgw_config& c = e->cfg;
    if (c.agent_rule != SGW_AGENT_RULE_MOVE) return fail(SGW_EINVAL, "sgw_turn_resolve: plain movers only (SGW_AGENT_RULE_MOVE)");
    if (c.num_agents > 64) return fail(SGW_EINVAL, "sgw_turn_resolve: at most 64 agents (an agent per lane)");
    if (e->obs_format != SGW_OBS_F32) return fail(SGW_EINVAL, "sgw_turn_resolve: float32 windows only");
    for (int a = 0; a < c.num_agents; ++a)
        if (c.type_passable[c.agent_type[a]]) return fail(SGW_EINVAL, "sgw_turn_resolve: agent types must be impassable");
    if (row_elems < (int64_t)e->base.C * e->base.VV) return fail(SGW_EINVAL, "sgw_turn_resolve: row_elems is smaller than one window");
    if (reinterpret_cast<uintptr_t>(rows) & 3u) return fail(SGW_EINVAL, "sgw_turn_resolve: rows is not 4-byte aligned");
    if (pass < 0) return fail(SGW_EINVAL, "sgw_turn_resolve: pass must be 0 (render the pre-move windows) or the number of the pass, 1 ...");
    const int64_t EA = (int64_t)c.num_envs * c.num_agents;
    if ((dirty_list == nullptr) != (counters == nullptr)) return fail(SGW_EINVAL, "sgw_turn_resolve: dirty_list and counters come together");
    if (new_actions && (n_new < 0 || n_new > EA || (pass == 1 && n_new != EA) || pass == 0))
        return fail(SGW_EINVAL, "sgw_turn_resolve: new_actions holds every row's action in pass 1 (n_new = E * A), the previous pass's dirty rows' afterwards");
    if (new_actions && pass > 1 && !dirty_list) return fail(SGW_EINVAL, "sgw_turn_resolve: new_actions of a later pass follow the dirty list");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = time_begin(e, s)) return rc;
    if (pass == 1 && counters) HIP_TRY(hipMemsetAsync(counters, 0, 8 * sizeof(uint32_t), s));
    if (new_actions && n_new > 0) {      // the policy's output -> actions[env][agent]
        const int64_t* lst = pass == 1 ? nullptr : dirty_list + ((pass - 1) & 1) * EA;
        const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n_new, kBlock), (int64_t)e->num_cus * 8);
        hipLaunchKernelGGL(resolve_apply_actions, dim3(blocks), dim3(kBlock), 0, s, actions, lst, new_actions, n_new, (int64_t)c.num_envs, c.num_agents);
    }
    Params p = e->base;
    p.grid = grid; p.pos = agent_pos; p.actions = actions; p.rewards = rewards; p.total = total_reward;
    p.a0 = 0; p.a1 = c.num_agents; p.do_move = 1; p.flags = 0;
    ResolveArgs ra;
    ra.rows = rows; ra.row_elems = row_elems;
    ra.env_done = scratch; ra.pristine = scratch + EA; ra.dirty = scratch + 2 * EA; ra.prev = scratch + 3 * EA;
    // the dirty list: appended with one atomic per env -- or, from 8 192 envs on (where those atomics on one counter would serialise for
    // hundreds of microseconds), laid out afterwards by a scan over per-env counts
    const bool scan = dirty_list && pass >= 1 && c.num_envs >= 8192;
    const int nb = (int)ceil_div(c.num_envs, 256);
    if (scan && (!e->d_dcount || !e->d_doffsets)) return fail(SGW_EINVAL, "sgw_turn_resolve: the scan buffers of this engine are missing");   // (sgw_create allocates both from 8 192 envs on)
    ra.list = (dirty_list && !scan) ? dirty_list + (pass & 1) * EA : nullptr;
    ra.count = (counters && !scan) ? counters + (pass & 7) : nullptr;
    ra.count_next = counters ? counters + ((pass + 1) & 7) : nullptr;
    ra.dcount = scan ? e->d_dcount : nullptr;
    ra.bsum = scan ? e->d_doffsets : nullptr;
    ra.reward_rows = reward_rows; ra.action_rows = action_rows;
    ra.first = pass == 0 ? 2 : (pass == 1 ? 1 : 0);
    ra.diag = e->opt.resolve_diag;
    // a workgroup per env (four waves share the windows to verify) from 16 agents on, a wave per env below
    const bool wide = c.num_agents >= 16;
    const unsigned blocks = (unsigned)(wide ? p.E : ceil_div(p.E, 4));
    const size_t tab = e->onehot ? (size_t)4 * SGW_MAX_TYPES * 4 : (size_t)SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8;
    const size_t lds = 4 * tab + 64;
    if (e->onehot) {
        if (wide) hipLaunchKernelGGL((turn_resolve<true, 4>), dim3(blocks), dim3(kBlock), lds, s, p, ra);
        else hipLaunchKernelGGL((turn_resolve<true, 1>), dim3(blocks), dim3(kBlock), lds, s, p, ra);
    } else {
        if (wide) hipLaunchKernelGGL((turn_resolve<false, 4>), dim3(blocks), dim3(kBlock), lds, s, p, ra);
        else hipLaunchKernelGGL((turn_resolve<false, 1>), dim3(blocks), dim3(kBlock), lds, s, p, ra);
    }
    if (scan) {
        hipLaunchKernelGGL(resolve_scan_blocks, dim3(1), dim3(256), 0, s, e->d_doffsets, nb, e->d_doffsets + nb, counters + (pass & 7));
        hipLaunchKernelGGL(resolve_fill_list, dim3((unsigned)nb), dim3(256), 0, s, ra.dirty, e->d_dcount, ra.env_done, e->d_doffsets + nb,
                           (int64_t)c.num_envs, c.num_agents, dirty_list + (pass & 1) * EA);
    }
    HIP_TRY(hipGetLastError());
    return time_end(e, s);
}

int sgw_verify_rows(sgw_engine* e, const float* obs, const uint8_t* state_at_pov, float* rows, int64_t row_elems, int64_t* list, uint32_t* count, void* stream) {
    if (!e || !obs || !rows || !list || !count) return fail(SGW_EINVAL, "sgw_verify_rows: NULL argument");
    if (e->obs_format != SGW_OBS_F32) return fail(SGW_EINVAL, "sgw_verify_rows: float32 windows only");
    const int N = e->base.C * e->base.VV;
    const bool tail_it = e->tail_kind == SGW_TAIL_AGENT_IS_IT;
    if (row_elems < N + e->tail_len) return fail(SGW_EINVAL, "sgw_verify_rows: row_elems is smaller than one window + the bound row tail");
    if (tail_it && !state_at_pov) return fail(SGW_EINVAL, "sgw_verify_rows: SGW_TAIL_AGENT_IS_IT needs the scratch turn's state_at_pov");
    if ((reinterpret_cast<uintptr_t>(obs) | reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(count)) & 3u) return fail(SGW_EINVAL, "sgw_verify_rows: misaligned pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
    const int64_t waves = (int64_t)e->cfg.num_envs * e->cfg.num_agents;
    hipLaunchKernelGGL(verify_rows_kernel, dim3((unsigned)ceil_div(waves, kBlock / 64)), dim3(kBlock), 0, s, obs, state_at_pov, rows, row_elems, N, (int64_t)e->cfg.num_envs,
                       e->cfg.num_agents, tail_it ? 1 : 0, (uint32_t)e->cfg.tag_it_type, list, count);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_apply_actions(sgw_engine* e, uint8_t* actions, const int64_t* list, const int64_t* new_actions, int64_t n, void* stream) {
    if (!e || !actions || !new_actions) return fail(SGW_EINVAL, "sgw_apply_actions: NULL argument");
    if (n < 0 || n > (int64_t)e->cfg.num_envs * e->cfg.num_agents) return fail(SGW_EINVAL, "sgw_apply_actions: n = %lld outside [0, E * A]", (long long)n);
    if (n == 0) return SGW_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, kBlock), (int64_t)e->num_cus * 8);
    hipLaunchKernelGGL(resolve_apply_actions, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), actions, list, new_actions, n, (int64_t)e->cfg.num_envs, e->cfg.num_agents);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_gather_rows(const float* src, int64_t row_elems, const int64_t* idx, int64_t n, float* dst, void* stream) {
    if (!src || !idx || !dst || row_elems < 1 || n < 0) return fail(SGW_EINVAL, "sgw_gather_rows: bad argument");
    if (n == 0) return SGW_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, kBlock / 64), 256 * 32);
    const bool v2 = (row_elems & 1) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 7) == 0;
    if (v2) hipLaunchKernelGGL(gather_rows_kernel<2>, dim3(blocks), dim3(kBlock), 0, s, src, row_elems, idx, n, dst);
    else hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(blocks), dim3(kBlock), 0, s, src, row_elems, idx, n, dst);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_choose_actions(sgw_engine* e, const float* values, const int64_t* idx, int64_t n, uint32_t epoch, uint32_t turn, int64_t* out, void* stream) {
    if (!e || !values || !out) return fail(SGW_EINVAL, "sgw_choose_actions: NULL argument");
    if (n < 0 || n > (int64_t)e->cfg.num_envs * e->cfg.num_agents) return fail(SGW_EINVAL, "sgw_choose_actions: n = %lld outside [0, E * A]", (long long)n);
    if ((reinterpret_cast<uintptr_t>(values) & 3) || (reinterpret_cast<uintptr_t>(out) & 7) || (reinterpret_cast<uintptr_t>(idx) & 7))
        return fail(SGW_EINVAL, "sgw_choose_actions: misaligned pointer");
    if (n == 0) return SGW_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, kBlock), (int64_t)e->num_cus * 16);
    hipLaunchKernelGGL(choose_actions_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), e->d_turn, values, (int)e->cfg.num_actions,
                       idx, n, (int64_t)e->cfg.num_envs, (uint32_t)e->cfg.first_env_id, epoch, turn, (uint32_t)e->cfg.seed, (uint32_t)(e->cfg.seed >> 32), out);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_turn_prev_rows(sgw_engine* e, int32_t agent, int32_t count, void* out, void* stream) {
    if (!e || !out) return fail(SGW_EINVAL, "sgw_turn_prev_rows: NULL argument");
    if (agent < 0 || agent >= e->cfg.num_agents) return fail(SGW_EINVAL, "sgw_turn_prev_rows: agent %d out of range", agent);
    if (e->turn_cap[agent] <= 0) return fail(SGW_EINVAL, "sgw_turn_prev_rows: agent %d has no replay states bound (sgw_turn_bind)", agent);
    if (count < 1 || count > e->turn_cap[agent]) return fail(SGW_EINVAL, "sgw_turn_prev_rows: count must be in [1, capacity = %lld]", (long long)e->turn_cap[agent]);
    const int64_t rb = e->turn_row_bytes[agent];
    const bool v16 = (rb & 15) == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(e->turn_states[agent])) & 15) == 0;
    const bool v4 = (rb & 3) == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(e->turn_states[agent])) & 3) == 0;
    const int vec = v16 ? 16 : (v4 ? 4 : 1);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(rb / vec * count, kBlock), (int64_t)e->num_cus * 16));
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint8_t* o = static_cast<uint8_t*>(out);
    if (vec == 16) hipLaunchKernelGGL((turn_prev_rows_kernel<16>), dim3(blocks), dim3(kBlock), 0, s, e->d_turn, agent, count, o, rb);
    else if (vec == 4) hipLaunchKernelGGL((turn_prev_rows_kernel<4>), dim3(blocks), dim3(kBlock), 0, s, e->d_turn, agent, count, o, rb);
    else hipLaunchKernelGGL((turn_prev_rows_kernel<1>), dim3(blocks), dim3(kBlock), 0, s, e->d_turn, agent, count, o, rb);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_turn_end(sgw_engine* e, const void* obs, void* stream) {
    if (!e) return fail(SGW_EINVAL, "sgw_turn_end: NULL engine");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int A = e->cfg.num_agents;
    const int N = e->base.C * e->base.VV;
    if (e->turn_rows && obs) {   // this turn's windows -> the agents' replay rows
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div((int64_t)e->cfg.num_envs * A, kBlock / 64), (int64_t)e->num_cus * 32));   // a wave per window
        const bool even = (N & 1) == 0 && e->turn_rows_even && (reinterpret_cast<uintptr_t>(obs) & 7) == 0;
        if (e->obs_format == SGW_OBS_U8) {
            hipLaunchKernelGGL((turn_commit_kernel<uint8_t, 1>), dim3(blocks), dim3(kBlock), 0, s, e->d_turn, static_cast<const uint8_t*>(obs), (int64_t)e->cfg.num_envs, A, N);
        } else if (even) {
            hipLaunchKernelGGL((turn_commit_kernel<float, 2>), dim3(blocks), dim3(kBlock), 0, s, e->d_turn, static_cast<const float*>(obs), (int64_t)e->cfg.num_envs, A, N);
        } else {
            hipLaunchKernelGGL((turn_commit_kernel<float, 1>), dim3(blocks), dim3(kBlock), 0, s, e->d_turn, static_cast<const float*>(obs), (int64_t)e->cfg.num_envs, A, N);
        }
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(turn_advance_kernel, dim3(1), dim3(SGW_MAX_AGENTS), 0, s, e->d_turn, A);   // every ring advances; the turn counts as completed
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_turn_state(sgw_engine* e, uint32_t* epoch_turn, int64_t* rows, void* stream) {
    if (!e || !epoch_turn) return fail(SGW_EINVAL, "sgw_turn_state: NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TurnState h;
    HIP_TRY(hipMemcpyAsync(&h, e->d_turn, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    epoch_turn[0] = h.epoch; epoch_turn[1] = h.turn;
    if (rows) for (int a = 0; a < e->cfg.num_agents; ++a) rows[a] = h.row[a];
    return SGW_OK;
}

int sgw_set_obs_format(sgw_engine* e, int format) {
    if (!e) return fail(SGW_EINVAL, "sgw_set_obs_format: NULL engine");
    if (format != SGW_OBS_F32 && format != SGW_OBS_U8) return fail(SGW_EINVAL, "unknown observation format %d", format);
    if (format == SGW_OBS_U8 && !e->onehot)
        return fail(SGW_EINVAL, "SGW_OBS_U8 needs a one-hot appearance table (counts are exact small integers)");
    e->obs_format = format;
    return SGW_OK;
}

int sgw_bind_agent_state(sgw_engine* e, uint8_t* agent_state, uint8_t* state_at_pov) {
    if (!e) return fail(SGW_EINVAL, "sgw_bind_agent_state: NULL engine");
    if (!agent_state && state_at_pov) return fail(SGW_EINVAL, "sgw_bind_agent_state: state_at_pov without agent_state");
    e->agent_state = agent_state;
    e->state_at_pov = state_at_pov;
    return SGW_OK;
}

int sgw_bind_row_tail(sgw_engine* e, int kind, int tail_len, const float* table) {
    if (!e) return fail(SGW_EINVAL, "sgw_bind_row_tail: NULL engine");
    if (kind == SGW_TAIL_NONE) { e->tail_kind = SGW_TAIL_NONE; e->tail_len = 0; e->tail_table = nullptr; return SGW_OK; }
    // the whole-env instance of sgw_sweep_observe_rows has a twin that writes the tail (round 6): compiled / loaded HERE, not inside a stream-ordered call; a
    // refusal only costs the capability bit (the sweep alone + sgw_observe_rows do the same in two launches)
    auto tail_twin = [&]() {
        Kernel& k = e->k_sweep_rows_tail;
        if (!k.jit && !k.want.empty() && !k.tried) {
            k.tried = true;
            std::string err;
            k.jit = jit_get(k.want, e->opt, e->arch.c_str(), e->dev, &err);
        }
    };
    if (kind == SGW_TAIL_AGENT_IS_IT) {
        if (e->cfg.agent_rule != SGW_AGENT_RULE_TAG) return fail(SGW_EINVAL, "SGW_TAIL_AGENT_IS_IT is the tail of SGW_AGENT_RULE_TAG agents");
        e->tail_kind = kind; e->tail_len = 1; e->tail_table = nullptr;
        tail_twin();
        return SGW_OK;
    }
    if (kind != SGW_TAIL_POSITION_TABLE) return fail(SGW_EINVAL, "sgw_bind_row_tail: unknown kind %d", kind);
    if (tail_len < 1 || tail_len > 4096 || !table) return fail(SGW_EINVAL, "SGW_TAIL_POSITION_TABLE needs a device table of [H][W][tail_len] floats, 1 <= tail_len <= 4096");
    e->tail_kind = kind; e->tail_len = tail_len; e->tail_table = table;
    tail_twin();
    return SGW_OK;
}

int sgw_bind_agent_dir(sgw_engine* e, uint8_t* agent_dir) {
    if (!e) return fail(SGW_EINVAL, "sgw_bind_agent_dir: NULL engine");
    e->agent_dir = agent_dir;
    return SGW_OK;
}

int sgw_init_agent_state(sgw_engine* e, uint8_t* agent_state, void* stream) {
    if (!e || !agent_state) return fail(SGW_EINVAL, "sgw_init_agent_state: NULL argument");
    Params p = e->base;
    p.agent_state = agent_state;
    const int blocks = (int)std::min<int64_t>(ceil_div(p.E, kBlock), (int64_t)e->num_cus * 8);
    hipLaunchKernelGGL(init_agent_state_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_random_actions(sgw_engine* e, uint8_t* actions, uint32_t epoch, uint32_t turn, void* stream) {
    if (!e || !actions) return fail(SGW_EINVAL, "sgw_random_actions: NULL argument");
    Params p = e->base;
    p.actions = actions; p.epoch = epoch; p.turn = turn;
    const int64_t n = p.E * p.A;
    const int blocks = (int)std::min<int64_t>(ceil_div(n, kBlock), (int64_t)e->num_cus * 8);
    hipLaunchKernelGGL(random_actions_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_reduce_metrics(sgw_engine* e, const double* total_reward, double* out, void* stream) {
    if (!e || !total_reward || !out) return fail(SGW_EINVAL, "sgw_reduce_metrics: NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(reduce_stage1, dim3(kRedBlocks), dim3(kBlock), 0, s, total_reward, e->cfg.num_envs, e->d_part);
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(kBlock), 0, s, e->d_part, e->cfg.num_envs, out);
    HIP_TRY(hipGetLastError());
    return SGW_OK;
}

int sgw_get_status(sgw_engine* e, int32_t* status_out, void* stream) {
    if (!e || !status_out) return fail(SGW_EINVAL, "sgw_get_status: NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int32_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, e->d_status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(e->d_status, 0, sizeof(int32_t), s));
    HIP_TRY(hipStreamSynchronize(s));
    *status_out = v;
    return SGW_OK;
}

int sgw_set_timing(sgw_engine* e, int enable) {
    if (!e) return fail(SGW_EINVAL, "sgw_set_timing: NULL engine");
    if (enable && e->ev0.empty()) {
        e->ev0.resize(kEventPool);
        e->ev1.resize(kEventPool);
        for (int i = 0; i < kEventPool; ++i) {
            HIP_TRY(hipEventCreate(&e->ev0[i]));
            HIP_TRY(hipEventCreate(&e->ev1[i]));
        }
    }
    e->timing = enable != 0;
    e->ev_used = 0;
    e->ms_acc = 0.0;
    e->launches = 0;
    e->series.clear();
    e->series_dropped = 0;
    return SGW_OK;
}

int sgw_get_step_time_ms(sgw_engine* e, double* total_ms, int64_t* launches) {
    if (!e || !total_ms || !launches) return fail(SGW_EINVAL, "sgw_get_step_time_ms: NULL argument");
    if (int rc = time_drain(e)) return rc;
    *total_ms = e->ms_acc;
    *launches = e->launches;
    e->ms_acc = 0.0;
    e->launches = 0;
    return SGW_OK;
}

int sgw_get_step_times_ms(sgw_engine* e, float* out_ms, int64_t capacity, int64_t* count) {
    if (!e || !count || (capacity > 0 && !out_ms)) return fail(SGW_EINVAL, "sgw_get_step_times_ms: NULL argument");
    if (int rc = time_drain(e)) return rc;
    const int64_t n = std::min<int64_t>((int64_t)e->series.size(), std::max<int64_t>(capacity, 0));
    for (int64_t i = 0; i < n; ++i) out_ms[i] = e->series[(size_t)i];
    const int64_t lost = (int64_t)e->series.size() - n + e->series_dropped;
    *count = n;
    e->series.clear();
    e->series_dropped = 0;
    if (lost > 0) {   // the durations read are right; the caller learns that the series is not complete
        fail(SGW_OK, "sgw_get_step_times_ms: %lld launches not returned (capacity %lld, series cap %lld)", (long long)lost,
             (long long)capacity, (long long)kSeriesCap);
        return 1;
    }
    return SGW_OK;
}

int sgw_set_auto_reset(sgw_engine* e, uint32_t max_turns, double* episode_return) {
    if (!e) return fail(SGW_EINVAL, "sgw_set_auto_reset: NULL engine");
    e->auto_max_turns = max_turns;
    e->episode_return = max_turns ? episode_return : nullptr;
    return SGW_OK;
}

int sgw_set_wg_per_cu(sgw_engine* e, int wg_per_cu) {
    if (!e) return fail(SGW_EINVAL, "sgw_set_wg_per_cu: NULL engine");
    if (wg_per_cu < -1 || wg_per_cu > 8) return fail(SGW_EINVAL, "wg_per_cu must be -1 (never cap), 0 (automatic) or 1..8");
    e->wg_per_cu = wg_per_cu;
    return SGW_OK;
}

int sgw_launch_info(sgw_engine* e, char* buf, int64_t capacity) {
    if (!e || !buf || capacity < 1) return fail(SGW_EINVAL, "sgw_launch_info: NULL argument");
    // what a whole-batch, whole-turn sgw_step with 16-byte-aligned observations launches: the kernel, the LDS bytes it
    // REQUESTS (a workgroup-per-CU cap is part of that request) and the workgroups per CU the runtime then admits
    const bool walk = e->k_walk.usable() && e->base.E > e->walk_min_envs && e->base.E <= e->walk_max_envs;
    Params p = e->base;
    p.a0 = 0; p.a1 = p.A; p.flags = SGW_STEP_SWEEP | SGW_STEP_RANDOM_ACTIONS; p.do_move = 1;
    p.obs = reinterpret_cast<float*>(16); p.obs_u8 = e->obs_format == SGW_OBS_U8 ? 1 : 0;
    p.obs_stage = e->fast ? e->obs_stage : 0;
    int cap = 0;
    size_t lds = step_lds_request(e, p, &cap);
    const bool big_staged = e->big && !walk && e->base.E > e->big_stage_min_envs;
    if (e->big && !big_staged) lds -= (size_t)(e->big_threads / 64) * e->big_stage;
    lds = big_cap_lds(e, lds);
    const int threads = e->big ? e->big_threads : kBlock;
    Kernel& k = walk ? e->k_walk : e->k_step;
    if (walk) (void)resolve_kernel(e, k);
    int per_cu = 0;
    hipError_t oe = k.jit ? hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.jit, threads, lds)
                          : (k.host ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.host, threads, lds) : hipErrorInvalidValue);
    if (oe != hipSuccess) per_cu = -1;
    const char* phase = e->k_rows.usable() ? e->k_rows.name() : (e->phase_ok ? (e->onehot ? "phase_kernel<true>" : "phase_kernel<false>") : "the step kernel");
    const char* srows = e->k_sweep_rows.usable() ? ((e->tail_kind != SGW_TAIL_NONE && e->fast && !e->sweep_rows_chunked && e->k_sweep_rows_tail.jit) ? e->k_sweep_rows_tail.name()
                                                                                                                                                   : e->k_sweep_rows.name()) : "-";   // what sgw_sweep_observe_rows launches
    snprintf(buf, (size_t)capacity, "%s group=%d threads=%d lds=%zu env_lds=%d obs_stage=%d stage_agents=%d grid=%d wg_per_cu=%d cap=%s%d phase=%s big_stage=%d specialised=%d sweep_rows=%s",
             k.name(),
             (e->fast || e->big) ? (e->big ? e->big_threads : e->wpe * kWave) : e->group,
             threads, lds, e->step_env_lds, e->obs_stage, e->stage_agents,
             walk ? e->walk_blocks : e->grid_blocks, per_cu,
             e->wg_per_cu == 0 ? "auto:" : (e->wg_per_cu < 0 ? "never:" : "forced:"), cap, phase, big_staged ? e->big_stage : 0,
             k.jit ? 1 : 0, srows);
    return SGW_OK;
}

}  // extern "C"

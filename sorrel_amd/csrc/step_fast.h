// step_fast<...>: a wave per env, worlds <= 4 KiB (the headline kernel and its STAGE / TAG / RULES / MULTI variants).
#pragma once

// ---------------------------------------------------------------- fast step kernel
// Wave-per-env specialisation for worlds whose byte count is a multiple of 16 and
// <= 4 KiB with at most one spawning type (all BASELINE configs up to 32x32x2):
//   * the grid is loaded straight into registers (16 B per lane per unit) one env
//     AHEAD of its use, so HBM latency hides under the previous env's work;
//   * the Bernoulli half of the sweep runs on those registers (byte-parallel
//     spawner match, one Philox block per dword); the rare "what spawns" draw is
//     deferred to a short divergent loop that patches single bytes in LDS;
//   * everything about an agent's move that does not depend on the other agents
//     (action -> target cell, bounds, status) is computed for all agents at once,
//     lane a = agent a; the strictly sequential part is a handful of scalar ops:
//     read the target type from LDS, test passability, patch two bytes;
//   * window geometry (L, C, r, and for the BASELINE shapes H, W) is compile-time,
//     so gather/emit is branch-free: v_cvt_f32_ubyteN + global_store_dword.
constexpr int kMaxUnits = 4;        // 16-byte units per lane (cells <= 4096)
constexpr int kMaxUnitsRules = 8;   // ... of the RULES variant, whose sweep runs in LDS: units beyond the first kMaxUnits per lane go from HBM to LDS in a
                                    // second round (cells <= 8192)
constexpr int kMaxUnitsPlain = 11;  // ... of plain / Tag worlds of large batches on the same second round (cells <= 11 264: three workgroups per CU)
constexpr size_t kLdsPerCu = 160 * 1024;
constexpr size_t kCacheResidentGrid = (size_t)288 << 20;   // on-die capacity: 256 MiB Infinity Cache + 8 x 4 MiB L2; grids of a batch up to this size can stay
                                                            // resident from turn to turn (measured: 256 MiB of grids still do, 512 MiB do not)

// (non-temporal observation stores were measured: slower)
#define OBS_STORE(ptr, val) (*(ptr) = (val))


// Bernoulli draws of one 16-byte unit: returns a 16-bit mask of the cells that spawn.
// The spawner test is a byte compare per cell (the compiler emits it as ONE v_cmp with a byte select on the dword) and meets the
// draw's compare as lane masks in scalar registers; until round 4 a byte-parallel match produced 0x80 flags that every cell then
// tested again in vector registers (24 + 5.5 vector instructions per dword against 16 now; the kernel is vector-issue bound).
template <bool OWN_KEYS>
__device__ __forceinline__ uint32_t sweep_hits(const uint4& u, const uint32_t unit, const Params& p, const uint32_t env_id, const uint32_t turn, const uint32_t ep4) {
    uint32_t hits = 0;
    const uint32_t pat = p.spawn_pat & 0xFFu;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t dv = k == 0 ? u.x : k == 1 ? u.y : k == 2 ? u.z : u.w;
        const bool m0 = (dv & 0xFFu) == pat, m1 = ((dv >> 8) & 0xFFu) == pat, m2 = ((dv >> 16) & 0xFFu) == pat, m3 = (dv >> 24) == pat;
        if (m0 | m1 | m2 | m3) {
            const U4 w = philox4x32_10<OWN_KEYS>(opaque(unit * 4 + k), turn, env_id, ep4 | SGW_STREAM_SPAWN, p.seed_lo, p.seed_hi);
            const bool f = p.spawn_full != 0;
            uint32_t hb = 0;
            hb |= (m0 && (f || w.x < p.spawn_thr)) ? 1u : 0u;
            hb |= (m1 && (f || w.y < p.spawn_thr)) ? 2u : 0u;
            hb |= (m2 && (f || w.z < p.spawn_thr)) ? 4u : 0u;
            hb |= (m3 && (f || w.w < p.spawn_thr)) ? 8u : 0u;
            hits |= hb << (4 * k);
        }
    }
    return hits;
}

// Rare second draw: what spawns in each hit cell; written straight into the LDS grid.
template <bool OWN_KEYS>
__device__ __forceinline__ void sweep_apply(uint32_t hits, const uint32_t unit, uint8_t* lg, const Params& p,
                                            const uint32_t env_id, const uint32_t turn, const uint32_t ep4) {
    while (hits) {
        const uint32_t cell = (uint32_t)__ffs(hits) - 1u;
        hits &= hits - 1u;
        const uint32_t off = unit * 16u + cell;   // byte offset == RNG index
        const U4 kw = philox4x32_10<OWN_KEYS>(opaque(off >> 2), turn, env_id, ep4 | SGW_STREAM_SPAWN_KIND, p.seed_lo, p.seed_hi);
        const uint32_t pick = __umulhi(word_of(kw, off & 3u), p.spawn_n);
        lg[off] = (uint8_t)(((pick < 4 ? p.choice_lo : p.choice_hi) >> (8 * (pick & 3u))) & 0xFFu);
    }
}

#ifdef SGW_STAMPS
// Diagnostic build only (-DSGW_STAMPS, read with tools/stamps.py): coarse s_memrealtime stamps (10 ns, chip-wide) per wave, stored per
// env and segment with plain stores (atomics would serialise), plus where and when the wave started.  Read the
// SHARES, not the run time.  No stamp executes in the product build.
constexpr int kStampEnvs = 65536;
__device__ unsigned long long g_stamps[kStampEnvs * 8];
#define STAMP(i)                                                                                             \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long t_;                                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");   /* 100 MHz, chip-wide */                          \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (lane == 0 && (i) > 0 && env < kStampEnvs) g_stamps[env * 8 + (i)-1] = t_ - tprev_;                \
        tprev_ = t_;                                                                                         \
    } while (0)
#else
#define STAMP(i)
#endif

// RULES: the layered rule set on the wave-per-env kernel -- an ordered LDS sweep, one dword (four
// cells, one Philox block) per lane and layer by layer, for any number of spawners and SGW_RULE_BECOME_IF types,
// and CleanupAgent.act (facing, beams on the layer above, all-layer reward) in the agent loop.
// STAGE (run-time-shape variants): the one-hot observations of `stage_agents` agents at a time are staged as bytes in LDS
// and leave as one burst of streaming 16-byte stores, aligned in GLOBAL memory whatever A * C * V * V is (the chunk's
// first element need not sit on a 16-byte boundary: the staging area is shifted by its misalignment, edge elements
// leave as single stores).  A STAGE kernel has no direct-store path at all (the two together do not fit the 64-register
// budget of 8 waves per SIMD); the host launches the plain variant for calls that cannot be staged (a range of agents,
// SGW_STEP_OBS_NEXT, an observation pointer that is not 16-byte aligned).  The fixed-shape kernels of the BASELINE
// configs keep their own, simpler whole-env burst and ignore the parameter.
// MULTI: the variant sgw_rollout launches for nturns > 1 (a turn loop around sweep / agents / emit, the grid staying in
// LDS).  It is a separate instantiation because the loop costs registers (config 3's kernel: 39 -> 64 VGPRs), which the
// single-turn kernel must not pay.
#ifndef SGW_FAST_RULES_PLAIN_WAVES
#define SGW_FAST_RULES_PLAIN_WAVES 6   // the direct-store RULES instances (agent ranges, OBS_NEXT: the policy path of Cleanup): at 8 the
                                      // 3-layer / 9-channel one spilled 19 registers; Cleanup policy turn 495 -> 346 us (16 384 envs)
#endif
#ifndef SGW_FAST_MULTI_WAVES
#define SGW_FAST_MULTI_WAVES 6   // waves per SIMD the MULTI (sgw_rollout) instances are compiled for: at 8 (64 VGPRs) the config-3
                                 // instance spills two registers, and a scratch reload waits on vmcnt -- behind the observation
                                 // stores in flight; 8 / 7 / 6 / 5 / 4: 112.4 / 107.6 / 102.2 / 103.2 / 103.2 us per turn (config 3, 50 turns)
#endif
// P3 (round 3; RULES instances with one-hot tables of <= 10 channels): 3-bit packed counters -- the appearance of a type is
// ONE word (1 << 3 c for its channel c; a count never exceeds the <= 7 layers), so a cell costs one table read and one add
// per layer instead of ceil(C / 4) of each (Cleanup: 9 channels, three words).
// I16 (round 3; integer appearance tables of <= 4 channels behind the clip / 255 post-processing = the reference's RGBObservationSpec with
// its uint8 colours): the window pipeline of the one-hot path with 16-bit counters -- the layer sum of a cell is an integer, the clip
// makes it a byte, the byte is staged, and the burst turns it into (float)(k / 255.0) through a 256-entry table of exactly those
// floats (DevTables::post_lut, one copy per workgroup in LDS) instead of a float64 sum, clip and DIVISION per cell and channel.
// step_fast_rows' emit (called once by EVERY wave of the workgroup, also one whose env lies beyond the batch):
template <int TL, int TC, int TR, int TH, int TW, bool TAIL = false>
__device__ __forceinline__ void fast_rows_emit(const Params& p, const RowPtrs* rp, uint8_t* smem, const int sub, const int lane) {
    // Agent a's C * V * V staged bytes of env e -> floats at rp->p[a] + e * rp->stride.  The workgroup's four envs are consecutive,
    // so with stride = C * V * V their windows of ONE agent are one run of 4 C V V floats: the waves meet (the only barrier of
    // this kernel), then wave w writes the runs of agents w, w + 4, ... out of all four staging areas -- 16-byte streaming
    // stores with lane 0 on a 128-byte line, as below.  (A wave writing its own env's eight windows, 1 176 bytes each: 121 us
    // at 65 536 envs on a good day, 148-185 on others -- eight short runs per wave, each between two other waves' runs,
    // made the kernel depend on where the rows lie.)
    typedef float vfloat4 __attribute__((ext_vector_type(4)));
    typedef float vfloat2 __attribute__((ext_vector_type(2)));
    constexpr int kN = TC * (2 * TR + 1) * (2 * TR + 1);   // (TR = 0 is the radius 0 itself here: the host instantiates the engine's own radius)
    const int ob_off = p.tab_bytes + ((TL * TH * TW + 15) & ~15);   // (= ob - wl in step_fast_body: [table words][grid][staged windows])
    const int64_t env_first = (int64_t)blockIdx.x * 4;
    const int live = (int)(p.E - env_first < 4 ? p.E - env_first : 4);     // envs of this workgroup
    __syncthreads();
    // floats [0, n * kN) of agent a's windows of envs k0 .. k0 + n - 1, contiguous from dst on
    auto emit_run = [&](float* dst, const int a, const int k0, const int n) {
        const uintptr_t ad = reinterpret_cast<uintptr_t>(dst);
        const int T = n * kN;
        auto src16 = [&](const int t) -> uint32_t {       // staged bytes 2t, 2t + 1 of the run (kN even)
            const int k = t / (kN / 2), j2 = t - k * (kN / 2);
            return *reinterpret_cast<const uint16_t*>(smem + (k0 + k) * p.env_lds + ob_off + a * kN + 2 * j2);
        };
        if ((kN & 1) == 0 && (ad & 15u) == 0) {
            const int q = T >> 2;
            const int mis = (int)((ad >> 4) & 7u);
            for (int i = lane - mis; i < q; i += 64) {
                if (i < 0) continue;
                const uint32_t lo = src16(2 * i), hi = src16(2 * i + 1);
                vfloat4 v;
                v.x = (float)(lo & 0xFFu);
                v.y = (float)(lo >> 8);
                v.z = (float)(hi & 0xFFu);
                v.w = (float)(hi >> 8);
                __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(dst) + i);
            }
            if ((T & 3) && lane == 0) {                   // (an odd number of envs of an odd kN / 2: two floats behind the last float4)
                const uint32_t b = src16(2 * q);
                vfloat2 v;
                v.x = (float)(b & 0xFFu);
                v.y = (float)(b >> 8);
                __builtin_nontemporal_store(v, reinterpret_cast<vfloat2*>(dst) + 2 * q);
            }
        } else if ((kN & 1) == 0 && (ad & 7u) == 0) {
            const int mis = (int)((ad >> 3) & 15u);
            for (int i = lane - mis; i < (T >> 1); i += 64) {
                if (i < 0) continue;
                const uint32_t b = src16(i);
                vfloat2 v;
                v.x = (float)(b & 0xFFu);
                v.y = (float)(b >> 8);
                __builtin_nontemporal_store(v, reinterpret_cast<vfloat2*>(dst) + i);
            }
        } else {
            for (int i = lane; i < T; i += 64) {
                const int k = i / kN, j = i - k * kN;
                __builtin_nontemporal_store((float)smem[(k0 + k) * p.env_lds + ob_off + a * kN + j], dst + i);
            }
        }
    };
    for (int a = sub; a < p.A; a += 4) {
        float* base = static_cast<float*>(rp->p[a]) + env_first * rp->stride;
        if (rp->stride == kN) emit_run(base, a, 0, live);
        else for (int k = 0; k < live; ++k) emit_run(base + k * rp->stride, a, k, 1);
        // a recorded turn (sgw_turn_begin_rows): the windows ALSO go to the agent's replay row of the turn in flight, by the
        // engine's own row count
        if (rp->dual && rp->ts->cap[a] > 0 && rp->ts->states[a]) {
            const int64_t re = rp->ts->row_elems[a];
            float* ring = static_cast<float*>(rp->ts->states[a]) + (rp->ts->row[a] * p.E + env_first) * re;
            if (re == kN) emit_run(ring, a, 0, live);
            else for (int k = 0; k < live; ++k) emit_run(ring + k * re, a, k, 1);
        }
        // (round 6) what pov() appends behind the flattened window (phase.h, observe_rows: the same two kinds); nobody acts in this launch, so the
        // agents' types and cells are the bound tensors'
        // (the TAIL twin only: in the instance without it this code cost 41 spilled scalars -- vector instructions, on a kernel that is bound by them)
        if constexpr (TAIL) if (p.tail_kind != SGW_TAIL_NONE && rp->stride >= kN + p.tail_len) {
            for (int k = 0; k < live; ++k) {
                float* t = base + k * rp->stride + kN;
                const int64_t ea = (env_first + k) * p.A + a;
                if (p.tail_kind == SGW_TAIL_AGENT_IS_IT) {
                    if (lane == 0) t[0] = (p.agent_state && p.agent_state[ea] == p.tag_it) ? 1.f : 0.f;
                } else {
                    const uint32_t yx = reinterpret_cast<const uint16_t*>(p.pos)[ea];
                    const float* src = p.tail_table + ((int64_t)(yx & 0xFFu) * TW + (yx >> 8)) * p.tail_len;
                    for (int j = lane; j < p.tail_len; j += 64) t[j] = src[j];
                }
            }
        }
    }
}

// ROWS (round 5): the instance behind sgw_sweep_observe_rows -- the sweep and EVERY agent's window in one launch, each window going to
// its agent's own destination (rp->p[a] + env * rp->stride: the row of that agent's replay buffer) instead of the [E][A][C][V][V] tensor.
// Only the emit differs (see there); compiled for compile-time shapes with the whole-env burst.
// ROWX (round 6): the same call on the instances that stage CHUNKS of agents (STAGE: layered rule sets -- Cleanup --, Tag, run-time maps and tables): the host
// launches with one agent per chunk, and a chunk leaves for its agent's own row -- emit_chunk's line-aligned 16-byte streaming stores, the staging offset
// taken from the ROW's address -- followed by the bound row tail (TagAgent.pov's flag, CleanupObservation's positional code).
template <bool ONEHOT, int TL, int TC, int TR, int TH, int TW, bool TAG, bool RULES, bool STAGE, bool MULTI, bool P3, bool I16, bool ROWS, bool ROWX = false, bool TAIL = false>
__device__ __forceinline__ void step_fast_body(const Params p, [[maybe_unused]] const RowPtrs* rp) {
    static_assert(!ROWS || (ONEHOT && TL && TC && TH && TW && !RULES && !STAGE && !MULTI && !P3 && !I16), "ROWS: plain or Tag movers, one-hot, compile-time shape");
    static_assert(!ROWX || (ONEHOT && STAGE && !MULTI && !I16 && !ROWS), "ROWX: a chunk-staging single-turn instance");
    // One wave = one env, one pass: no persistent loop (letting the dispatcher hand out
    // workgroups measured 17 % faster than a persistent grid with software prefetch),
    // wave-private LDS (grid slice + the table words this wave reads), no s_barrier.
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps all env-indexed address math scalar
    const int64_t env = (int64_t)blockIdx.x * 4 + sub;
    if (env >= p.E) {         // whole wave exits together
        if constexpr (ROWS)      // (the workgroup's waves meet once, in the emit, and every wave writes its share of the agents)
            if (p.obs_stage > 0 && p.a0 == 0 && p.a1 == p.A && !(p.flags & SGW_STEP_NO_OBS))   // (= `stage && write_obs` of the waves that have an env)
                fast_rows_emit<TL, TC, TR, TH, TW, TAIL>(p, rp, smem, sub, lane);
        return;
    }
#ifdef SGW_STAMPS
    unsigned long long tprev_ = 0;
    STAMP(0);
    if (lane == 0 && env < kStampEnvs) {   // where and when this wave started
        g_stamps[env * 8 + 6] = tprev_;
        g_stamps[env * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
#endif

    const int L = TL ? TL : p.L;
    const int C = TC ? TC : p.C;
    const int r = TR ? TR : p.r;
    const int V = 2 * r + 1, VV = V * V;
    const int H = TH ? TH : p.H, W = TW ? TW : p.W, HW = H * W;
    constexpr bool kStatic = TL && TH && TW;
    constexpr bool kOwnKeys = !kStatic;   // Philox key schedule per block (common.h): pays for the run-time-shape instances only
    const int cells = kStatic ? TL * TH * TW : p.cells;
    const int nunits = (cells + 15) >> 4;   // the last unit may be partly padding (env stride is a multiple of 16)
    constexpr int NU = kStatic ? ((TL * TH * TW + 15) / 16 + 63) / 64 : kMaxUnits;   // units per lane (RULES: the first round)
    // compile-time shapes: rounds of the register sweep in which every lane holds a unit, and whether the partly filled round behind
    // them is swept as dwords on the LDS copy instead (it is when that takes fewer Philox blocks than the round's four)
    constexpr int kFullRounds = kStatic ? ((TL * TH * TW + 15) / 16) / 64 : 0;
    constexpr int kTailUnits = kStatic ? ((TL * TH * TW + 15) / 16) % 64 : 0;
    constexpr bool kTailDword = kStatic && !RULES && kTailUnits > 0 && kTailUnits <= 48;
    const int zoff = p.zA * HW;
    static_assert(!I16 || (ONEHOT && !P3 && !RULES && !(TL && TH && TW)), "I16: a byte-staging instance with run-time map");
    constexpr int NW = P3 ? 1 : I16 ? 2 : (TC ? (TC + 3) / 4 : 4);   // counter words
    constexpr int CMAX = P3 ? (TC ? TC : 10) : I16 ? (TC ? TC : 4) : 4 * NW;      // channels the counter words can hold

    // wave-private LDS: [table words][grid]
    uint8_t* wl = smem + sub * p.env_lds;
    const DevTables* gtab = p.tab;
    const uint32_t env_id = p.first_env + (uint32_t)env;
    uint32_t turn0 = p.turn, ep4 = p.epoch << 4;     // Environment.turn and the epoch: kernel arguments, or (sgw_turn_*) the engine's device-side count
    if (p.ts) { turn0 = p.ts->turn + 1u; ep4 = p.ts->epoch << 4; }   // (ts->turn: turns completed)

    // ---- issue every global load of this env first
    uint4 u[NU];
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.grid + env * p.env_stride);
#pragma unroll
        for (int k = 0; k < NU; ++k)
            if (lane + 64 * k < nunits) u[k] = src[lane + 64 * k];
    }
    const bool mine = lane >= p.a0 && lane < p.a1 && lane < p.A;   // this lane's agent is stepped in this call
    const bool rnd = (p.flags & SGW_STEP_RANDOM_ACTIONS) != 0;
    uint32_t yx = 0, act = 0;
    if (lane < p.A) yx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + lane];
    if ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W) {   // garbage in: stay inside this env's LDS slice, and say so
        yx = 0;
        atomicOr(p.status, SGW_STATUS_BAD_POS);
    }
    if (mine && p.do_move && !rnd) act = p.actions[env * p.A + lane];
    // register-resident tables: lane t holds value[t] (f64 bits + its f32 rounding); lane a holds agent a's type
    const double vtab = gtab->value[lane & 31];
    uint32_t atype = gtab->agent_type[lane];   // lane a: CURRENT entity type of agent a
    if (p.agent_state && lane < p.A) atype = p.agent_state[env * p.A + lane];
    uint32_t pov_type = atype;                 // ... and its type when it observed (TagAgent.pov)
    if constexpr (ONEHOT) {
        // the one-hot counter words this wave looks up, [NW][32] u32
        uint32_t* wd = reinterpret_cast<uint32_t*>(wl);
        if constexpr (I16) {
            wd[lane] = reinterpret_cast<const uint32_t*>(gtab->delta16)[lane];           // [2][32]
            float* l = reinterpret_cast<float*>(smem + 4 * p.env_lds);                   // the result table: one copy per workgroup, every wave writes the same floats
#pragma unroll
            for (int j = 0; j < 4; ++j) l[lane + 64 * j] = gtab->post_lut[lane + 64 * j];
        } else if constexpr (P3) {
            if (lane < SGW_MAX_TYPES) wd[lane] = gtab->delta3[lane];
        } else {
#pragma unroll
            for (int q = 0; q < (NW + 1) / 2; ++q) wd[lane + 64 * q] = reinterpret_cast<const uint32_t*>(gtab->delta)[lane + 64 * q];
        }
    } else {
        double* wa = reinterpret_cast<double*>(wl);
        for (int i = lane; i < SGW_MAX_TYPES * SGW_MAX_CHANNELS; i += 64) wa[i] = reinterpret_cast<const double*>(gtab->appearance)[i];
    }
    const uint32_t* wdelta = reinterpret_cast<const uint32_t*>(wl);                 // [NW][32]
    // a staged / counted value as the float the tensor holds: the count itself, or (I16) the post-processed colour out of the result table
    [[maybe_unused]] const float* post_lut = reinterpret_cast<const float*>(smem + 4 * p.env_lds);
    auto tof = [&](const uint32_t k) -> float {
        if constexpr (I16) return post_lut[k];
        else return (float)k;
    };
    const double(*wapp)[SGW_MAX_CHANNELS] = reinterpret_cast<const double(*)[SGW_MAX_CHANNELS]>(wl);
    uint8_t* lg = wl + p.tab_bytes + (RULES ? kRuleLds : 0);
    uint4* lg16 = reinterpret_cast<uint4*>(lg);
    [[maybe_unused]] const RuleLds* rt = reinterpret_cast<const RuleLds*>(wl + p.tab_bytes);
    [[maybe_unused]] uint32_t adir = 2;        // lane a: facing of agent a (Cleanup)
    [[maybe_unused]] uint32_t kind_v = 0;      // lane a: SGW_ACTION_* of its action
    if constexpr (RULES) {
        uint32_t* rd = reinterpret_cast<uint32_t*>(wl + p.tab_bytes);
        const uint32_t* gA = reinterpret_cast<const uint32_t*>(gtab->thr_lo);
        const uint32_t* gB = reinterpret_cast<const uint32_t*>(gtab->spawn_choice);   // + spawn_count: 72 dwords
        const uint32_t* gC = reinterpret_cast<const uint32_t*>(gtab->rule);           // rule .. rule_mask: 64 dwords
        if (lane < 32) rd[lane] = gA[lane];
        rd[32 + lane] = gB[lane];
        if (lane < 8) rd[96 + lane] = gB[64 + lane];
        rd[104 + lane] = gC[lane];
        if (p.agent_dir && lane < p.A) adir = p.agent_dir[env * p.A + lane];
    }
    // One-hot observations of a whole env are staged in LDS as byte counts in their final [A][C][V][V] order and
    // leave for HBM in one burst of 16-byte stores after the agent loop (instead of 6 dword stores per agent
    // dribbling out over the wave's life): the chip then has far fewer half-written observation streams open.
    uint8_t* ob = lg + ((cells + 15) & ~15);
    constexpr bool kStageAlways = ONEHOT && STAGE;   // (compile-time shapes too: Cleanup as shipped stages bursts of agents)
    const bool stage = kStageAlways || (ONEHOT && kStatic && p.obs_stage > 0 && p.a0 == 0 && p.a1 == p.A);
#ifdef SGW_DIAG_NO_LINE_ALIGN
    constexpr uint32_t kLineMask = 3u;     // diagnostic A/B: 16-byte aligned chunks as before round 3
#else
    constexpr uint32_t kLineMask = 31u;    // elements per 128-byte line of f32 observations, minus one
#endif
    [[maybe_unused]] int ch_a0 = 0;            // first agent of the chunk being staged (STAGE)
    [[maybe_unused]] uint32_t ch_shift = 0;    // staging byte of the chunk's first element = its offset (in elements) from a 128-byte line of global memory
    [[maybe_unused]] uint32_t ch_lo = 0;       // first staged byte that has not left yet (bytes carried over from the chunk before sit in front of ch_shift)
    if constexpr (kStageAlways) ch_lo = ch_shift = (uint32_t)(env * (int64_t)(p.A * C * VV)) & kLineMask;
    [[maybe_unused]] float* row_dst = nullptr; // ROWX: where the chunk's (one) agent's window goes

    // per-lane window geometry: up to two cells per lane
    int wdi[2], wdj[2], woff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int w = lane + 64 * k;
        const int i = w / V, j = w - i * V;
        wdi[k] = i - r;
        wdj[k] = j - r;
        woff[k] = wdi[k] * W + wdj[k];
    }
    const uint32_t vt_lo = (uint32_t)__double_as_longlong(vtab), vt_hi = (uint32_t)(__double_as_longlong(vtab) >> 32);
    const uint32_t vt_f32 = __float_as_uint((float)vtab);
    const bool write_obs = !(p.flags & SGW_STEP_NO_OBS);
    const bool do_sweep = (p.flags & SGW_STEP_SWEEP) != 0;
    const bool dirty = do_sweep || (p.do_move && p.a1 > p.a0);
    const bool sweep_only = !RULES && (MULTI ? p.nturns : 1u) == 1u && !(p.do_move && p.a1 > p.a0);   // nobody acts in this launch

    {
        double tot = p.do_move ? p.total[env] : 0.0;

#ifdef SGW_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(1);   // global loads have arrived
        // ---- grid -> LDS; the Bernoulli half of the sweep runs on the registers
        [[maybe_unused]] uint32_t hits[NU];
        if constexpr (!kStatic || ((TL * TH * TW) & 15) != 0) {
            if (cells & 15) {   // ragged world: bytes past the last cell are not cells (no type, no RNG index)
#pragma unroll
                for (int k = 0; k < NU; ++k)
                    if (lane + 64 * k == nunits - 1) {
                        const int tail = cells & 15;
                        uint32_t d[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int keep = tail - 4 * q;   // valid bytes in this dword
                            if (keep <= 0) d[q] = 0xFFFFFFFFu;
                            else if (keep < 4) d[q] |= 0xFFFFFFFFu << (8 * keep);
                        }
                        u[k] = make_uint4(d[0], d[1], d[2], d[3]);
                    }
            }
        }
        // the env's grid goes to LDS once; sgw_rollout's turns (nturns > 1) all run on it
#pragma unroll
        for (int k = 0; k < NU; ++k)
            if (lane + 64 * k < nunits) lg16[lane + 64 * k] = u[k];
        if constexpr (RULES || !kStatic) {
            // worlds above 4 KiB per env (up to 8 KiB: layered rule sets; plain and Tag worlds of large batches): the rest of the grid in a
            // second round, straight to LDS
            const uint4* src = reinterpret_cast<const uint4*>(p.grid + env * p.env_stride);
            for (int i = 64 * NU + lane; i < nunits; i += 64) {
                uint4 v = src[i];
                if ((cells & 15) && i == nunits - 1) {
                    const int tail = cells & 15;
                    uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int keep = tail - 4 * q;
                        if (keep <= 0) d[q] = 0xFFFFFFFFu;
                        else if (keep < 4) d[q] |= 0xFFFFFFFFu << (8 * keep);
                    }
                    v = make_uint4(d[0], d[1], d[2], d[3]);
                }
                lg16[i] = v;
            }
        }
        int st_lane = 0;
        uint32_t taddr_v = 0xFFFFFFFFu, oaddr_v = 0, npos = 0, rew_bits = 0, moved = 0;   // per turn; the write-back reads the last turn's
        const uint32_t nturns = MULTI ? p.nturns : 1u;
        for (uint32_t tix = 0; tix < nturns; ++tix) {
        const uint32_t turn = turn0 + tix;
        if constexpr (RULES) {
            gsync<1>();
            if (do_sweep) {
                // Ordered sweep in LDS (common.h): layer by layer, lower layers first, a dword (four cells, one Philox block) per lane
                sweep_ordered<1, 64>(p, rt, lg, env_id, lane, turn, ep4, L, HW);
            }
        } else if constexpr (!kStatic) {
            // run-time shapes: one dword per lane and round on the LDS copy.  The unit-per-lane form below leaves a last
            // round with a handful of lanes paying four Philox blocks each (32x33x2: 132 units = 64 + 64 + 4), and a wave
            // skips the blocks of a round only when NO lane of it holds a spawner -- dword rounds are 4x finer on both counts
            gsync<1>();
            if (do_sweep) {
                sweep_single<64>(p, lg, env_id, lane, turn, ep4,
                                 sweep_only ? reinterpret_cast<uint32_t*>(p.grid + env * p.env_stride) : nullptr);
                gsync<1>();
            }
        } else {
            if (tix > 0) {   // later turns of a rollout: the units come back from LDS (moves and spawns of the turns before)
                gsync<1>();
#pragma unroll
                for (int k = 0; k < NU; ++k)
                    if (lane + 64 * k < nunits) u[k] = lg16[lane + 64 * k];
            }
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                hits[k] = 0;
                if ((k < kFullRounds || !kTailDword) && lane + 64 * k < nunits && do_sweep)
                    hits[k] = sweep_hits<kOwnKeys>(u[k], (uint32_t)(lane + 64 * k), p, env_id, turn, ep4);
            }
            gsync<1>();
            if (do_sweep) {
                // the rare second draws of ALL the lane's units in ONE loop: the wave pays a Philox block per iteration whatever the number
                // of lanes that still hold a hit, so a loop per unit cost ~1.2 blocks per spawning LAYER (two or three in own-entity
                // worlds); merged, the iterations are the largest hit count of any lane over all its units (~1.5 blocks)
                auto kind_draw = [&](const uint32_t b) {
                    const uint32_t off = ((uint32_t)lane + 64u * (b >> 4)) * 16u + (b & 15u);   // byte offset == RNG index
                    const U4 kw = philox4x32_10<kOwnKeys>(opaque(off >> 2), turn, env_id, ep4 | SGW_STREAM_SPAWN_KIND, p.seed_lo, p.seed_hi);
                    const uint32_t pick = __umulhi(word_of(kw, off & 3u), p.spawn_n);
                    lg[off] = (uint8_t)(((pick < 4 ? p.choice_lo : p.choice_hi) >> (8 * (pick & 3u))) & 0xFFu);
                };
                if constexpr (NU <= 2) {
                    uint32_t hm = hits[0];
                    if constexpr (NU == 2) hm |= hits[1] << 16;
                    while (hm) {
                        const uint32_t b = (uint32_t)__ffs(hm) - 1u;
                        hm &= hm - 1u;
                        kind_draw(b);
                    }
                } else {
                    static_assert(NU <= 4, "16 hit bits per unit, four units per lane");
                    uint64_t hm = 0;
#pragma unroll
                    for (int k = 0; k < NU; ++k) hm |= (uint64_t)hits[k] << (16 * k);
                    while (hm) {
                        const uint32_t b = (uint32_t)__ffsll((unsigned long long)hm) - 1u;
                        hm &= hm - 1ull;
                        kind_draw(b);
                    }
                }
                // the last, partly filled round of units (a 24x24x2 map: 72 units = one full round + 8) as DWORDS on the LDS copy: a
                // wave pays a round's four Philox blocks whether 8 or 64 of its lanes hold a unit, the dword rounds of the same cells
                // cost ceil(tail dwords / 64) blocks (24x24x2: 8 -> 5 blocks per turn, 32x33x2: 12 -> 9)
                if constexpr (kTailDword)
                    sweep_single<64>(p, lg, env_id, lane, turn, ep4, sweep_only ? reinterpret_cast<uint32_t*>(p.grid + env * p.env_stride) : nullptr,
                                     256 * kFullRounds);
                gsync<1>();
            }
        }

        STAMP(2);   // sweep done
        // ---- everything about agent `lane`'s move that does not depend on the other agents
        const uint32_t py = yx & 0xFFu, px = yx >> 8;
        taddr_v = 0xFFFFFFFFu;                   // target cell (LDS byte offset) or "invalid"
        npos = yx;                               // position if the move succeeds
        if (p.do_move && mine) {
            if (rnd) {
                const U4 w = philox4x32_10<kOwnKeys>(opaque((uint32_t)lane >> 2), turn, env_id, ep4 | SGW_STREAM_ACTION,
                                           p.seed_lo, p.seed_hi);
                act = __umulhi(word_of(w, lane & 3), (uint32_t)p.nact);
                p.actions[tix * p.ts_act + env * p.A + lane] = (uint8_t)act;
            } else if (tix > 0) {
                act = p.actions[tix * p.ts_act + env * p.A + lane];
            }
            const bool act_ok = act < (uint32_t)p.nact;
            int dy = (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1;
            int dx = (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1;
            if constexpr (RULES) {
                if (p.agent_rule == SGW_AGENT_RULE_CLEANUP) {   // clean / zap stay in place; a move action also turns the agent
                    const uint32_t kind = act_ok ? (p.kind_pack >> (2 * (act & 15u))) & 3u : 0u;
                    if (kind != SGW_ACTION_MOVE || !act_ok) dy = dx = 0;
                    const uint32_t ndir = (dy == -1 && dx == 0) ? 0u : (dy == 1 && dx == 0) ? 2u : (dy == 0 && dx == -1) ? 3u : (dy == 0 && dx == 1) ? 1u : 4u;
                    kind_v = kind | (act_ok ? 4u : 0u) | (ndir << 4);
                }
            }
            const int ty = (int)py + dy, tx = (int)px + dx;
            const bool inb = (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
            if (act_ok && inb) {
                taddr_v = (uint32_t)(zoff + ty * W + tx);
                npos = (uint32_t)ty | ((uint32_t)tx << 8);
            }
            st_lane |= !act_ok ? SGW_STATUS_BAD_ACTION : (!inb ? SGW_STATUS_OOB_MOVE : 0);
        }
        oaddr_v = (uint32_t)zoff + py * (uint32_t)W + px;   // own cell
        rew_bits = 0;
        moved = 0;
        const int64_t turn_obs = tix * p.ts_obs;   // this turn's observation slot (elements)
        if constexpr (kStageAlways) {
            ch_a0 = 0;
            ch_lo = ch_shift = (uint32_t)(turn_obs + env * (int64_t)(p.A * C * VV)) & kLineMask;
        }

        STAMP(3);   // move inputs (action draw) done
        // STAGE: the staged chunk [a_lo, a_hi) leaves for HBM.  Byte s of the staging area is element (e0 - sh) + s of the
        // observation tensor, and e0 - sh is a multiple of 32 elements: dword i of the staging area is float4 number i of a
        // 128-BYTE LINE-ALIGNED span, so lane 0 of every wave-wide store sits on a line boundary and the store covers eight
        // whole lines.  (Round 3.  A streaming store that covers PART of a line is expensive -- tools/micro/region_writer.hip:
        // the same 1 KiB stores shifted by 16 / 32 / 64 bytes write 4.44 / 4.44 / 4.91 TB/s against 5.48 aligned -- and with
        // 16-byte alignment only, every store of a chunk had a partial line at both ends.)  A chunk that is not the env's last
        // leaves only up to its last line boundary; the < 32 bytes behind it are carried to the front of the staging area and
        // leave with the next chunk.  What remains partial: the env's first and last line (shared with the neighbouring envs'
        // waves), element-wise where they do not fill a float4.
        [[maybe_unused]] auto emit_chunk = [&](const int a_lo, const int a_hi, const bool last) {
            gsync<1>();
            typedef float vfloat4 __attribute__((ext_vector_type(4)));
            const int N = (a_hi - a_lo) * C * VV;
            const int64_t e0 = turn_obs + (env * p.A + a_lo) * (int64_t)(C * VV);
            const int sh = (int)ch_shift, lo = (int)ch_lo;
            const int hi = sh + N;
            const int he = last ? hi : (hi & ~(int)kLineMask);    // bytes [lo, he) leave now
            if (he <= lo) {                           // (a chunk that ends inside the env's first line: nothing to write yet)
                ch_shift = (uint32_t)hi;
                return;
            }
            const uint32_t* ob4 = reinterpret_cast<const uint32_t*>(ob);
            const int i0 = (lo + 3) >> 2, i1 = he >> 2;   // dwords [i0, i1) lie wholly inside [lo, he): unconditional 16-byte streaming stores
            // edge dwords (the env's first / last float4, partly another env's): element by element, by lanes 0 / 1
            const int ie = lane == 0 ? i0 - 1 : i1;
            const bool edge = lane == 0 ? (lo & 3) != 0 : (lane == 1 && (he & 3) != 0 && (i1 >= i0 || (lo & 3) == 0));
            if (ROWX || !p.obs_u8) {
                float* gb = ROWX ? row_dst - sh : p.obs + (e0 - sh);
                for (int i = lane; i < i1; i += 64) {
                    if (i < i0) continue;
                    const uint32_t b = ob4[i];
                    vfloat4 v;
                    v.x = tof(b & 0xFFu);
                    v.y = tof((b >> 8) & 0xFFu);
                    v.z = tof((b >> 16) & 0xFFu);
                    v.w = tof(b >> 24);
#ifdef SGW_DIAG_PLAIN_STORES
                    *reinterpret_cast<vfloat4*>(gb + 4 * i) = v;
#else
                    __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(gb + 4 * i));
#endif
                }
                if (edge) {
                    const uint32_t b = ob4[ie];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * ie + j >= lo && 4 * ie + j < he) gb[4 * ie + j] = tof((b >> (8 * j)) & 0xFFu);
                }
            } else {
                uint8_t* gb = reinterpret_cast<uint8_t*>(p.obs) + (e0 - sh);
                for (int i = lane; i < i1; i += 64) {
                    if (i < i0) continue;
                    __builtin_nontemporal_store(ob4[i], reinterpret_cast<uint32_t*>(gb + 4 * i));
                }
                if (edge) {
                    const uint32_t b = ob4[ie];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * ie + j >= lo && 4 * ie + j < he) gb[4 * ie + j] = (uint8_t)(b >> (8 * j));
                }
            }
            if constexpr (ROWX) {
                if (p.tail_kind != SGW_TAIL_NONE) {      // what pov() appends behind the flattened window (phase.h, observe_rows: the same two kinds)
                    float* t = row_dst + C * VV;
                    if (p.tail_kind == SGW_TAIL_AGENT_IS_IT) {
                        const uint32_t ty_ = (uint32_t)__builtin_amdgcn_readlane((int)atype, a_lo);
                        if (lane == 0) t[0] = ty_ == p.tag_it ? 1.f : 0.f;
                    } else {
                        const int ay_ = __builtin_amdgcn_readlane((int)(yx & 0xFFu), a_lo), ax_ = __builtin_amdgcn_readlane((int)(yx >> 8), a_lo);
                        const float* src = p.tail_table + ((int64_t)ay_ * W + ax_) * p.tail_len;
                        for (int k = lane; k < p.tail_len; k += 64) t[k] = src[k];
                    }
                }
            }
            if (!last) {   // the bytes behind the last line boundary: to the front, they leave with the next chunk
                uint8_t t = 0;
                if (lane < hi - he) t = ob[he + lane];
                gsync<1>();
                if (lane < hi - he) ob[lane] = t;
                ch_shift = (uint32_t)(hi - he);
                ch_lo = 0;
            }
            gsync<1>();
        };
        // ---- agents, strictly in list order (SGW_STEP_OBS_NEXT: one extra, observe-only iteration for agent a1)
        const int a_end = (p.obs_next && p.a1 < p.A) ? p.a1 + 1 : p.a1;
        for (int a = p.a0; a < a_end; ++a) {
            if constexpr (ROWX) {                    // a chunk = one agent = one row: the agent before leaves, this one's row sets the staging offset
                if (a > p.a0 && write_obs) emit_chunk(a - 1, a, true);
                ch_a0 = a;
                row_dst = static_cast<float*>(rp->p[a]) + env * rp->stride;
                ch_lo = ch_shift = (uint32_t)(reinterpret_cast<uintptr_t>(row_dst) >> 2) & kLineMask;
            } else if constexpr (kStageAlways) {
                if (a - ch_a0 == p.stage_agents) {   // the staging area is full: out with it, start the next chunk
                    if (write_obs) emit_chunk(ch_a0, a, false);   // (moves ch_shift / ch_lo on to the next chunk)
                    ch_a0 = a;
                }
            }
            const int s_o = __builtin_amdgcn_readlane((int)oaddr_v, a);
            if (p.obs_next ? a == p.a1 : write_obs) {
                const int y = __builtin_amdgcn_readlane((int)py, a);
                const int x = __builtin_amdgcn_readlane((int)px, a);
                const int cbase = s_o - zoff;
                float* obase = p.obs + turn_obs + ((env * p.obs_A + (a - p.obs_a0)) * (int64_t)C) * VV;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (64 * k >= VV) break;
                    const int w = lane + 64 * k;
                    if (w < VV) {
#ifdef SGW_DIAG_SKIP_GATHER
                        if (p.turn != 0xFFFFFFFFu) continue;
#endif
                        const bool inb = (unsigned)(y + wdi[k]) < (unsigned)H && (unsigned)(x + wdj[k]) < (unsigned)W;
                        const int off = inb ? cbase + woff[k] : 0;   // clamped: the read is always in range
                        float* o = obase + w;
                        if constexpr (ONEHOT) {
                            uint32_t cnt[NW];
#pragma unroll
                            for (int q = 0; q < NW; ++q) cnt[q] = 0;
#pragma unroll
                            for (int z = 0; z < (TL ? TL : 1); ++z) {
                                const uint32_t t = lg[z * HW + off] & 31u;
#pragma unroll
                                for (int q = 0; q < NW; ++q) cnt[q] += wdelta[q * 32 + t];
                            }
                            if constexpr (TL == 0) {
                                for (int z = 1; z < L; ++z) {
                                    const uint32_t t = lg[z * HW + off] & 31u;
#pragma unroll
                                    for (int q = 0; q < NW; ++q) cnt[q] += wdelta[q * 32 + t];
                                }
                            }
#pragma unroll
                            for (int q = 0; q < NW; ++q) cnt[q] = inb ? cnt[q] : (P3 ? p.fill_delta3 : I16 ? p.fill_delta16[q] : p.fill_delta[q]);
                            // the count of channel c: a byte of the counter words, (P3) a 3-bit field of the one word, (I16) a 16-bit field clipped to 255
                            auto chan = [&](const int c) -> uint32_t {
                                if constexpr (P3) return (cnt[0] >> (3 * c)) & 7u;
                                else if constexpr (I16) return min((cnt[c >> 1] >> (16 * (c & 1))) & 0xFFFFu, 255u);
                                else return (cnt[c >> 2] >> (8 * (c & 3))) & 0xFFu;
                            };
                            if (stage) {
                                uint8_t* os = ob + (kStageAlways ? (int)ch_shift + ((a - ch_a0) * C) * VV : (a * C) * VV) + w;
                                if constexpr (TC != 0) {
#pragma unroll
                                    for (int c = 0; c < CMAX; ++c)
                                        if (c < C) os[c * VV] = (uint8_t)chan(c);
                                } else {
                                    // run-time channel count: the planes go out in groups of four behind ONE test per group (a test per
                                    // plane is a scalar branch each, on conditions that end up spilled into vector lanes).  The up to three
                                    // planes past C land in the next agent's area, which that agent rewrites, or -- behind a chunk's last
                                    // agent -- in the 3 * V * V bytes of slack the host adds to the staging area for these instances
#pragma unroll
                                    for (int g = 0; g < (CMAX + 3) / 4; ++g)
                                        if (4 * g < C) {
#pragma unroll
                                            for (int b = 0; b < 4; ++b)
                                                if (4 * g + b < CMAX) os[(4 * g + b) * VV] = (uint8_t)chan(4 * g + b);
                                        }
                                }
                            } else if constexpr (kStageAlways) {
                                // unreachable: a STAGE kernel always stages
                            } else if (!p.obs_u8) {
#pragma unroll
                                for (int c = 0; c < CMAX; ++c)
                                    if (c < C) OBS_STORE(o + c * VV, tof(chan(c)));
                            } else {   // compact format: the same counts as bytes
                                uint8_t* o8 = reinterpret_cast<uint8_t*>(p.obs) + (o - p.obs);
#pragma unroll
                                for (int c = 0; c < CMAX; ++c)
                                    if (c < C) o8[c * VV] = (uint8_t)chan(c);
                            }
                        } else {
                            for (int c = 0; c < C; ++c) {
                                double acc = wapp[lg[off] & 31u][c];   // left-to-right float64 layer sum
                                for (int z = 1; z < L; ++z) acc += wapp[lg[z * HW + off] & 31u][c];
                                OBS_STORE(o + c * VV, obs_finish(inb ? acc : wapp[p.fill_type][c], p.obs_post));
                            }
                        }
                    }
                }
            }
            if (!p.do_move || a >= p.a1) continue;
            // ---- the sequential part (agent.py:219-221, gridworld.py:110-122): scalar
            const uint32_t s_t = (uint32_t)__builtin_amdgcn_readlane((int)taddr_v, a);
            const uint32_t my_type = (uint32_t)__builtin_amdgcn_readlane((int)atype, a);
            const bool valid = s_t != 0xFFFFFFFFu;
            if constexpr (RULES) {
                if (p.agent_rule == SGW_AGENT_RULE_CLEANUP) {
                    // ---- CleanupAgent.act (sorrel/examples/cleanup/agents.py:92-177); everything below is wave-uniform
                    const uint32_t kd = (uint32_t)__builtin_amdgcn_readlane((int)kind_v, a);
                    const uint32_t kind = kd & 3u, ndir = kd >> 4;
                    const bool aok = (kd & 4u) != 0;
                    const uint32_t facing = (uint32_t)__builtin_amdgcn_readlane((int)adir, a) & 3u;
                    const int ay = __builtin_amdgcn_readlane((int)py, a), ax = __builtin_amdgcn_readlane((int)px, a);
                    if (aok && kind != SGW_ACTION_MOVE && p.zA + 1 < L) {
                        // beam cells on the layer above: 1..R ahead; 0..R-1 ahead of the right / left neighbours
                        if (lane < 3 * p.beam_radius) {
                            const int arm = lane / p.beam_radius, i = lane - arm * p.beam_radius;
                            const int fy = facing == 0 ? -1 : facing == 2 ? 1 : 0, fx = facing == 1 ? 1 : facing == 3 ? -1 : 0;
                            const int ry = facing == 1 ? 1 : facing == 3 ? -1 : 0, rx = facing == 0 ? 1 : facing == 2 ? -1 : 0;
                            const int step = arm == 0 ? i + 1 : i, side = arm == 0 ? 0 : (arm == 1 ? 1 : -1);
                            const int by = ay + side * ry + step * fy, bx = ax + side * rx + step * fx;
                            if ((unsigned)by < (unsigned)H && (unsigned)bx < (unsigned)W) {
                                const int boff = (p.zA + 1) * HW + by * W + bx;
                                if (!((p.beam_block_mask >> (lg[boff] & 31u)) & 1u))
                                    lg[boff] = (uint8_t)(kind == SGW_ACTION_CLEAN ? p.clean_beam : p.zap_beam);
                            }
                        }
                        gsync<1>();
                    }
                    double val = 0.0;           // reward: every layer of the target cell, BEFORE the move
                    uint32_t t = 0xFFu;
                    if (valid) {
                        const int tc = (int)s_t - zoff;
                        for (int zl = 0; zl < L; ++zl) {
                            const uint32_t tz = (uint32_t)__builtin_amdgcn_readfirstlane((int)lg[zl * HW + tc]) & 31u;
                            const uint32_t lo_ = (uint32_t)__builtin_amdgcn_readlane((int)vt_lo, (int)tz);
                            const uint32_t hi_ = (uint32_t)__builtin_amdgcn_readlane((int)vt_hi, (int)tz);
                            val += __longlong_as_double(((long long)hi_ << 32) | lo_);
                        }
                        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)lg[s_t]);
                    }
                    const bool pass = valid && t < (uint32_t)p.T && ((p.pass_mask >> (t & 31u)) & 1u);
                    if (pass && lane == 0) {
                        lg[s_t] = (uint8_t)my_type;
                        lg[s_o] = (uint8_t)p.default_type;
                    }
                    moved = lane == a ? (pass ? 1u : 0u) : moved;
                    adir = (lane == a && aok && kind == SGW_ACTION_MOVE && ndir < 4u) ? ndir : adir;   // movement() turns the agent even if the move fails
                    rew_bits = lane == a ? __float_as_uint((float)val) : rew_bits;
                    tot += val * (double)(p.total_factor - 1);   // the extra add inside act() (agents.py:172) ...
                    tot += val;                                  // ... and Agent.transition's own (agent.py:172)
                    gsync<1>();
                    continue;
                }
            }
            const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)lg[valid ? s_t : (uint32_t)s_o]);
            const bool tok = valid && t < (uint32_t)p.T;
            const uint32_t tl = t & 31u;
            const uint32_t v_lo = (uint32_t)__builtin_amdgcn_readlane((int)vt_lo, (int)tl);
            const uint32_t v_hi = (uint32_t)__builtin_amdgcn_readlane((int)vt_hi, (int)tl);
            const uint32_t v_f = (uint32_t)__builtin_amdgcn_readlane((int)vt_f32, (int)tl);
            const bool pass = tok && ((p.pass_mask >> tl) & 1u);
            if (pass && lane == 0) {
                lg[s_t] = (uint8_t)my_type;
                lg[s_o] = (uint8_t)p.default_type;
            }
            moved = lane == a ? (pass ? 1u : 0u) : moved;
            if constexpr (!TAG) {
                if (tok) tot += __longlong_as_double(((long long)v_hi << 32) | v_lo);   // reward BEFORE the move; float64, agent order
                rew_bits = lane == a ? (tok ? v_f : 0u) : rew_bits;
            } else {
                // ---- TagAgent.act (sorrel/examples/tag/agents.py:84-106), scalar: the four neighbours of the
                // cell the agent now stands on, in Location.adjacent order (up, right, down, left; off-map
                // skipped); an agent that is "it" hands the flag to the first NotIt neighbour.
                gsync<1>();
                pov_type = lane == a ? my_type : pov_type;
                const uint32_t np_a = (uint32_t)__builtin_amdgcn_readlane((int)npos, a);
                const int cy = pass ? (int)(np_a & 0xFFu) : (int)(((uint32_t)s_o - (uint32_t)zoff) / (uint32_t)W);
                const int cx = pass ? (int)((np_a >> 8) & 0xFFu) : (int)(((uint32_t)s_o - (uint32_t)zoff) % (uint32_t)W);
                const int own = zoff + cy * W + cx;
                int dstar = -1;
                if (my_type == p.tag_it) {   // wave-uniform: only the agent that is "it" looks around (one in A; the others skip four LDS round trips)
                    uint32_t nt[4];
                    bool ain[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int ay = cy + (d == 0 ? -1 : d == 2 ? 1 : 0), ax = cx + (d == 1 ? 1 : d == 3 ? -1 : 0);
                        ain[d] = (unsigned)ay < (unsigned)H && (unsigned)ax < (unsigned)W;
                        nt[d] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lg[ain[d] ? zoff + ay * W + ax : own]);
                    }
#pragma unroll
                    for (int d = 3; d >= 0; --d)
                        if (ain[d] && nt[d] == p.tag_notit) dstar = d;
                }
                uint32_t mine_now = my_type;
                if (my_type == p.tag_it && dstar >= 0) {
                    const int ay = cy + (dstar == 0 ? -1 : dstar == 2 ? 1 : 0), ax = cx + (dstar == 1 ? 1 : dstar == 3 ? -1 : 0);
                    if (lane == 0) {
                        lg[own] = (uint8_t)p.tag_notit;
                        lg[zoff + ay * W + ax] = (uint8_t)p.tag_it;
                    }
                    // who stands there: lane b's current position is its start position or, if it moved, its target
                    const uint32_t curpos = moved ? npos : yx;
                    const uint32_t key = (uint32_t)ay | ((uint32_t)ax << 8);
                    atype = (lane < p.A && lane != a && curpos == key) ? p.tag_it : atype;
                    atype = lane == a ? p.tag_notit : atype;
                    mine_now = p.tag_notit;
                }
                const double val = mine_now != p.tag_it ? p.tag_reward : 0.0;
                tot += val;
                rew_bits = lane == a ? __float_as_uint((float)val) : rew_bits;
            }
            if (valid && !tok) st_lane |= SGW_STATUS_BAD_TYPE;
            gsync<1>();
        }

        STAMP(4);   // agent loop done
        if constexpr (kStageAlways) {
            if (write_obs) emit_chunk(ch_a0, p.a1, true);
        } else if (stage && write_obs) {
            gsync<1>();
            const int nd = (p.A * C * VV) >> 2;   // dwords of staged bytes (the host stages only multiples of 4 elements)
            const uint32_t* ob4 = reinterpret_cast<const uint32_t*>(ob);
            if constexpr (ROWS) {
                fast_rows_emit<TL, TC, TR, TH, TW, TAIL>(p, rp, smem, sub, lane);
            } else if (!p.obs_u8) {
                // Non-temporal (streaming) stores: every wave instruction here writes eight whole 128-byte lines that
                // nothing reads again in this launch; keeping them out of the caches leaves those to the grids (134 MB,
                // re-read next turn) and takes config 3 from 167 to 125-132 us.  (The same hint on the per-agent dword
                // stores of the unstaged path, which write partial lines, was measured SLOWER.)
                typedef float vfloat4 __attribute__((ext_vector_type(4)));
                vfloat4* o4 = reinterpret_cast<vfloat4*>(p.obs + turn_obs + env * (int64_t)(p.A * C * VV));
                // (round 3) lane 0 of every store sits on a 128-byte line: an env's block is a whole number of 64-byte half lines
                // (config 3: 9 408 B = 73.5 lines), so every second env starts mid-line, and a streaming store that covers part
                // of a line costs as if ... tools/micro/region_writer.hip: 1 KiB stores shifted by 64 / 32 / 16 bytes write at
                // 4.91 / 4.44 / 4.44 TB/s against 5.48 aligned.  mis = float4s between the line and the env's first element.
                const int mis = (int)((reinterpret_cast<uintptr_t>(o4) >> 4) & 7u);
                for (int i = lane - mis; i < nd; i += 64) {
                    if (i < 0) continue;
                    const uint32_t b = ob4[i];
                    vfloat4 v;
                    v.x = (float)(b & 0xFFu);
                    v.y = (float)((b >> 8) & 0xFFu);
                    v.z = (float)((b >> 16) & 0xFFu);
                    v.w = (float)(b >> 24);
                    __builtin_nontemporal_store(v, &o4[i]);
                }
            } else {
                uint32_t* o1 = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(p.obs) + turn_obs + env * (int64_t)(p.A * C * VV));
                for (int i = lane; i < nd; i += 64) __builtin_nontemporal_store(ob4[i], &o1[i]);   // two whole lines per wave instruction
            }
        }
        if (p.do_move && mine) {      // this turn's rewards (and what TagAgent.pov appends)
            p.rewards[tix * p.ts_rew + env * p.A + lane] = __uint_as_float(rew_bits);
            if (p.state_at_pov) p.state_at_pov[env * p.A + lane] = (uint8_t)pov_type;
        }
        if (tix + 1 < nturns) yx = moved ? npos : yx;   // the next turn starts where this one ended
        }   // turns
        if (dirty) {
            if (!TAG && !RULES && !do_sweep && nturns == 1) {   // (a rollout's earlier turns moved other cells too)
                // a policy-driven phase (no sweep, plain moves): only the movers' two cells changed -- write those bytes,
                // not the whole grid (two movers touching one cell both write its FINAL content: no race)
                if (mine && moved) {
                    uint8_t* g = p.grid + env * p.env_stride;
                    g[oaddr_v] = lg[oaddr_v];
                    g[taddr_v] = lg[taddr_v];
                }
            } else if (sweep_only) {
                // the sweep alone changed the grid (a policy-driven turn's first launch: nobody acts in it): write back only
                // the 16-byte units in which something spawned -- ~4 of config 3's 128 units per env; the sweep-only launch
                // is then a read of the grid plus a few scattered units instead of a read and a full write
                // (run-time shapes: the dword sweep above stored its changed dwords as it went)
                if constexpr (kStatic) {
                    uint4* dst = reinterpret_cast<uint4*>(p.grid + env * p.env_stride);
#pragma unroll
                    for (int k = 0; k < NU; ++k)
                        if (lane + 64 * k < nunits && hits[k]) dst[lane + 64 * k] = lg16[lane + 64 * k];
                }
            } else {
                uint4* dst = reinterpret_cast<uint4*>(p.grid + env * p.env_stride);
#pragma unroll
                for (int k = 0; k < NU; ++k)
                    if (lane + 64 * k < nunits) dst[lane + 64 * k] = lg16[lane + 64 * k];
                if constexpr (RULES || !kStatic)
                    for (int i = 64 * NU + lane; i < nunits; i += 64) dst[i] = lg16[i];
            }
        }
        if (p.do_move) {
            if (mine) {
                reinterpret_cast<uint16_t*>(p.pos)[env * p.A + lane] = (uint16_t)(moved ? npos : yx);
                if (st_lane) atomicOr(p.status, st_lane);
            }
            if (TAG && p.agent_state && lane < p.A) p.agent_state[env * p.A + lane] = (uint8_t)atype;   // a tag can flip any agent
            if (RULES && p.agent_dir && lane < p.A) p.agent_dir[env * p.A + lane] = (uint8_t)adir;
            if (lane == 0) p.total[env] = tot;
        }
        STAMP(5);   // all stores issued
#ifdef SGW_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(6);   // all stores acknowledged
    }
}

template <bool ONEHOT, int TL, int TC, int TR, int TH, int TW, bool TAG = false, bool RULES = false, bool STAGE = false, bool MULTI = false, bool P3 = false, bool I16 = false>
__global__ __launch_bounds__(kBlock, MULTI ? SGW_FAST_MULTI_WAVES : (RULES ? (STAGE ? 7 : SGW_FAST_RULES_PLAIN_WAVES) : 8)) void step_fast(const Params p) {
    step_fast_body<ONEHOT, TL, TC, TR, TH, TW, TAG, RULES, STAGE, MULTI, P3, I16, false>(p, nullptr);
}

// ... on the chunk-staging instances (ROWX; specialised in-process only: the library holds no prebuilt twin)
template <int TL, int TC, int TR, int TH, int TW, bool TAG, bool RULES, bool P3>
__global__ __launch_bounds__(kBlock, RULES ? 7 : 8) void step_fast_rowsx(const Params p, const RowPtrs rp) {
    step_fast_body<true, TL, TC, TR, TH, TW, TAG, RULES, true, false, P3, false, false, true>(p, &rp);
}

// sweep + every agent's window into per-agent rows (sgw_sweep_observe_rows): nobody acts in this launch
template <int TL, int TC, int TR, int TH, int TW, bool TAG = false, bool TAIL = false>
__global__ __launch_bounds__(kBlock, 8) void step_fast_rows(const Params p, const RowPtrs rp) {
    step_fast_body<true, TL, TC, TR, TH, TW, TAG, false, false, false, false, false, true, false, TAIL>(p, &rp);
}

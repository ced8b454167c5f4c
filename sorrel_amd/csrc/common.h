// common.h -- part of the single translation unit sgw.hip (included inside its anonymous namespace).
// Constants, the device tables and launch parameters, the counter RNG, group sync, grid <-> LDS copies, the entity sweeps.
#pragma once


constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kMaxPass = 2;  // window cells per thread held in registers

// ---------------------------------------------------------------- tables
// Constant per-engine tables: built on the host in sgw_create, kept in device
// memory, copied into LDS at the start of every workgroup.
struct DevTables {
    uint32_t thr_lo[SGW_MAX_TYPES];    // spawn threshold, low 32 bits of floor(p * 2^32)
    uint32_t delta[4][SGW_MAX_TYPES];  // one-hot: word c/4 holds 1 << 8*(c%4) for the type's channel c, else 0
    double value[SGW_MAX_TYPES];
    uint8_t spawn_choice[SGW_MAX_TYPES][SGW_MAX_CHOICES];
    uint8_t spawn_count[SGW_MAX_TYPES];
    uint8_t agent_type[SGW_MAX_AGENTS];
    uint8_t dense_choice[SGW_MAX_CHOICES];
    uint8_t layer_fill[8];
    uint8_t layer_border[8];
    uint8_t pad_[8];
    uint8_t rule[SGW_MAX_TYPES];          // SGW_RULE_* per type
    int8_t rule_layer[SGW_MAX_TYPES];     // SGW_RULE_BECOME_IF: layer to test (< 0: always)
    uint8_t rule_become[SGW_MAX_TYPES];
    uint8_t pad2_[SGW_MAX_TYPES];
    uint32_t rule_mask[SGW_MAX_TYPES];
    uint32_t delta3[SGW_MAX_TYPES];      // one-hot, <= 10 channels: ONE word per type, 1 << 3 * c for its channel c (3-bit counters: <= 7 layers)
    double appearance[SGW_MAX_TYPES][SGW_MAX_CHANNELS];  // general (non one-hot) path only
    // integer appearance tables behind SGW_OBS_POST_CLIP255_DIV255 (the reference's RGBObservationSpec: uint8 colours, np.clip(sum, 0, 255) / 255):
    // 16-bit counters, two channels per word (a sum over <= 7 layers of values <= 9 362 cannot carry), and the 256 possible results
    uint32_t delta16[2][SGW_MAX_TYPES];
    float post_lut[256];                                 // (float)(min(k, 255) / 255.0), k = 0 .. 255: what obs_finish returns for an integer sum
};
constexpr int kTabFastBytes = offsetof(DevTables, appearance);
static_assert(kTabFastBytes % 16 == 0, "LDS table block must keep 16-byte alignment");
static_assert(sizeof(DevTables) % 16 == 0, "LDS table block must keep 16-byte alignment");

// What changes from turn to turn of a policy-driven loop, kept in device memory and advanced by the engine's own kernels
// (sgw_turn_begin / sgw_turn_end) instead of arriving as kernel arguments: Environment.turn and the epoch, and for every agent
// the row of its replay ring that this turn's window / action / reward go to (sorrel/buffers.py:46-63: Buffer.add's idx).
struct TurnState {
    uint32_t epoch, turn;                 // turn: turns of this epoch COMPLETED (the turn in flight is turn + 1; sgw_turn_end advances it)
    uint32_t pad_[2];
    int64_t row[SGW_MAX_AGENTS];          // ring row of the turn in flight
    int64_t cap[SGW_MAX_AGENTS];          // rows in that agent's ring (0: the agent keeps no replay rows)
    int64_t step[SGW_MAX_AGENTS];         // rows the ring advances per turn (agents sharing one Buffer: how many share it)
    void* states[SGW_MAX_AGENTS];         // [cap][E][row_elems] of the observation format
    float* rewards[SGW_MAX_AGENTS];       // [cap][E]
    int64_t* actions[SGW_MAX_AGENTS];     // [cap][E]
    float* dones[SGW_MAX_AGENTS];         // [cap][E]: zeroed for the turn's row (done is False inside an epoch); NULL: rows are all zero already
    int64_t row_elems[SGW_MAX_AGENTS];    // elements per env of a states row (>= C*V*V; the tail is the caller's)
    uint64_t eps_thr[SGW_MAX_AGENTS];     // SGW_ACT_QF32: explore when u32 < eps_thr (floor(epsilon * 2^32): 2^32 = always)
};

// per-agent window destinations (sgw_observe_rows / sgw_act): agent a's window of env e starts at p[a] + e * stride elements
struct RowPtrs {
    void* p[SGW_MAX_AGENTS];
    int64_t stride;
    // sgw_act only: where the acting agent's action comes from and where else its outputs go (all optional)
    const void* agent_action;   // [E] of the agent's own actions (the policy's output tensor as it is); NULL: actions[E][A]
    int action_kind;            // SGW_ACT_U8 / _I32 / _I64 / _QF32 (agent_action: [E][nact] action values)
    float* reward_row;          // [E]: a second copy of the rewards (the row of the agent's replay buffer)
    int64_t* action_row;        // [E]: the actions as int64 (the row of the agent's replay buffer)
    const TurnState* ts;        // sgw_turn_act: reward_row / action_row are the acting agent's ring rows of the turn the engine has counted up to
    const TurnState* ets;       // SGW_ACT_QF32: where the exploration reads epsilon, epoch and turn (= ts under the turn protocol)
    int dual;                   // sgw_turn_*_rows: every window (and every repair) is ALSO written to the agent's ring row of the turn in flight
    int rows_mode2;             // ... how those second copies leave (kRowsFlat where the rings allow it, else kRowsRun)
};

struct Params {
    int H, W, L, A, r, V, VV, C, T, nact, zA;
    int cells;      // L*H*W bytes of one env's grid
    int64_t env_stride;  // bytes between envs in HBM (>= cells; multiple of 16 on the vector paths)
    int cells_pad;  // rounded up to 16
    int env_lds;    // LDS bytes per env slice
    int tab_bytes;  // LDS bytes of the table block
    uint32_t default_type, fill_type;
    uint32_t spawn_mask, thr_full_mask, pass_mask;
    uint32_t become_mask;      // types that carry SGW_RULE_BECOME_IF
    uint32_t agent_mask;       // types the agents have
    uint32_t dy_pack, dx_pack;  // 2 bits per action: (d + 1)
    uint32_t fill_delta[4];
    uint32_t fill_delta3;      // the fill entity's word of DevTables::delta3
    uint32_t fill_delta16[2];  // ... of DevTables::delta16
    // single-spawner fast path (exactly one type carries SGW_RULE_SPAWN)
    uint32_t spawn_pat;      // type id replicated in 4 bytes
    uint32_t spawn_thr;      // low 32 bits of floor(p * 2^32)
    uint32_t spawn_full;     // p >= 1
    uint32_t spawn_n;        // number of choices
    uint32_t choice_lo, choice_hi;  // the <= 8 choice type ids, one per byte
    uint32_t seed_lo, seed_hi;
    uint32_t first_env;
    int64_t E;
    uint32_t epoch, turn, flags;
    int a0, a1;
    int do_move;  // 0: observe only
    int obs_post; // SGW_OBS_POST_*
    int obs_u8;   // SGW_OBS_U8: observations are uint8 counts (one-hot specs only)
    int agent_rule;            // SGW_AGENT_RULE_*
    uint32_t tag_it, tag_notit;
    double tag_reward;
    uint8_t* agent_state;      // optional [E][A]: current type of every agent
    uint8_t* agent_dir;        // optional [E][A]: facing (SGW_AGENT_RULE_CLEANUP)
    int has_become;            // some type carries SGW_RULE_BECOME_IF: ordered, layer-by-layer sweep
    uint32_t quiet0, quiet1;   // ordered sweep: dwords equal to one of these hold four cells of a rule-less layer-fill type (or pad bytes): skipped
    uint32_t kind_pack;        // 2 bits per action: SGW_ACTION_*
    int beam_radius;
    uint32_t clean_beam, zap_beam, beam_block_mask;
    int total_factor;
    uint8_t* state_at_pov;     // optional [E][A]: type at observation time
    uint64_t dense_thr;
    int dense_count;
    uint8_t* grid;
    uint8_t* pos;
    uint8_t* actions;
    float* obs;
    float* rewards;
    double* total;
    const DevTables* tab;
    const uint8_t* tmpl;   // reset: the fill + border image of one env (cells_pad bytes, pad = 0xFF)
    int* status;
    int obs_stage;    // step_fast: bytes of the per-wave LDS observation staging area (0: observations go straight to HBM)
    int stage_agents; // step_fast<..., STAGE>: agents whose observations are staged together and leave in one burst
    int obs_next;     // SGW_STEP_OBS_NEXT: write only the observation of agent a1, after the moves of [a0, a1)
    int obs_A, obs_a0; // where agent a's window goes: obs + ((env * obs_A + (a - obs_a0)) * C) * V * V.  (A, 0): the [E][A][C][V][V]
                      // tensor; (1, a1) with SGW_STEP_OBS_NEXT_PACKED: one window per env, [E][C][V][V] (an agent's replay slot)
    int64_t obs_ag;   // step_big, SGW_STEP_OBS_AGENT_MAJOR: elements between two agents' rows of an [A][E][C][V][V] observation tensor (0: [E][A][C][V][V])
    int big_pitch;    // step_big: bytes between grid rows in LDS (W, or W + 16 to spread window rows over the banks)
    int big_stage;    // step_big: bytes of LDS observation staging per wave (0: windows go straight to HBM, a dword store per lane and channel)
    int big_stage_off; // ... and where the first wave's area starts (behind the grid image)
    // step_big<..., WALK>: the envs behind every workgroup's static share are handed out through a counter in device memory
    // (whichever workgroup is free takes the next one: the XCDs of a chip do not run at the same speed)
    uint32_t* walk_ctr;   // the counter (0 between launches: the workgroup that takes the last number resets it)
    int walk_word;        // LDS byte offset of the word the next env's index is passed through
    int walk_static;      // envs per workgroup that are assigned statically (blockIdx + k * gridDim, k < walk_static)
    int single_spawner;   // at most one type carries SGW_RULE_SPAWN: the byte-parallel sweep applies
    int rows_mode;        // phase_rows / observe_rows: how the staged windows leave (kRowsFlat / kRowsPair / kRowsSingle, phase.h)
    int rows_by_agent;    // observe_rows: a wave carries consecutive envs of ONE agent (per-agent destinations) instead of consecutive agents of an env
    int onehot;           // every appearance row is a one-hot (or zero) vector and there is no post-processing: byte counters apply
    // Row tails (sgw_bind_row_tail): what an agent's pov() appends to its flattened window, written behind the window in its row by
    // sgw_observe_rows (and kept current by sgw_act): Tag's "it" flag (examples/tag/agents.py:57-65), Cleanup's positional code
    // (examples/cleanup/agents.py:52-60, observation/embedding.py:8-44)
    int tail_kind, tail_len;
    const float* tail_table;   // SGW_TAIL_POSITION_TABLE: [H][W][tail_len]
    // Device-side turn state (sgw_turn_*): when set, the kernels take epoch and turn from here instead of the two fields above, so
    // that a launch recorded once (a hipGraph of a whole policy turn) plays the turn the engine has counted up to
    const TurnState* ts;
    // sgw_rollout: `nturns` whole turns in ONE launch (the env's grid stays in LDS from turn to turn); turn t of the call
    // writes its observations / actions / rewards `t * ts_*` elements further on (0 = every turn overwrites the same tensors)
    uint32_t nturns;
    int64_t ts_obs, ts_act, ts_rew;
};

// ---------------------------------------------------------------- RNG
struct U4 {
    uint32_t x, y, z, w;
};

// a ^ b ^ k in one VALU op (gfx950: v_bitop3_b32 with truth table 0x96; there is no v_xor3_b32 on gfx9).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t k) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
}

// Philox-4x32-10 (Salmon et al., SC'11); key = (k0, k1) wave-uniform.
// OWN_KEYS: the block walks its own key schedule (two s_add per round).  Left to itself the compiler computes the twenty round
// keys once and keeps them in SGPRs across all the blocks of a kernel, and spills other scalars into vector lanes for it
// (v_writelane / v_readlane are vector instructions, and these kernels are bound by those): the packed small-world kernels gain
// 5-9 % from own keys, the run-time-shape wave-per-env kernels 1.5 %; the compile-time-shape instances of step_fast (the
// headline: 22 -> 6 spilled scalars, but 772 instead of 673 scalar instructions per wave) LOSE 0.5-1.3 % and keep the shared keys.
template <bool OWN_KEYS = true>
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#ifndef SGW_DIAG_SHARED_KEYS
    if constexpr (OWN_KEYS) asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// Keeps the compiler from hoisting the env-invariant first Philox round of every
// (lane, block) pair out of the persistent env loop (that costs ~2 VGPRs per block).
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ uint32_t word_of(const U4& v, int i) {
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// RGBObservationSpec: np.clip(obs, 0, 255) / 255 on the float64 layer sum (observation_spec.py:483)
__device__ __forceinline__ float obs_finish(double acc, int post) {
    if (post == SGW_OBS_POST_CLIP255_DIV255) acc = fmin(fmax(acc, 0.0), 255.0) / 255.0;
    return (float)acc;
}

// ---------------------------------------------------------------- group sync
// WPE == 1: the group is one wavefront.  DS instructions of a wave execute in
// issue order, so a compiler-level fence is all that is needed.
template <int WPE>
__device__ __forceinline__ void gsync() {
    if constexpr (WPE == 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// ---------------------------------------------------------------- grid <-> LDS
template <int G>
__device__ __forceinline__ void load_grid(const Params& p, const uint8_t* __restrict__ src,
                                          uint8_t* lds, int gtid) {
    if ((p.env_stride & 15) == 0 && p.env_stride >= p.cells_pad) {
        // whole 16-byte units, the env's pad bytes included (a ragged world in a padded stride: the bytes past the last
        // cell are not cells -- no type, no RNG index -- and are masked to 0xFF in LDS)
        const uint4* s = reinterpret_cast<const uint4*>(src);
        uint4* d = reinterpret_cast<uint4*>(lds);
        const int nu = p.cells_pad >> 4;
        for (int i = gtid; i < nu; i += G) {
            uint4 v = s[i];
            if (i == nu - 1 && (p.cells & 15)) {
                const int tail = p.cells & 15;
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int keep = tail - 4 * q;   // valid bytes in this dword
                    if (keep <= 0) w[q] = 0xFFFFFFFFu;
                    else if (keep < 4) w[q] |= 0xFFFFFFFFu << (8 * keep);
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            d[i] = v;
        }
    } else if ((p.cells & 3) == 0 && (p.env_stride & 3) == 0) {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
        uint32_t* d = reinterpret_cast<uint32_t*>(lds);
        for (int i = gtid; i < (p.cells >> 2); i += G) d[i] = s[i];
    } else {
        for (int i = gtid; i < p.cells_pad; i += G) lds[i] = i < p.cells ? src[i] : (uint8_t)0xFF;
    }
}

template <int G>
__device__ __forceinline__ void store_grid(const Params& p, uint8_t* __restrict__ dst,
                                           const uint8_t* lds, int gtid) {
    if ((p.env_stride & 15) == 0 && p.env_stride >= p.cells_pad) {
        uint4* d = reinterpret_cast<uint4*>(dst);
        const uint4* s = reinterpret_cast<const uint4*>(lds);
        for (int i = gtid; i < (p.cells_pad >> 4); i += G) d[i] = s[i];   // the pad bytes of the stride are nobody's cells
    } else if ((p.cells & 3) == 0 && (p.env_stride & 3) == 0) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        const uint32_t* s = reinterpret_cast<const uint32_t*>(lds);
        for (int i = gtid; i < (p.cells >> 2); i += G) d[i] = s[i];
    } else {
        for (int i = gtid; i < p.cells; i += G) dst[i] = lds[i];
    }
}


// ---------------------------------------------------------------- sweep
// At most one spawning type (every Treasurehunt-shaped world): byte-parallel match of the spawner id, one Philox block
// per dword that holds a spawner, thresholds and choices from scalar registers instead of per-byte table reads.
template <int G>
__device__ __forceinline__ void sweep_single(const Params& p, uint8_t* lds_grid, uint32_t env_id, int gtid, uint32_t turn, const uint32_t ep4,
                                             uint32_t* through = nullptr,    // through: the env's grid in global memory -- changed dwords go there too
                                             const int d0 = 0) {             // first dword swept (step_fast: the tail behind the rounds its register sweep covers)
    uint32_t* g32 = reinterpret_cast<uint32_t*>(lds_grid);
    const int ndw = (p.cells + 3) >> 2;
    for (int d = d0 + gtid; d < ndw; d += G) {
        // the spawner test: a byte compare per cell (one v_cmp with a byte select each), combined with the draws as lane masks
        const uint32_t dv = g32[d], pat = p.spawn_pat & 0xFFu;
        const bool m0 = (dv & 0xFFu) == pat, m1 = ((dv >> 8) & 0xFFu) == pat, m2 = ((dv >> 16) & 0xFFu) == pat, m3 = (dv >> 24) == pat;
        if (!(m0 | m1 | m2 | m3)) continue;
        const U4 w = philox4x32_10((uint32_t)d, turn, env_id, ep4 | SGW_STREAM_SPAWN, p.seed_lo, p.seed_hi);
        const bool f = p.spawn_full != 0;
        uint32_t hits = 0;
        hits |= (m0 && (f || w.x < p.spawn_thr)) ? 1u : 0u;
        hits |= (m1 && (f || w.y < p.spawn_thr)) ? 2u : 0u;
        hits |= (m2 && (f || w.z < p.spawn_thr)) ? 4u : 0u;
        hits |= (m3 && (f || w.w < p.spawn_thr)) ? 8u : 0u;
        if (hits == 0) continue;
        const U4 k = philox4x32_10((uint32_t)d, turn, env_id, ep4 | SGW_STREAM_SPAWN_KIND, p.seed_lo, p.seed_hi);
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if ((hits >> b) & 1u) {
                const uint32_t pick = __umulhi(word_of(k, b), p.spawn_n);
                lds_grid[4 * d + b] = (uint8_t)(((pick < 4 ? p.choice_lo : p.choice_hi) >> (8 * (pick & 3u))) & 0xFFu);
            }
        if (through) through[d] = g32[d];
    }
}

// Entity transitions (reference: environment.py:88-91).  RNG index of a cell ==
// its byte offset in the [L][H][W] slice, so one LDS dword == one Philox block.
template <int G>
__device__ __forceinline__ void sweep(const Params& p, const DevTables* tab, uint8_t* lds_grid,
                                      uint32_t env_id, int gtid, uint32_t turn, const uint32_t ep4) {
    uint32_t* g32 = reinterpret_cast<uint32_t*>(lds_grid);
    const int ndw = (p.cells + 3) >> 2;
    const uint32_t c3 = ep4 | SGW_STREAM_SPAWN;
    for (int d = gtid; d < ndw; d += G) {
        uint32_t v = g32[d];
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t t = (v >> (8 * b)) & 0xFFu;
            const uint32_t is = (t < SGW_MAX_TYPES) ? ((p.spawn_mask >> t) & 1u) : 0u;
            m |= is << b;
        }
        if (m == 0) continue;
        const U4 w = philox4x32_10((uint32_t)d, turn, env_id, c3, p.seed_lo, p.seed_hi);
        uint32_t hits = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if ((m >> b) & 1u) {
                const uint32_t t = (v >> (8 * b)) & 31u;
                const bool hit = ((p.thr_full_mask >> t) & 1u) || (word_of(w, b) < tab->thr_lo[t]);
                hits |= (hit ? 1u : 0u) << b;
            }
        }
        if (hits == 0) continue;
        const U4 k = philox4x32_10((uint32_t)d, turn, env_id, ep4 | SGW_STREAM_SPAWN_KIND,
                                   p.seed_lo, p.seed_hi);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if ((hits >> b) & 1u) {
                const uint32_t t = (v >> (8 * b)) & 31u;
                const uint32_t n = tab->spawn_count[t];
                const uint32_t pick = (uint32_t)(((uint64_t)word_of(k, b) * n) >> 32);
                const uint32_t nt = tab->spawn_choice[t][pick];
                v = (v & ~(0xFFu << (8 * b))) | (nt << (8 * b));
            }
        }
        g32[d] = v;
    }
}

// Ordered sweep for rule sets with cross-layer conditions (SGW_RULE_BECOME_IF, e.g. Cleanup): the
// reference visits cells in (y, x, z) order over a LIVE view, so within a column a lower layer has
// already transitioned when a higher one is visited and a higher one has not when a lower one is.
// Columns never read each other and rules write their own cell only, so: one pass per layer, lower layers first, all
// cells of the layer in parallel -- one dword (four cells = one Philox block) per lane and step.  `Tab` is DevTables
// (generic kernels: the table block in LDS) or RuleLds (the RULES variant of step_fast: its wave-private copy).
// (Round 3: the generic kernels walked this byte by byte with a Philox block per CELL; Cleanup 48x48x3: 30 of 101 us.)
template <int WPE, int G, typename Tab>
__device__ __forceinline__ void sweep_ordered(const Params& p, const Tab* rt, uint8_t* lg, const uint32_t env_id, const int gtid,
                                              const uint32_t turn, const uint32_t ep4, const int L, const int HW) {
    const uint32_t* lg32 = reinterpret_cast<const uint32_t*>(lg);
    for (int z = 0; z < L; ++z) {
        const int lo = z * HW, hi = lo + HW;
        for (int d = (lo >> 2) + gtid; d < ((hi + 3) >> 2); d += G) {   // one dword = four cells = one Philox block
            const uint32_t word = lg32[d];
            if (word == p.quiet0 || word == p.quiet1) continue;       // four cells of a fill type without a rule (most of an agent / beam layer)
            uint32_t tj[4];
            bool spj[4], bcj[4];
            bool any_sp = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = 4 * d + j;
                tj[j] = (word >> (8 * j)) & 0xFFu;
                const bool in = off >= lo && off < hi && tj[j] < (uint32_t)SGW_MAX_TYPES;   // (a dword may straddle two layers)
                spj[j] = in && ((p.spawn_mask >> (tj[j] & 31u)) & 1u);
                bcj[j] = in && ((p.become_mask >> (tj[j] & 31u)) & 1u);
                any_sp = any_sp || spj[j];
            }
            if (any_sp) {
                const U4 w = philox4x32_10(opaque((uint32_t)d), turn, env_id, ep4 | SGW_STREAM_SPAWN, p.seed_lo, p.seed_hi);
                uint32_t hit = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (spj[j] && (((p.thr_full_mask >> tj[j]) & 1u) || word_of(w, j) < rt->thr_lo[tj[j]])) hit |= 1u << j;
                if (hit) {   // rare: what spawns
                    const U4 kw = philox4x32_10(opaque((uint32_t)d), turn, env_id, ep4 | SGW_STREAM_SPAWN_KIND, p.seed_lo, p.seed_hi);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((hit >> j) & 1u)
                            lg[4 * d + j] = rt->spawn_choice[tj[j]][__umulhi(word_of(kw, j), (uint32_t)rt->spawn_count[tj[j]])];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (bcj[j]) {
                    const int zl = rt->rule_layer[tj[j]];
                    const bool fire = zl < 0 || ((rt->rule_mask[tj[j]] >> (lg[zl * HW + (4 * d + j - lo)] & 31u)) & 1u);
                    if (fire) lg[4 * d + j] = rt->rule_become[tj[j]];
                }
        }
        gsync<WPE>();
    }
}

// Rule tables of the RULES variant of step_fast, copied per wave into LDS: three contiguous pieces of DevTables.
struct RuleLds {
    uint32_t thr_lo[SGW_MAX_TYPES];
    uint8_t spawn_choice[SGW_MAX_TYPES][SGW_MAX_CHOICES];
    uint8_t spawn_count[SGW_MAX_TYPES];
    uint8_t rule[SGW_MAX_TYPES];
    int8_t rule_layer[SGW_MAX_TYPES];
    uint8_t rule_become[SGW_MAX_TYPES];
    uint8_t pad2_[SGW_MAX_TYPES];
    uint32_t rule_mask[SGW_MAX_TYPES];
};
constexpr int kRuleLds = (int)sizeof(RuleLds);
static_assert(kRuleLds == 672 && kRuleLds % 16 == 0, "RuleLds mirrors three pieces of DevTables");
static_assert(offsetof(DevTables, spawn_count) == offsetof(DevTables, spawn_choice) + SGW_MAX_TYPES * SGW_MAX_CHOICES, "piece B is contiguous");
static_assert(offsetof(DevTables, rule_mask) == offsetof(DevTables, rule) + 4 * SGW_MAX_TYPES, "piece C is contiguous");
static_assert(SGW_MAX_CHOICES == 8, "RuleLds copy assumes 8 choices");


// sgw.hip -- MI355X (gfx950 / CDNA4) batched gridworld step + observation engine.
//
// Hand-written HIP behind the C ABI of include/sgw.h.  One thread GROUP owns one environment for a whole take_turn
// (or, through sgw_rollout, for T turns): the env's grid (uint8 type ids, [L][H][W]) is staged once into LDS with 16-byte
// loads, the entity sweep and all sequential agent phases run against LDS, observation windows are gathered from LDS
// (lane = window cell) and the grid is written back once.  Integer / indexing work only: no MFMA; the bound is HBM
// bandwidth for the big shapes (observation stores dominate) and instruction issue for the small ones.
//
// Kernels, in file order:
//   step_kernel<G, ONEHOT, L, C, RULE>   every shape and rule; G = lanes per env: 256 (a workgroup per env, worlds above
//                                        4 KiB), 64 (a wave per env), or 32 / 16 -- two / four SMALL envs share a wave and
//                                        its instruction stream (what 10x10 ... 24x24 worlds of large batches run on);
//                                        all per-env state in the group's LDS slice, no cross-lane instruction; turn loop
//                                        for sgw_rollout built in
//   step_fast<ONEHOT, L, C, r, H, W, TAG, RULES, STAGE, MULTI>
//                                        a wave per env, worlds <= 4 KiB: register sweep, per-lane move inputs + scalar
//                                        move resolution, compile-time window geometry for the BASELINE shapes; one-hot
//                                        observations staged as bytes in LDS and emitted as one burst of streaming
//                                        16-byte stores (fixed shapes: whole env; STAGE: chunks of agents, any alignment);
//                                        MULTI = the turn loop of sgw_rollout
//   step_big<ONEHOT, L, C, r, MULTI>     a 512-thread workgroup per env, worlds above 4 KiB (config 5): padded LDS row
//                                        pitch, moves resolved in registers by wave 0 (only interfering agents are walked),
//                                        observations rendered by all waves from the post-move grid with later moves
//                                        undone in registers
//   phase_kernel<ONEHOT>                 one policy-driven phase of a world above 4 KiB without staging the env
//   reset_kernel, random_actions_kernel, init_agent_state_kernel, reduce_stage1/2
// then the host side: validation, table building, kernel selection (sgw_create), the launchers.
//
// Semantics follow the reference Python step loop bit for bit; see include/sgw.h for the reference file:line each entry
// point replaces and oracle/gridstep_oracle.py for the line-by-line CPU restatement the kernels are tested against
// (the product never calls it).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/sgw.h"

namespace {

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
    double appearance[SGW_MAX_TYPES][SGW_MAX_CHANNELS];  // general (non one-hot) path only
};
constexpr int kTabFastBytes = offsetof(DevTables, appearance);
static_assert(kTabFastBytes % 16 == 0, "LDS table block must keep 16-byte alignment");
static_assert(sizeof(DevTables) % 16 == 0, "LDS table block must keep 16-byte alignment");

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
    int* status;
    int obs_stage;    // step_fast: bytes of the per-wave LDS observation staging area (0: observations go straight to HBM)
    int stage_agents; // step_fast<..., STAGE>: agents whose observations are staged together and leave in one burst
    int obs_next;     // SGW_STEP_OBS_NEXT: write only the observation of agent a1, after the moves of [a0, a1)
    int big_pitch;    // step_big: bytes between grid rows in LDS (W, or W + 16 to spread window rows over the banks)
    int single_spawner;   // at most one type carries SGW_RULE_SPAWN: the byte-parallel sweep applies
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
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
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

__device__ __forceinline__ uint32_t match_bytes(uint32_t v, uint32_t pat) {
    // 0x80 in every byte of v that equals the corresponding byte of pat (exact, no carries between bytes)
    const uint32_t x = v ^ pat;
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

// ---------------------------------------------------------------- sweep
// At most one spawning type (every Treasurehunt-shaped world): byte-parallel match of the spawner id, one Philox block
// per dword that holds a spawner, thresholds and choices from scalar registers instead of per-byte table reads.
template <int G>
__device__ __forceinline__ void sweep_single(const Params& p, uint8_t* lds_grid, uint32_t env_id, int gtid, uint32_t turn) {
    uint32_t* g32 = reinterpret_cast<uint32_t*>(lds_grid);
    const int ndw = (p.cells + 3) >> 2;
    for (int d = gtid; d < ndw; d += G) {
        const uint32_t m = match_bytes(g32[d], p.spawn_pat);
        if (m == 0) continue;
        const U4 w = philox4x32_10((uint32_t)d, turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN, p.seed_lo, p.seed_hi);
        const bool f = p.spawn_full != 0;
        uint32_t hits = 0;
        hits |= ((m & 0x80u) && (f || w.x < p.spawn_thr)) ? 1u : 0u;
        hits |= ((m & 0x8000u) && (f || w.y < p.spawn_thr)) ? 2u : 0u;
        hits |= ((m & 0x800000u) && (f || w.z < p.spawn_thr)) ? 4u : 0u;
        hits |= ((m & 0x80000000u) && (f || w.w < p.spawn_thr)) ? 8u : 0u;
        if (hits == 0) continue;
        const U4 k = philox4x32_10((uint32_t)d, turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN_KIND, p.seed_lo, p.seed_hi);
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if ((hits >> b) & 1u) {
                const uint32_t pick = __umulhi(word_of(k, b), p.spawn_n);
                lds_grid[4 * d + b] = (uint8_t)(((pick < 4 ? p.choice_lo : p.choice_hi) >> (8 * (pick & 3u))) & 0xFFu);
            }
    }
}

// Entity transitions (reference: environment.py:88-91).  RNG index of a cell ==
// its byte offset in the [L][H][W] slice, so one LDS dword == one Philox block.
template <int G>
__device__ __forceinline__ void sweep(const Params& p, const DevTables* tab, uint8_t* lds_grid,
                                      uint32_t env_id, int gtid, uint32_t turn) {
    uint32_t* g32 = reinterpret_cast<uint32_t*>(lds_grid);
    const int ndw = (p.cells + 3) >> 2;
    const uint32_t c3 = (p.epoch << 4) | SGW_STREAM_SPAWN;
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
        const U4 k = philox4x32_10((uint32_t)d, turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN_KIND,
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
// Columns never read each other, so: one pass per layer, all cells of the layer in parallel.
template <int WPE, int G>
__device__ __forceinline__ void sweep_ordered(const Params& p, const DevTables* tab, uint8_t* lg, uint32_t env_id, int gtid, uint32_t turn) {
    const int HW = p.H * p.W;
    for (int z = 0; z < p.L; ++z) {
        for (int cidx = gtid; cidx < HW; cidx += G) {
            const int off = z * HW + cidx;
            const uint32_t t = lg[off];
            if (t >= SGW_MAX_TYPES) continue;
            const uint32_t rule = tab->rule[t];
            if (rule == SGW_RULE_BECOME_IF) {
                const int zl = tab->rule_layer[t];
                const bool fire = zl < 0 || ((tab->rule_mask[t] >> (lg[zl * HW + cidx] & 31u)) & 1u);
                if (fire) lg[off] = tab->rule_become[t];
            } else if (rule == SGW_RULE_SPAWN) {
                const U4 w = philox4x32_10((uint32_t)off >> 2, turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN, p.seed_lo, p.seed_hi);
                if (((p.thr_full_mask >> t) & 1u) || word_of(w, off & 3) < tab->thr_lo[t]) {
                    const U4 k = philox4x32_10((uint32_t)off >> 2, turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN_KIND, p.seed_lo, p.seed_hi);
                    lg[off] = tab->spawn_choice[t][__umulhi(word_of(k, off & 3), (uint32_t)tab->spawn_count[t])];
                }
            }
        }
        gsync<WPE>();
    }
}

// ---------------------------------------------------------------- step kernel
// Per-env LDS slice: [grid cells_pad][pos 2*64][act 64][rew f32 x64]
constexpr int kPosOff = 0;
constexpr int kActOff = 2 * SGW_MAX_AGENTS;
constexpr int kRewOff = kActOff + SGW_MAX_AGENTS;
constexpr int kTypeOff = kRewOff + 4 * SGW_MAX_AGENTS;   // current type of each agent
constexpr int kPovOff = kTypeOff + SGW_MAX_AGENTS;      // its type when it observed
constexpr int kDirOff = kPovOff + SGW_MAX_AGENTS;       // its facing (Cleanup)
constexpr int kAgentLds = kDirOff + SGW_MAX_AGENTS;     // 640 bytes, multiple of 16

#ifndef SGW_GENERIC_WAVES
#define SGW_GENERIC_WAVES 6
#endif
// G = threads per environment: 256 (a workgroup per env, worlds above 4 KiB), 64 (a wave per env) or, for small worlds,
// 32 / 16 lanes of a wave -- two or four envs share a wave and its instruction stream.  The kernel keeps every piece of
// per-env state in the group's LDS slice and uses no cross-lane instruction, so a sub-wave group needs nothing but the
// wave-level ordering of DS instructions; what it buys is that the per-env instruction count, which bounds small worlds
// (a 21x21x2 world keeps 29 of 64 lanes busy in the sweep and 25 in the window gather), is shared by 2 or 4 envs.
template <int G, bool ONEHOT, int TL = 0, int TC = 0, int RULE = SGW_AGENT_RULE_MOVE>
__global__ __launch_bounds__(kBlock, SGW_GENERIC_WAVES) void step_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int WPE = G <= kWave ? 1 : G / kWave;   // waves that must synchronise
    constexpr int EPB = kBlock / G;    // envs per workgroup
    const int tid = threadIdx.x;
    const int sub = tid / G;
    const int gtid = tid - sub * G;

    // constant tables -> LDS (once per workgroup)
    {
        const uint4* s = reinterpret_cast<const uint4*>(p.tab);
        uint4* d = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < (p.tab_bytes >> 4); i += kBlock) d[i] = s[i];
    }
    __syncthreads();
    const DevTables* tab = reinterpret_cast<const DevTables*>(smem);
    uint8_t* slice = smem + p.tab_bytes + sub * p.env_lds;
    uint8_t* lg = slice;                              // grid
    uint8_t* s_pos = slice + p.cells_pad + kPosOff;   // [A][2]
    uint8_t* s_act = slice + p.cells_pad + kActOff;   // [A]
    float* s_rew = reinterpret_cast<float*>(slice + p.cells_pad + kRewOff);
    uint8_t* s_type = slice + p.cells_pad + kTypeOff;   // [A] current entity type of each agent
    uint8_t* s_pov = slice + p.cells_pad + kPovOff;     // [A] its type when it observed
    uint8_t* s_dir = slice + p.cells_pad + kDirOff;     // [A] its facing

    // window cell(s) this thread renders: fixed for the whole kernel
    int wi[kMaxPass], wj[kMaxPass];
#pragma unroll
    for (int k = 0; k < kMaxPass; ++k) {
        const int w = gtid + k * G;
        wi[k] = w / p.V;
        wj[k] = w - wi[k] * p.V;
    }
    const bool write_obs = !(p.flags & SGW_STEP_NO_OBS);
    const bool dirty = (p.flags & SGW_STEP_SWEEP) || (p.do_move && p.a1 > p.a0);
    const int zoff = p.zA * p.H * p.W;
    const int HW = p.H * p.W;

    // one env per group and launch (no persistent loop: nothing stays live from one env to the next, and the
    // dispatcher balances the workgroups)
    const int64_t env = (int64_t)blockIdx.x * EPB + sub;
    if (env < p.E) {
        const uint32_t env_id = p.first_env + (uint32_t)env;
        uint8_t* ggrid = p.grid + env * p.env_stride;
        load_grid<G>(p, ggrid, lg, gtid);
        double tot = 0.0;
        if (gtid == 0 && p.do_move) tot = p.total[env];
        uint32_t yx0 = 0;                     // this thread's agent: position at the start of the call
        if (gtid < p.A) {
            uint16_t yx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + gtid];
            if ((yx & 0xFF) >= p.H || (yx >> 8) >= p.W) {   // garbage in: stay inside this env's LDS slice, and say so
                yx = 0;
                atomicOr(p.status, SGW_STATUS_BAD_POS);
            }
            yx0 = yx;
            reinterpret_cast<uint16_t*>(s_pos)[gtid] = yx;
            s_type[gtid] = p.agent_state ? p.agent_state[env * p.A + gtid] : tab->agent_type[gtid];
            s_dir[gtid] = p.agent_dir ? p.agent_dir[env * p.A + gtid] : (uint8_t)2;
        }
        int st_bits = 0;
        // sgw_rollout: nturns whole turns on the LDS-resident env (nturns == 1: an ordinary sgw_step / sgw_observe)
        for (uint32_t tix = 0; tix < p.nturns; ++tix) {
        const uint32_t turn = p.turn + tix;
        if (gtid < p.A && p.do_move && gtid >= p.a0 && gtid < p.a1) {
            uint8_t* acts = p.actions + tix * p.ts_act;
            uint32_t act;
            if (p.flags & SGW_STEP_RANDOM_ACTIONS) {
                const U4 w = philox4x32_10((uint32_t)gtid >> 2, turn, env_id,
                                           (p.epoch << 4) | SGW_STREAM_ACTION, p.seed_lo, p.seed_hi);
                act = (uint32_t)(((uint64_t)word_of(w, gtid & 3) * (uint32_t)p.nact) >> 32);
                acts[env * p.A + gtid] = (uint8_t)act;
            } else {
                act = acts[env * p.A + gtid];
            }
            s_act[gtid] = (uint8_t)act;
        }
        gsync<WPE>();
        if (p.flags & SGW_STEP_SWEEP) {
            if (p.has_become) {
                sweep_ordered<WPE, G>(p, tab, lg, env_id, gtid, turn);
            } else {
                if (p.single_spawner) sweep_single<G>(p, lg, env_id, gtid, turn);
                else sweep<G>(p, tab, lg, env_id, gtid, turn);
                gsync<WPE>();
            }
        }

        const int a_end = (p.obs_next && p.a1 < p.A) ? p.a1 + 1 : p.a1;   // OBS_NEXT: one extra, observe-only iteration
        for (int a = p.a0; a < a_end; ++a) {
            const int y = s_pos[2 * a], x = s_pos[2 * a + 1];
            // ---- pov: egocentric window (visual_field.py:9-101)
            if (p.obs_next ? a == p.a1 : write_obs) {
                float* obase = p.obs + tix * p.ts_obs + ((env * p.A + a) * (int64_t)p.C) * p.VV;
                auto render = [&](const int w, const int i, const int j) {
                    const int gy = y - p.r + i, gx = x - p.r + j;
                    const bool inb = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                    const int off = gy * p.W + gx;
                    float* o = obase + w;
                    if constexpr (ONEHOT) {
                        constexpr int NWq = TC ? (TC + 3) / 4 : 4;     // counter words (static for the common channel counts)
                        const int Cn = TC ? TC : p.C, Ln = TL ? TL : p.L;
                        uint32_t cnt[NWq];
#pragma unroll
                        for (int q = 0; q < NWq; ++q) cnt[q] = 0u;
                        const int nw = (Cn + 3) >> 2;
                        if (inb) {
#pragma unroll
                            for (int z = 0; z < (TL ? TL : 1); ++z) {
                                const uint32_t t = lg[z * HW + off] & 31u;
#pragma unroll
                                for (int q = 0; q < NWq; ++q)
                                    if (q < nw) cnt[q] += tab->delta[q][t];
                            }
                            if constexpr (TL == 0) {
                                for (int z = 1; z < Ln; ++z) {
                                    const uint32_t t = lg[z * HW + off] & 31u;
#pragma unroll
                                    for (int q = 0; q < NWq; ++q)
                                        if (q < nw) cnt[q] += tab->delta[q][t];
                                }
                            }
                        } else {   // fill entity's appearance, once (visual_field.py:89-94)
#pragma unroll
                            for (int q = 0; q < NWq; ++q) cnt[q] = p.fill_delta[q];
                        }
#pragma unroll
                        for (int q = 0; q < NWq; ++q) {
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const int c = 4 * q + b;
                                if (c < Cn) {
                                    const uint32_t v = (cnt[q] >> (8 * b)) & 0xFFu;
                                    if (p.obs_u8) reinterpret_cast<uint8_t*>(p.obs)[(o - p.obs) + c * p.VV] = (uint8_t)v;
                                    else o[c * p.VV] = (float)v;
                                }
                            }
                        }
                    } else {
                        for (int c = 0; c < p.C; ++c) {
                            double acc;
                            if (inb) {   // np.sum over layers: left to right, float64 (visual_field.py:51)
                                acc = tab->appearance[lg[off] & 31u][c];
                                for (int z = 1; z < p.L; ++z) acc += tab->appearance[lg[z * HW + off] & 31u][c];
                            } else {
                                acc = tab->appearance[p.fill_type][c];
                            }
                            o[c * p.VV] = obs_finish(acc, p.obs_post);
                        }
                    }
                };
#pragma unroll
                for (int k = 0; k < kMaxPass; ++k) {
                    const int w = gtid + k * G;
                    if (w < p.VV) render(w, wi[k], wj[k]);
                }
                {   // further passes (small groups, wide windows): (i, j) advance by G cells, no division
                    int i = wi[kMaxPass - 1], j = wj[kMaxPass - 1];
                    for (int w = gtid + kMaxPass * G; w < p.VV; w += G) {
                        j += G;
                        while (j >= p.V) { j -= p.V; ++i; }
                        render(w, i, j);
                    }
                }
            }
            if (!p.do_move || a >= p.a1) continue;
            if constexpr (RULE == SGW_AGENT_RULE_CLEANUP) {
                // ---- CleanupAgent.act (sorrel/examples/cleanup/agents.py:146-177).  Every thread evaluates the
                // same LDS bytes, so all control flow here is uniform; single threads do the writes.
                const uint32_t act = s_act[a];
                const uint32_t my_type = s_type[a];
                const bool act_ok = act < (uint32_t)p.nact;
                const uint32_t kind = act_ok ? (p.kind_pack >> (2 * act)) & 3u : 0u;
                const int dy = (act_ok && kind == SGW_ACTION_MOVE) ? (int)((p.dy_pack >> (2 * act)) & 3u) - 1 : 0;
                const int dx = (act_ok && kind == SGW_ACTION_MOVE) ? (int)((p.dx_pack >> (2 * act)) & 3u) - 1 : 0;
                const int ny = y + dy, nx = x + dx;
                const uint32_t facing = s_dir[a] & 3u;
                gsync<WPE>();
                if (act_ok && kind != SGW_ACTION_MOVE && p.zA + 1 < p.L && gtid < 3 * p.beam_radius) {
                    // beam cells on the layer above: 1..R ahead; 0..R-1 ahead of the right / left neighbours
                    const int arm = gtid / p.beam_radius, i = gtid - arm * p.beam_radius;
                    const int fy = facing == 0 ? -1 : facing == 2 ? 1 : 0, fx = facing == 1 ? 1 : facing == 3 ? -1 : 0;
                    const int ry = facing == 1 ? 1 : facing == 3 ? -1 : 0, rx = facing == 0 ? 1 : facing == 2 ? -1 : 0;
                    const int step = arm == 0 ? i + 1 : i, side = arm == 0 ? 0 : (arm == 1 ? 1 : -1);
                    const int by = y + side * ry + step * fy, bx = x + side * rx + step * fx;
                    if ((unsigned)by < (unsigned)p.H && (unsigned)bx < (unsigned)p.W) {
                        const int boff = (p.zA + 1) * HW + by * p.W + bx;
                        if (!((p.beam_block_mask >> (lg[boff] & 31u)) & 1u))
                            lg[boff] = (uint8_t)(kind == SGW_ACTION_CLEAN ? p.clean_beam : p.zap_beam);
                    }
                }
                gsync<WPE>();
                const bool inb = act_ok && (unsigned)ny < (unsigned)p.H && (unsigned)nx < (unsigned)p.W;
                double val = 0.0;
                uint32_t t = 0xFFu;
                if (inb) {
                    for (int zl = 0; zl < p.L; ++zl) val += tab->value[lg[zl * HW + ny * p.W + nx] & 31u];   // all layers, BEFORE the move
                    t = lg[zoff + ny * p.W + nx];
                }
                const bool pass = inb && t < (uint32_t)p.T && ((p.pass_mask >> (t & 31u)) & 1u);
                gsync<WPE>();
                if (gtid == 0) {
                    s_pov[a] = (uint8_t)my_type;
                    if (act_ok && kind == SGW_ACTION_MOVE) {            // movement() turns the agent even if the move fails
                        if (dy == -1 && dx == 0) s_dir[a] = 0;
                        else if (dy == 1 && dx == 0) s_dir[a] = 2;
                        else if (dy == 0 && dx == -1) s_dir[a] = 3;
                        else if (dy == 0 && dx == 1) s_dir[a] = 1;
                    }
                    if (pass) {
                        lg[zoff + ny * p.W + nx] = (uint8_t)my_type;
                        lg[zoff + y * p.W + x] = (uint8_t)p.default_type;
                        s_pos[2 * a] = (uint8_t)ny;
                        s_pos[2 * a + 1] = (uint8_t)nx;
                    }
                    s_rew[a] = (float)val;
                    tot += val * (double)(p.total_factor - 1);       // the extra add inside act() (agents.py:172) ...
                    tot += val;                                      // ... and Agent.transition's own (agent.py:172)
                    st_bits |= (!act_ok ? SGW_STATUS_BAD_ACTION : 0) | ((act_ok && !inb) ? SGW_STATUS_OOB_MOVE : 0);
                }
                gsync<WPE>();
                continue;
            }
            // ---- act: MovingAgent.movement / act, Gridworld.move (agent.py:187-225, gridworld.py:95-122)
            const uint32_t act = s_act[a];
            const uint32_t my_type = s_type[a];
            const bool act_ok = act < (uint32_t)p.nact;
            const int dy = act_ok ? (int)((p.dy_pack >> (2 * act)) & 3u) - 1 : 0;
            const int dx = act_ok ? (int)((p.dx_pack >> (2 * act)) & 3u) - 1 : 0;
            const int ty = y + dy, tx = x + dx;
            const bool inb = act_ok && (unsigned)ty < (unsigned)p.H && (unsigned)tx < (unsigned)p.W;
            const int taddr = zoff + ty * p.W + tx;
            const int oaddr = zoff + y * p.W + x;
            const uint32_t t = inb ? lg[taddr] : 0xFFu;
            const bool tok = t < (uint32_t)p.T;
            double val = (inb && tok && RULE == SGW_AGENT_RULE_MOVE) ? tab->value[t & 31u] : 0.0;   // reward read BEFORE the move
            const bool pass = inb && tok && ((p.pass_mask >> (t & 31u)) & 1u);
            const int cy = pass ? ty : y, cx = pass ? tx : x;   // where the agent stands after the move
            gsync<WPE>();   // every thread has read s_type / the target before thread 0 rewrites them
            if (gtid == 0) {
                s_pov[a] = (uint8_t)my_type;
                if (pass) {
                    lg[taddr] = (uint8_t)my_type;
                    lg[oaddr] = (uint8_t)p.default_type;
                    s_pos[2 * a] = (uint8_t)ty;
                    s_pos[2 * a + 1] = (uint8_t)tx;
                }
                st_bits |= (!act_ok ? SGW_STATUS_BAD_ACTION : 0) | ((act_ok && !inb) ? SGW_STATUS_OOB_MOVE : 0) |
                           ((inb && !tok) ? SGW_STATUS_BAD_TYPE : 0);
            }
            if constexpr (RULE == SGW_AGENT_RULE_TAG) {
                // TagAgent.act (sorrel/examples/tag/agents.py:84-106): look at the four neighbours in
                // Location.adjacent order (up, right, down, left; off-map skipped); an agent that is
                // "it" hands the flag to the FIRST neighbour that is a NotIt agent.  Every thread
                // evaluates the same LDS bytes, so `mine_now` stays uniform.
                gsync<WPE>();
                uint32_t mine_now = my_type;
                const int own = zoff + cy * p.W + cx;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int ay = cy + (d == 0 ? -1 : d == 2 ? 1 : 0);
                    const int ax = cx + (d == 1 ? 1 : d == 3 ? -1 : 0);
                    const bool ain = (unsigned)ay < (unsigned)p.H && (unsigned)ax < (unsigned)p.W;
                    const uint32_t nt = ain ? lg[zoff + ay * p.W + ax] : 0xFFu;
                    if (mine_now == p.tag_it && nt == p.tag_notit) {
                        mine_now = p.tag_notit;
                        if (gtid == 0) {
                            lg[own] = (uint8_t)p.tag_notit;
                            lg[zoff + ay * p.W + ax] = (uint8_t)p.tag_it;
                            s_type[a] = (uint8_t)p.tag_notit;
                        }
                        // the neighbour's slot: the agent standing on (ay, ax)
                        if (gtid < p.A && gtid != a && s_pos[2 * gtid] == ay && s_pos[2 * gtid + 1] == ax)
                            s_type[gtid] = (uint8_t)p.tag_it;
                    }
                }
                val = mine_now != p.tag_it ? p.tag_reward : 0.0;
            }
            if (gtid == 0) {
                s_rew[a] = (float)val;
                tot += val;   // world.total_reward += reward, float64, agent order (agent.py:172)
            }
            gsync<WPE>();
        }
        if (p.do_move && gtid >= p.a0 && gtid < p.a1) {      // this turn's rewards (and what TagAgent.pov appends)
            p.rewards[tix * p.ts_rew + env * p.A + gtid] = s_rew[gtid];
            if (p.state_at_pov) p.state_at_pov[env * p.A + gtid] = s_pov[gtid];
        }
        }   // turns

        if (dirty) {
            if (RULE == SGW_AGENT_RULE_MOVE && !(p.flags & SGW_STEP_SWEEP)) {
                // a policy-driven phase (no sweep, plain moves): only the movers' two cells changed -- write those bytes,
                // not the whole grid (with agents i < j both touching a cell, both write its FINAL content: no race)
                if (gtid >= p.a0 && gtid < p.a1) {
                    const uint32_t now = reinterpret_cast<const uint16_t*>(s_pos)[gtid];
                    if (now != yx0) {
                        const int o0 = zoff + (int)(yx0 & 0xFFu) * p.W + (int)(yx0 >> 8), o1 = zoff + (int)(now & 0xFFu) * p.W + (int)(now >> 8);
                        ggrid[o0] = lg[o0];
                        ggrid[o1] = lg[o1];
                    }
                }
            } else {
                store_grid<G>(p, ggrid, lg, gtid);
            }
        }
        if (p.do_move) {
            if (gtid >= p.a0 && gtid < p.a1)
                reinterpret_cast<uint16_t*>(p.pos)[env * p.A + gtid] = reinterpret_cast<const uint16_t*>(s_pos)[gtid];
            if (gtid < p.A && p.agent_state) p.agent_state[env * p.A + gtid] = s_type[gtid];   // a tag can flip any agent
            if (gtid < p.A && p.agent_dir) p.agent_dir[env * p.A + gtid] = s_dir[gtid];
            if (gtid == 0) {
                p.total[env] = tot;
                if (st_bits) atomicOr(p.status, st_bits);
            }
        }
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
constexpr int kMaxUnits = 4;   // 16-byte units per lane (cells <= 4096)
constexpr size_t kLdsPerCu = 160 * 1024;
constexpr size_t kCacheResidentGrid = (size_t)384 << 20;   // grids of a batch up to about this size stay in the 256 MB Infinity Cache + L2 from turn to turn

// (non-temporal observation stores were measured: slower)
#define OBS_STORE(ptr, val) (*(ptr) = (val))


// Bernoulli draws of one 16-byte unit: returns a 16-bit mask of the cells that spawn.
__device__ __forceinline__ uint32_t sweep_hits(const uint4& u, const uint32_t unit, const Params& p, const uint32_t env_id, const uint32_t turn) {
    uint32_t hits = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t dv = k == 0 ? u.x : k == 1 ? u.y : k == 2 ? u.z : u.w;
        const uint32_t m = match_bytes(dv, p.spawn_pat);
        if (m) {
            const U4 w = philox4x32_10(opaque(unit * 4 + k), turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN, p.seed_lo, p.seed_hi);
            const bool f = p.spawn_full != 0;
            uint32_t hb = 0;
            hb |= ((m & 0x80u) && (f || w.x < p.spawn_thr)) ? 1u : 0u;
            hb |= ((m & 0x8000u) && (f || w.y < p.spawn_thr)) ? 2u : 0u;
            hb |= ((m & 0x800000u) && (f || w.z < p.spawn_thr)) ? 4u : 0u;
            hb |= ((m & 0x80000000u) && (f || w.w < p.spawn_thr)) ? 8u : 0u;
            hits |= hb << (4 * k);
        }
    }
    return hits;
}

// Rare second draw: what spawns in each hit cell; written straight into the LDS grid.
__device__ __forceinline__ void sweep_apply(uint32_t hits, const uint32_t unit, uint8_t* lg, const Params& p,
                                            const uint32_t env_id, const uint32_t turn) {
    while (hits) {
        const uint32_t cell = (uint32_t)__ffs(hits) - 1u;
        hits &= hits - 1u;
        const uint32_t off = unit * 16u + cell;   // byte offset == RNG index
        const U4 kw = philox4x32_10(opaque(off >> 2), turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN_KIND, p.seed_lo, p.seed_hi);
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
template <bool ONEHOT, int TL, int TC, int TR, int TH, int TW, bool TAG = false, bool RULES = false, bool STAGE = false, bool MULTI = false>
__global__ __launch_bounds__(kBlock, 8) void step_fast(const Params p) {
    // One wave = one env, one pass: no persistent loop (letting the dispatcher hand out
    // workgroups measured 17 % faster than a persistent grid with software prefetch),
    // wave-private LDS (grid slice + the table words this wave reads), no s_barrier.
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps all env-indexed address math scalar
    const int64_t env = (int64_t)blockIdx.x * 4 + sub;
    if (env >= p.E) return;   // whole wave exits together
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
    const int cells = kStatic ? TL * TH * TW : p.cells;
    const int nunits = (cells + 15) >> 4;   // the last unit may be partly padding (env stride is a multiple of 16)
    constexpr int NU = kStatic ? (TL * TH * TW / 16 + 63) / 64 : kMaxUnits;   // units per lane
    const int zoff = p.zA * HW;
    constexpr int NW = TC ? (TC + 3) / 4 : 4;   // counter words

    // wave-private LDS: [table words][grid]
    uint8_t* wl = smem + sub * p.env_lds;
    const DevTables* gtab = p.tab;
    const uint32_t env_id = p.first_env + (uint32_t)env;

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
#pragma unroll
        for (int q = 0; q < (NW + 1) / 2; ++q) wd[lane + 64 * q] = reinterpret_cast<const uint32_t*>(gtab->delta)[lane + 64 * q];
    } else {
        double* wa = reinterpret_cast<double*>(wl);
        for (int i = lane; i < SGW_MAX_TYPES * SGW_MAX_CHANNELS; i += 64) wa[i] = reinterpret_cast<const double*>(gtab->appearance)[i];
    }
    const uint32_t* wdelta = reinterpret_cast<const uint32_t*>(wl);                 // [NW][32]
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
    constexpr bool kStageAlways = ONEHOT && STAGE && !(TL && TH && TW);
    const bool stage = kStageAlways || (ONEHOT && (TL && TH && TW) && p.obs_stage > 0 && p.a0 == 0 && p.a1 == p.A);
    [[maybe_unused]] int ch_a0 = 0;            // first agent of the chunk being staged (STAGE)
    [[maybe_unused]] uint32_t ch_shift = 0;    // misalignment (in elements) of the chunk's first element in global memory
    if constexpr (kStageAlways) ch_shift = (uint32_t)(env * (int64_t)(p.A * C * VV)) & 3u;

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

    {
        double tot = p.do_move ? p.total[env] : 0.0;

#ifdef SGW_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(1);   // global loads have arrived
        // ---- grid -> LDS; the Bernoulli half of the sweep runs on the registers
        [[maybe_unused]] uint32_t hits[NU];
        if constexpr (!kStatic) {
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
        int st_lane = 0;
        uint32_t taddr_v = 0xFFFFFFFFu, oaddr_v = 0, npos = 0, rew_bits = 0, moved = 0;   // per turn; the write-back reads the last turn's
        const uint32_t nturns = MULTI ? p.nturns : 1u;
        for (uint32_t tix = 0; tix < nturns; ++tix) {
        const uint32_t turn = p.turn + tix;
        if constexpr (RULES) {
            gsync<1>();
            if (do_sweep) {
                // Ordered sweep in LDS.  The reference visits cells in (y, x, z) order and a rule may read another
                // layer of its own column (environment.py:88-91): going layer by layer, lower layers first, gives every
                // cell the same view (lower layers already swept, higher ones not yet); rules write their own cell only.
                const uint32_t* lg32 = reinterpret_cast<const uint32_t*>(lg);
                for (int z = 0; z < L; ++z) {
                    const int lo = z * HW, hi = lo + HW;
                    for (int d = (lo >> 2) + lane; d < ((hi + 3) >> 2); d += 64) {   // one dword = four cells = one Philox block
                        const uint32_t word = lg32[d];
                        uint32_t tj[4];
                        bool spj[4], bcj[4];
                        bool any_sp = false;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int off = 4 * d + j;
                            tj[j] = (word >> (8 * j)) & 0xFFu;
                            const bool in = off >= lo && off < hi && tj[j] < (uint32_t)SGW_MAX_TYPES;
                            spj[j] = in && ((p.spawn_mask >> (tj[j] & 31u)) & 1u);
                            bcj[j] = in && ((p.become_mask >> (tj[j] & 31u)) & 1u);
                            any_sp = any_sp || spj[j];
                        }
                        if (any_sp) {
                            const U4 w = philox4x32_10(opaque((uint32_t)d), turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN, p.seed_lo, p.seed_hi);
                            uint32_t hit = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (spj[j] && (((p.thr_full_mask >> tj[j]) & 1u) || word_of(w, j) < rt->thr_lo[tj[j]])) hit |= 1u << j;
                            if (hit) {   // rare: what spawns
                                const U4 kw = philox4x32_10(opaque((uint32_t)d), turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN_KIND, p.seed_lo, p.seed_hi);
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
                    gsync<1>();
                }
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
                if (lane + 64 * k < nunits && do_sweep) hits[k] = sweep_hits(u[k], (uint32_t)(lane + 64 * k), p, env_id, turn);
            }
            gsync<1>();
            if (do_sweep) {
#pragma unroll
                for (int k = 0; k < NU; ++k)
                    if (lane + 64 * k < nunits) sweep_apply(hits[k], (uint32_t)(lane + 64 * k), lg, p, env_id, turn);
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
                const U4 w = philox4x32_10(opaque((uint32_t)lane >> 2), turn, env_id, (p.epoch << 4) | SGW_STREAM_ACTION,
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
            ch_shift = (uint32_t)(turn_obs + env * (int64_t)(p.A * C * VV)) & 3u;
        }

        STAMP(3);   // move inputs (action draw) done
        // STAGE: the staged chunk [a_lo, a_hi) leaves for HBM.  Dword i of the (shifted) staging area is the 16-byte
        // aligned float4 number i of the chunk's span in global memory; the span's first and last float4 may also hold
        // elements of a neighbouring chunk / env, so those two leave element by element.
        [[maybe_unused]] auto emit_chunk = [&](const int a_lo, const int a_hi) {
            gsync<1>();
            typedef float vfloat4 __attribute__((ext_vector_type(4)));
            const int N = (a_hi - a_lo) * C * VV;
            const int64_t e0 = turn_obs + (env * p.A + a_lo) * (int64_t)(C * VV);
            const int sh = (int)ch_shift;
            const int nd = (sh + N + 3) >> 2;
            const uint32_t* ob4 = reinterpret_cast<const uint32_t*>(ob);
            if (!p.obs_u8) {
                float* gb = p.obs + (e0 - sh);
                for (int i = lane; i < nd; i += 64) {
                    const uint32_t b = ob4[i];
                    vfloat4 v;
                    v.x = (float)(b & 0xFFu);
                    v.y = (float)((b >> 8) & 0xFFu);
                    v.z = (float)((b >> 16) & 0xFFu);
                    v.w = (float)(b >> 24);
                    const int lo = 4 * i - sh;       // chunk element held by byte 0 of this dword
                    if (lo >= 0 && lo + 4 <= N) {
                        __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(gb + 4 * i));
                    } else {
                        if (lo >= 0 && lo < N) gb[4 * i] = v.x;
                        if (lo + 1 >= 0 && lo + 1 < N) gb[4 * i + 1] = v.y;
                        if (lo + 2 >= 0 && lo + 2 < N) gb[4 * i + 2] = v.z;
                        if (lo + 3 >= 0 && lo + 3 < N) gb[4 * i + 3] = v.w;
                    }
                }
            } else {
                uint8_t* gb = reinterpret_cast<uint8_t*>(p.obs) + (e0 - sh);
                for (int i = lane; i < nd; i += 64) {
                    const uint32_t b = ob4[i];
                    const int lo = 4 * i - sh;
                    if (lo >= 0 && lo + 4 <= N) {
                        __builtin_nontemporal_store(b, reinterpret_cast<uint32_t*>(gb + 4 * i));
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (lo + j >= 0 && lo + j < N) gb[4 * i + j] = (uint8_t)(b >> (8 * j));
                    }
                }
            }
            gsync<1>();
        };
        // ---- agents, strictly in list order (SGW_STEP_OBS_NEXT: one extra, observe-only iteration for agent a1)
        const int a_end = (p.obs_next && p.a1 < p.A) ? p.a1 + 1 : p.a1;
        for (int a = p.a0; a < a_end; ++a) {
            if constexpr (kStageAlways) {
                if (a - ch_a0 == p.stage_agents) {   // the staging area is full: out with it, start the next chunk
                    if (write_obs) emit_chunk(ch_a0, a);
                    ch_a0 = a;
                    ch_shift = (uint32_t)(turn_obs + (env * p.A + a) * (int64_t)(C * VV)) & 3u;
                }
            }
            const int s_o = __builtin_amdgcn_readlane((int)oaddr_v, a);
            if (p.obs_next ? a == p.a1 : write_obs) {
                const int y = __builtin_amdgcn_readlane((int)py, a);
                const int x = __builtin_amdgcn_readlane((int)px, a);
                const int cbase = s_o - zoff;
                float* obase = p.obs + turn_obs + ((env * p.A + a) * (int64_t)C) * VV;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (64 * k >= VV) break;
                    const int w = lane + 64 * k;
                    if (w < VV) {
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
                            for (int q = 0; q < NW; ++q) cnt[q] = inb ? cnt[q] : p.fill_delta[q];
                            if (stage) {
                                uint8_t* os = ob + (kStageAlways ? (int)ch_shift + ((a - ch_a0) * C) * VV : (a * C) * VV) + w;
#pragma unroll
                                for (int q = 0; q < NW; ++q) {
#pragma unroll
                                    for (int b = 0; b < 4; ++b) {
                                        const int c = 4 * q + b;
                                        if (c < C) os[c * VV] = (uint8_t)(cnt[q] >> (8 * b));
                                    }
                                }
                            } else if constexpr (kStageAlways) {
                                // unreachable: a STAGE kernel always stages
                            } else if (!p.obs_u8) {
#pragma unroll
                                for (int q = 0; q < NW; ++q) {
#pragma unroll
                                    for (int b = 0; b < 4; ++b) {
                                        const int c = 4 * q + b;
                                        if (c < C) OBS_STORE(o + c * VV, (float)((cnt[q] >> (8 * b)) & 0xFFu));
                                    }
                                }
                            } else {   // compact format: the same counts as bytes
                                uint8_t* o8 = reinterpret_cast<uint8_t*>(p.obs) + (o - p.obs);
#pragma unroll
                                for (int q = 0; q < NW; ++q) {
#pragma unroll
                                    for (int b = 0; b < 4; ++b) {
                                        const int c = 4 * q + b;
                                        if (c < C) o8[c * VV] = (uint8_t)((cnt[q] >> (8 * b)) & 0xFFu);
                                    }
                                }
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
                uint32_t nt[4];
                bool ain[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int ay = cy + (d == 0 ? -1 : d == 2 ? 1 : 0), ax = cx + (d == 1 ? 1 : d == 3 ? -1 : 0);
                    ain[d] = (unsigned)ay < (unsigned)H && (unsigned)ax < (unsigned)W;
                    nt[d] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lg[ain[d] ? zoff + ay * W + ax : own]);
                }
                int dstar = -1;
#pragma unroll
                for (int d = 3; d >= 0; --d)
                    if (ain[d] && nt[d] == p.tag_notit) dstar = d;
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
            if (write_obs) emit_chunk(ch_a0, p.a1);
        } else if (stage && write_obs) {
            gsync<1>();
            const int nd = (p.A * C * VV) >> 2;   // dwords of staged bytes (the host stages only multiples of 4 elements)
            const uint32_t* ob4 = reinterpret_cast<const uint32_t*>(ob);
            if (!p.obs_u8) {
                // Non-temporal (streaming) stores: every wave instruction here writes eight whole 128-byte lines that
                // nothing reads again in this launch; keeping them out of the caches leaves those to the grids (134 MB,
                // re-read next turn) and takes config 3 from 167 to 125-132 us.  (The same hint on the per-agent dword
                // stores of the unstaged path, which write partial lines, was measured SLOWER.)
                typedef float vfloat4 __attribute__((ext_vector_type(4)));
                vfloat4* o4 = reinterpret_cast<vfloat4*>(p.obs + turn_obs + env * (int64_t)(p.A * C * VV));
                for (int i = lane; i < nd; i += 64) {
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
            if (!TAG && !RULES && !do_sweep) {
                // a policy-driven phase (no sweep, plain moves): only the movers' two cells changed -- write those bytes,
                // not the whole grid (two movers touching one cell both write its FINAL content: no race)
                if (mine && moved) {
                    uint8_t* g = p.grid + env * p.env_stride;
                    g[oaddr_v] = lg[oaddr_v];
                    g[taddr_v] = lg[taddr_v];
                }
            } else {
                uint4* dst = reinterpret_cast<uint4*>(p.grid + env * p.env_stride);
#pragma unroll
                for (int k = 0; k < NU; ++k)
                    if (lane + 64 * k < nunits) dst[lane + 64 * k] = lg16[lane + 64 * k];
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

// ---------------------------------------------------------------- big step kernel
// Workgroup-per-env kernel for worlds above 4 KiB (BASELINE config 5: 128x128x2 = 32 KiB
// of LDS per env, 64 agents, 11x11 windows).  Same ingredients as step_fast (register
// sweep, per-agent move inputs computed in parallel, scalar sequential part), plus a
// JOURNAL so that the A sequential agent phases do not serialise the observation work:
//   phase M  wave 0 resolves all moves in registers: the targets of all agents are read from LDS at
//            once, and a short scalar loop corrects each for earlier movers with two ballots (no LDS
//            access, no cross-wave hand-off); it records what each agent found and whether it moved;
//   phase R  all waves render the observations in parallel from the FINAL grid; agent a must see the
//            grid after the moves of agents < a only, so the moves of agents >= a that touch its
//            window (found with one ballot) are undone in registers, latest first.
// History (config 5, 2048 envs, us per launch): generic kernel 274; turn word passed from wave to
// wave 131 -> 113 (three dependent LDS round trips per agent); LDS move chain + journal 124;
// the same with renderers racing the mover (progress words, dynamic queue) 108; moves resolved in
// registers + barrier 118-122.  The last is kept: it has no cross-wave race to reason about.
// Requires impassable agent types (a passable agent could be "entered" twice in one turn, which the
// two-batch patch cannot order); the host dispatch checks it.
// Eight waves per workgroup, four workgroups per CU = the CU's 32 wave slots: measured 115 us per config-5 launch
// against 123 us with four waves per workgroup and 143 us with two (round 2, same box, interleaved A/B).
#ifndef SGW_BIG_THREADS
#define SGW_BIG_THREADS 512
#endif
constexpr int kBigThreads = SGW_BIG_THREADS;
constexpr int kBigWaves = kBigThreads / 64;
constexpr int kBigAgentLds = 64 * 4 * 4 + 64 * 8 + 16 + 32 * 8 + 64;   // ta, oa, npos, rew | val f64 | turn+moved | value table | agent types

// MULTI: sgw_rollout's variant -- a turn loop around sweep / moves / observations with the env's 32 KiB resident in LDS
// (later turns sweep the units read back from LDS; only the last turn is followed by the write-back).
template <bool ONEHOT, int TL, int TC, int TR, bool MULTI = false>
__global__ __launch_bounds__(kBigThreads, kBigThreads == 512 ? 6 : (kBigThreads == 256 ? 3 : 1)) void step_big(const Params p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t env = blockIdx.x;
    const uint32_t env_id = p.first_env + (uint32_t)env;
#ifdef SGW_STAMPS
    unsigned long long tprev_ = 0;
#define STAMPB(i)                                                                                            \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long t_;                                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (tid == 0 && (i) > 0 && env < kStampEnvs) g_stamps[env * 8 + (i)-1] = t_ - tprev_;                 \
        tprev_ = t_;                                                                                         \
    } while (0)
    STAMPB(0);
    if (tid == 0 && env < kStampEnvs) {
        g_stamps[env * 8 + 6] = tprev_;
        g_stamps[env * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
#else
#define STAMPB(i)
#endif

    const int L = TL ? TL : p.L;
    const int C = TC ? TC : p.C;
    const int r = TR ? TR : p.r;
    const int V = 2 * r + 1, VV = V * V;
    const int H = p.H, W = p.W;
    // LDS image of the grid: rows of P >= W bytes.  With P == W + 16 (worlds whose width is a multiple of 16) window row
    // i of an observation starts (W + 16) / 4 = 4 (mod 32) banks after row i - 1, so the ~3 rows a 32-lane group of the
    // 11x11 gather touches fall on disjoint banks; with P == W (a 128-byte pitch) they all fell on the same ones
    // (34 % of the LDS cycles of config 5 were bank conflicts).
    const int P = p.big_pitch, HW = H * P;           // HW: LDS bytes of one layer
    const int upr = W >> 4;                           // 16-byte units per row (used only when P != W)
    const bool padded = P != W;
    const int cells = p.cells;
    const int nunits = (cells + 15) >> 4;   // the last unit may be partly padding (env stride is a multiple of 16)
    const int zoff = p.zA * HW;
    constexpr int NW = TC ? (TC + 3) / 4 : 4;
    constexpr int NP = TR ? ((2 * TR + 1) * (2 * TR + 1) + 63) / 64 : 2;   // window passes per wave held in registers
    // HBM unit index / byte offset -> LDS unit index / byte offset (one pad unit per row)
    auto lunit = [&](int idx) { return padded ? idx + idx / upr : idx; };
    auto lbyte = [&](uint32_t off) { return padded ? off + (off / (uint32_t)W) * 16u : off; };

    // LDS: [tables][agent arrays][grid]
    uint32_t* s_ta = reinterpret_cast<uint32_t*>(smem + p.tab_bytes);      // journal entry of each agent (phase M -> R)
    uint32_t* s_oa = s_ta + 64;                                             // packed (y, x) at the start of the turn
    uint32_t* s_np = s_oa + 64;                                             // packed (y, x) if the move succeeds
    uint32_t* s_rm = s_np + 64;                                             // reward f32 bits
    double* s_val = reinterpret_cast<double*>(s_rm + 64);                   // reward f64 (for total, in agent order)
    double* s_vtab = s_val + 64 + 2;                                         // value[32] (keeps global loads out of the chain)
    uint8_t* s_atype = reinterpret_cast<uint8_t*>(s_vtab + 32);                       // agent_type[64]
    uint8_t* lg = smem + p.tab_bytes + kBigAgentLds;
    uint4* lg16 = reinterpret_cast<uint4*>(lg);
    const DevTables* gtab = p.tab;

    const bool write_obs = !(p.flags & SGW_STEP_NO_OBS);
    const bool do_sweep = (p.flags & SGW_STEP_SWEEP) != 0;
    const bool dirty = do_sweep || (p.do_move && p.a1 > p.a0);
    const bool rnd = (p.flags & SGW_STEP_RANDOM_ACTIONS) != 0;

    // ---- tables -> LDS
    if constexpr (ONEHOT) {
        uint32_t* wd = reinterpret_cast<uint32_t*>(smem);
        if (tid < 4 * SGW_MAX_TYPES) wd[tid] = reinterpret_cast<const uint32_t*>(gtab->delta)[tid];
    } else {
        double* wa = reinterpret_cast<double*>(smem);
        for (int i = tid; i < SGW_MAX_TYPES * SGW_MAX_CHANNELS; i += kBigThreads) wa[i] = reinterpret_cast<const double*>(gtab->appearance)[i];
    }
    if (tid < 32) s_vtab[tid] = gtab->value[tid];
    if (tid >= 64 && tid < 128) s_atype[tid - 64] = gtab->agent_type[tid - 64];
    const uint32_t* wdelta = reinterpret_cast<const uint32_t*>(smem);
    const double(*wapp)[SGW_MAX_CHANNELS] = reinterpret_cast<const double(*)[SGW_MAX_CHANNELS]>(smem);

    // per-agent state of wave 0 (lane a = agent a), carried from turn to turn of a rollout
    uint32_t yx = 0;
    int st_lane = 0;
    const bool mine = tid >= p.a0 && tid < p.a1 && tid < p.A;
    uint32_t ta_v = 0xFFFFFFFFu, npos_v = 0, oaddr_v = 0, jr = 0;
    if (wv == 0 && tid < p.A) {
        yx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + tid];
        if ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W) {   // garbage in: stay inside the LDS grid, and say so
            yx = 0;
            atomicOr(p.status, SGW_STATUS_BAD_POS);
        }
    }
    double tot = (tid == 0 && p.do_move) ? p.total[env] : 0.0;
    const uint32_t nturns = MULTI ? p.nturns : 1u;
    for (uint32_t tix = 0; tix < nturns; ++tix) {
    const uint32_t turn = p.turn + tix;
    // ---- grid -> LDS (first turn), sweep on the registers, 4 units per thread per round
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.grid + env * p.env_stride);
        for (int base = 0; base < nunits; base += 4 * kBigThreads) {
            uint4 u[4];
            uint32_t hits[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + k * kBigThreads + tid;
                if (idx < nunits) u[k] = (MULTI && tix > 0) ? lg16[lunit(idx)] : src[idx];
                if ((cells & 15) && idx == nunits - 1) {   // ragged world: mask the bytes past the last cell
                    const int tail = cells & 15;
                    uint32_t d[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int keep = tail - 4 * q;
                        if (keep <= 0) d[q] = 0xFFFFFFFFu;
                        else if (keep < 4) d[q] |= 0xFFFFFFFFu << (8 * keep);
                    }
                    u[k] = make_uint4(d[0], d[1], d[2], d[3]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + k * kBigThreads + tid;
                hits[k] = 0;
                if (idx < nunits) {
                    if (!(MULTI && tix > 0)) lg16[lunit(idx)] = u[k];
                    if (do_sweep) hits[k] = sweep_hits(u[k], (uint32_t)idx, p, env_id, turn);
                }
            }
            if (do_sweep) {
                // this thread wrote these units itself (DS ops of a wave are ordered), so the rare
                // kind draws can patch LDS right away; one combined loop keeps the trip count low
                uint32_t h0 = hits[0], h1 = hits[1], h2 = hits[2], h3 = hits[3];   // named: keeps them in registers
                while (__builtin_amdgcn_readfirstlane(__any((h0 | h1 | h2 | h3) != 0))) {
                    // this lane's next hit cell: lowest set bit of the first non-empty unit
                    const int k = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
                    const uint32_t hk = h0 ? h0 : h1 ? h1 : h2 ? h2 : h3;
                    if (hk) {
                        const uint32_t cell = (uint32_t)__ffs(hk) - 1u;
                        const uint32_t cleared = hk & (hk - 1u);
                        h0 = k == 0 ? cleared : h0;
                        h1 = k == 1 ? cleared : h1;
                        h2 = k == 2 ? cleared : h2;
                        h3 = k == 3 ? cleared : h3;
                        const uint32_t off = (uint32_t)(base + k * kBigThreads + tid) * 16u + cell;
                        const U4 kw = philox4x32_10(opaque(off >> 2), turn, env_id, (p.epoch << 4) | SGW_STREAM_SPAWN_KIND,
                                                   p.seed_lo, p.seed_hi);
                        const uint32_t pick = __umulhi(word_of(kw, off & 3u), p.spawn_n);
                        lg[lbyte(off)] = (uint8_t)(((pick < 4 ? p.choice_lo : p.choice_hi) >> (8 * (pick & 3u))) & 0xFFu);
                    }
                }
            }
        }
    }

    // ---- per-agent move inputs, all agents at once (wave 0: lane a = agent a)
    ta_v = 0xFFFFFFFFu;
    jr = 0;
    if (wv == 0) {
        if (tid < p.A) {
            const uint32_t py = yx & 0xFFu, px = yx >> 8;
            oaddr_v = (uint32_t)zoff + py * (uint32_t)P + px;
            npos_v = yx;
            if (p.do_move && mine) {
                uint32_t act;
                if (rnd) {
                    const U4 w = philox4x32_10(opaque((uint32_t)tid >> 2), turn, env_id, (p.epoch << 4) | SGW_STREAM_ACTION,
                                               p.seed_lo, p.seed_hi);
                    act = __umulhi(word_of(w, tid & 3), (uint32_t)p.nact);
                    p.actions[tix * p.ts_act + env * p.A + tid] = (uint8_t)act;
                } else {
                    act = p.actions[tix * p.ts_act + env * p.A + tid];
                }
                const bool act_ok = act < (uint32_t)p.nact;
                const int dy = (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1;
                const int dx = (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1;
                const int ty = (int)py + dy, tx = (int)px + dx;
                const bool inb = (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
                if (act_ok && inb) {
                    ta_v = (uint32_t)(zoff + ty * P + tx);
                    npos_v = (uint32_t)ty | ((uint32_t)tx << 8);
                }
                st_lane |= !act_ok ? SGW_STATUS_BAD_ACTION : (!inb ? SGW_STATUS_OOB_MOVE : 0);
            }
        }
        s_oa[tid] = yx;          // packed (y, x) at the start of the turn
        s_np[tid] = npos_v;      // packed (y, x) if the move succeeds
        s_ta[tid] = 0;           // journal: empty
    }
    __syncthreads();             // grid (+ sweep patches) and tables visible to every wave
    STAMPB(1);                   // load + sweep done

    // ---- phase M: the strictly sequential part, by wave 0 alone, entirely in registers.
    // All targets are read from the pre-move grid in ONE LDS round trip (lane a = agent a).  What
    // agent a finds on its target when its turn comes differs from that only if an earlier mover
    // left from or entered that very cell; the scalar loop below finds the latest such mover with
    // two ballots (no LDS access inside the loop).  The grid is patched afterwards in two ordered
    // batches: every mover's old cell <- default, then every mover's new cell <- its type (a cell
    // can be left and then entered in one turn, never the other way round: an agent moves once).
    if (wv == 0 && p.do_move) {
        const uint32_t atype_v = s_atype[lane];
        const bool validv = ta_v != 0xFFFFFFFFu;
        const uint32_t t0_v = lg[validv ? ta_v : oaddr_v];
        uint32_t passed_v = 0;
        // What an agent finds on its target can differ from the pre-move grid only if an earlier mover entered that cell
        // (two agents share a target) or left it (the target is another agent's cell).  An agent INTERFERES if it shares
        // its target with another agent or targets another agent's cell; everyone else resolves at once from the
        // pre-move grid, and only the interfering agents (typically none, or a pair) are walked, in agent order.
        bool cf = false;
        const bool self = validv && ta_v == oaddr_v;     // targets its own cell (a non-move action): finds itself, whoever moves
        const bool markable = mine && validv && t0_v < 32u && !((p.agent_mask >> t0_v) & 1u);
        uint32_t* gw = reinterpret_cast<uint32_t*>(lg);
        const uint32_t msh = 8u * (ta_v & 3u);
        if (mine && validv && !markable && !self) cf = true;
        // two spare bits of the target's LDS byte (type ids are < 32): 0x40 = claimed, 0x80 = claimed more than once
        if (markable) cf = ((atomicOr(&gw[ta_v >> 2], 0x40u << msh) >> msh) & 0x40u) != 0;
        if (markable && cf) atomicOr(&gw[ta_v >> 2], 0x80u << msh);
        if (markable) cf = ((atomicOr(&gw[ta_v >> 2], 0u) >> msh) & 0x80u) != 0;   // every claimant of a contested cell, the first one too (an RMW: ordered behind the marks)
        unsigned long long cmask = __ballot(cf);
        if (markable) atomicAnd(&gw[ta_v >> 2], ~(0xC0u << msh));   // marks off again before anyone else reads the grid
        {
            const bool tok = validv && t0_v < (uint32_t)p.T;
            const bool pass = tok && ((p.pass_mask >> (t0_v & 31u)) & 1u);
            if (mine) {   // final for the agents that do not interfere, provisional (and not yet visible, see `lane < a`) for the others
                jr = (t0_v & 0xFFu) | (tok ? 0x100u : 0u) | (pass ? 0x200u : 0u) | ((validv && !tok) ? 0x400u : 0u);
                passed_v = pass ? 1u : 0u;
            }
        }
        while (cmask) {
            const int a = __builtin_ctzll(cmask);
            cmask &= cmask - 1ull;
            const uint32_t X = (uint32_t)__builtin_amdgcn_readlane((int)ta_v, a);
            const bool valid = X != 0xFFFFFFFFu;
            uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)t0_v, a);
            // what earlier movers did to that cell: the latest one that entered or left it decides
            const unsigned long long m_dst = __ballot(passed_v && lane < a && ta_v == X);
            const unsigned long long m_src = __ballot(passed_v && lane < a && oaddr_v == X);
            const unsigned long long m_any = m_dst | m_src;
            if (m_any) {
                const int last = 63 - __builtin_clzll(m_any);
                const uint32_t at_last = (uint32_t)__builtin_amdgcn_readlane((int)atype_v, last);
                t = ((m_dst >> last) & 1ull) ? at_last : p.default_type;
            }
            const bool tok = valid && t < (uint32_t)p.T;
            const bool pass = tok && ((p.pass_mask >> (t & 31u)) & 1u);
            const uint32_t entry = (t & 0xFFu) | (tok ? 0x100u : 0u) | (pass ? 0x200u : 0u) | ((valid && !tok) ? 0x400u : 0u);
            jr = lane == a ? entry : jr;
            passed_v = lane == a ? (pass ? 1u : 0u) : passed_v;
        }
        if (passed_v) lg[oaddr_v] = (uint8_t)p.default_type;
        gsync<1>();
        if (passed_v) lg[ta_v] = (uint8_t)atype_v;
        const double val = (jr & 0x100u) ? s_vtab[jr & 31u] : 0.0;     // reward = value of the target BEFORE the move
        s_val[lane] = val;
        s_rm[lane] = __float_as_uint((float)val);
        s_ta[lane] = jr;                                                // journal for the render phase
        if (jr & 0x400u) st_lane |= SGW_STATUS_BAD_TYPE;
        if (mine) p.rewards[tix * p.ts_rew + env * p.A + tid] = (float)val;   // this turn's rewards
        gsync<1>();
        if (tid == 0)
            for (int a = p.a0; a < p.a1; ++a) tot += s_val[a];           // float64, agent order (agent.py:172)
    }
    __syncthreads();
    STAMPB(2);                   // phase M done

    // ---- phase R: observations, all waves in parallel (agent a -> wave (a - a0) mod waves).
    // LDS now holds the grid AFTER all moves of this call; agent a must see it after the moves of
    // agents < a only, so the moves of agents b >= a that touch a's window (one ballot) are undone
    // in registers, latest first; an undo restores the two cells the move changed.
    if (write_obs || p.obs_next) {
        // per-lane window geometry: NP cells per lane
        int wdi[NP], wdj[NP], woff[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int w = lane + 64 * k;
            const int i = w / V, j = w - i * V;
            wdi[k] = i - r;
            wdj[k] = j - r;
            woff[k] = wdi[k] * P + wdj[k];
        }
        // lane b: journal of agent b (where it was, where it went, what it found there)
        const uint32_t jb = s_ta[lane], srcb = s_oa[lane], dstb = s_np[lane], atb = s_atype[lane];
        const bool movedb = p.do_move && (jb & 0x200u) && lane >= p.a0 && lane < p.a1;
        const int zsh = 8 * (p.zA & 3), zw = p.zA >> 2;
        // SGW_STEP_OBS_NEXT: only agent a1, which sees the grid after ALL moves of this call (nothing to undo)
        const int r_lo = p.obs_next ? p.a1 : p.a0, r_hi = p.obs_next ? (p.a1 < p.A ? p.a1 + 1 : p.a1) : p.a1;
        for (int a = r_lo + wv; a < r_hi; a += kBigWaves) {
            const uint32_t pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_oa[a]);
            const int y = (int)(pk & 0xFFu), x = (int)((pk >> 8) & 0xFFu);
            const int cbase = y * P + x;
            uint32_t tb[NP][2];
            bool inbk[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int w = lane + 64 * k;
                inbk[k] = (unsigned)(y + wdi[k]) < (unsigned)H && (unsigned)(x + wdj[k]) < (unsigned)W;
                const int off = (inbk[k] && w < VV) ? cbase + woff[k] : 0;
                uint32_t lo = 0, hi = 0;
                if constexpr (TL != 0) {
#pragma unroll
                    for (int z = 0; z < TL; ++z) {
                        const uint32_t t = lg[z * HW + off];
                        if (z < 4) lo |= (t & 31u) << (8 * z);
                        else hi |= (t & 31u) << (8 * (z - 4));
                    }
                } else {
                    for (int z = 0; z < L; ++z) {
                        const uint32_t t = lg[z * HW + off];
                        if (z < 4) lo |= (t & 31u) << (8 * z);
                        else hi |= (t & 31u) << (8 * (z - 4));
                    }
                }
                tb[k][0] = lo;
                tb[k][1] = hi;
            }
            // which later moves touch this window?  (lane b tests move b)
            const int sy = (int)(srcb & 0xFFu), sx = (int)((srcb >> 8) & 0xFFu);
            const int ey = (int)(dstb & 0xFFu), ex = (int)((dstb >> 8) & 0xFFu);
            const bool near_src = (unsigned)(sy - y + r) <= (unsigned)(2 * r) && (unsigned)(sx - x + r) <= (unsigned)(2 * r);
            const bool near_dst = (unsigned)(ey - y + r) <= (unsigned)(2 * r) && (unsigned)(ex - x + r) <= (unsigned)(2 * r);
            unsigned long long undo = __ballot(movedb && lane >= a && (near_src || near_dst));
            while (undo) {
                const int b = 63 - __builtin_clzll(undo);            // latest move first
                undo &= ~(1ull << b);
                const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)srcb, b);
                const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)dstb, b);
                const uint32_t oldt = (uint32_t)__builtin_amdgcn_readlane((int)jb, b) & 31u;    // what the target held
                const uint32_t agt = (uint32_t)__builtin_amdgcn_readlane((int)atb, b) & 31u;    // the mover itself
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const uint32_t key = (uint32_t)((y + wdi[k]) & 0xFF) | ((uint32_t)((x + wdj[k]) & 0xFF) << 8);
                    const bool at_dst = inbk[k] && key == dst, at_src = inbk[k] && key == src;
                    if (at_dst || at_src) {
                        const uint32_t nv = at_src ? agt : oldt;     // src restored last (matters only if src == dst)
                        if (zw == 0) tb[k][0] = (tb[k][0] & ~(0xFFu << zsh)) | (nv << zsh);
                        else tb[k][1] = (tb[k][1] & ~(0xFFu << zsh)) | (nv << zsh);
                    }
                }
            }
            float* obase = p.obs + tix * p.ts_obs + ((env * p.A + a) * (int64_t)C) * VV;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int w = lane + 64 * k;
                if (w < VV) {
                    float* o = obase + w;
                    if constexpr (ONEHOT) {
                        uint32_t cnt[NW];
#pragma unroll
                        for (int q = 0; q < NW; ++q) cnt[q] = 0;
                        for (int z = 0; z < L; ++z) {
                            const uint32_t t = z < 4 ? (tb[k][0] >> (8 * z)) & 31u : (tb[k][1] >> (8 * (z - 4))) & 31u;
#pragma unroll
                            for (int q = 0; q < NW; ++q) cnt[q] += wdelta[q * 32 + t];
                        }
#pragma unroll
                        for (int q = 0; q < NW; ++q) cnt[q] = inbk[k] ? cnt[q] : p.fill_delta[q];
                        if (!p.obs_u8) {
#pragma unroll
                            for (int q = 0; q < NW; ++q) {
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const int c = 4 * q + b;
                                    if (c < C) OBS_STORE(o + c * VV, (float)((cnt[q] >> (8 * b)) & 0xFFu));
                                }
                            }
                        } else {   // compact format: the same counts as bytes
                            uint8_t* o8 = reinterpret_cast<uint8_t*>(p.obs) + (o - p.obs);
#pragma unroll
                            for (int q = 0; q < NW; ++q) {
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const int c = 4 * q + b;
                                    if (c < C) o8[c * VV] = (uint8_t)((cnt[q] >> (8 * b)) & 0xFFu);
                                }
                            }
                        }
                    } else {
                        for (int c = 0; c < C; ++c) {
                            double acc = wapp[tb[k][0] & 31u][c];   // left-to-right float64 layer sum
                            for (int z = 1; z < L; ++z) {
                                const uint32_t t = z < 4 ? (tb[k][0] >> (8 * z)) & 31u : (tb[k][1] >> (8 * (z - 4))) & 31u;
                                acc += wapp[t][c];
                            }
                            OBS_STORE(o + c * VV, obs_finish(inbk[k] ? acc : wapp[p.fill_type][c], p.obs_post));
                        }
                    }
                }
            }
        }
    }
    // windows wider than NP*64 cells (not a BASELINE shape): handled by the generic kernel (host dispatch)
    __syncthreads();
    STAMPB(3);                   // phase R done (all waves)
    if (MULTI && tix + 1 < nturns && wv == 0 && tid < p.A) yx = (jr & 0x200u) ? npos_v : yx;   // the next turn starts where this one ended
    }   // turns

    // ---- write-back
    if (dirty && !do_sweep) {
        // a policy-driven phase (no sweep): only the movers' two cells changed -- write those bytes, not the whole grid
        if (wv == 0 && mine && (jr & 0x200u)) {
            uint8_t* g = p.grid + env * p.env_stride + p.zA * H * W;
            g[(yx & 0xFFu) * W + (yx >> 8)] = lg[oaddr_v];
            g[(npos_v & 0xFFu) * W + (npos_v >> 8)] = lg[ta_v];
        }
    } else if (dirty) {
        uint4* dst = reinterpret_cast<uint4*>(p.grid + env * p.env_stride);
        for (int idx = tid; idx < nunits; idx += kBigThreads) dst[idx] = lg16[lunit(idx)];
    }
    STAMPB(4);                   // write-back issued
#ifdef SGW_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMPB(5);                   // wave 0's stores acknowledged
    STAMPB(6);
    if (p.do_move) {
        if (wv == 0 && mine) {
            reinterpret_cast<uint16_t*>(p.pos)[env * p.A + tid] = (uint16_t)((jr & 0x200u) ? npos_v : yx);
            if (st_lane) atomicOr(p.status, st_lane);
        }
        if (tid == 0) p.total[env] = tot;
    }
}

// ---------------------------------------------------------------- phase kernel
// One policy-driven phase WITHOUT staging the env: MovingAgent.act of agent a0 (if a0 < a1) and / or the observation of
// ONE agent -- agent a1 after that move (SGW_STEP_OBS_NEXT), or agent a0 before it (the plain per-agent step /
// sgw_observe of one agent).  A phase touches one target cell and one (2r+1)^2 window; the step kernels stage the whole
// env through LDS for that (config 3: 2 KiB in, 2 KiB out per env and phase).  Here a wave per env reads the target
// byte and its window bytes straight from global memory (the grids of a batch sit in L2 / Infinity Cache between the
// phases of a turn), applies the move to the gathered bytes in registers (no reliance on store-to-load ordering across
// lanes) and writes the two changed cells.  Plain moves only (SGW_AGENT_RULE_MOVE), no sweep.
template <bool ONEHOT>
__global__ __launch_bounds__(kBlock, 8) void phase_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t env = (int64_t)blockIdx.x * 4 + sub;
    if (env >= p.E) return;
    const DevTables* gtab = p.tab;
    uint8_t* wl = smem + sub * p.env_lds;          // wave-private: the one-hot counter words or the appearance table
    if constexpr (ONEHOT) {
        uint32_t* wd = reinterpret_cast<uint32_t*>(wl);
        wd[lane] = reinterpret_cast<const uint32_t*>(gtab->delta)[lane];
        wd[lane + 64] = reinterpret_cast<const uint32_t*>(gtab->delta)[lane + 64];
    } else {
        double* wa = reinterpret_cast<double*>(wl);
        for (int i = lane; i < SGW_MAX_TYPES * SGW_MAX_CHANNELS; i += 64) wa[i] = reinterpret_cast<const double*>(gtab->appearance)[i];
    }
    const uint32_t* wdelta = reinterpret_cast<const uint32_t*>(wl);
    const double(*wapp)[SGW_MAX_CHANNELS] = reinterpret_cast<const double(*)[SGW_MAX_CHANNELS]>(wl);
    const int H = p.H, W = p.W, HW = H * W, L = p.L, C = p.C, V = p.V, VV = p.VV, r = p.r;
    uint8_t* g = p.grid + env * p.env_stride;
    const bool mover = p.do_move && p.a0 < p.a1;
    const int ra = p.obs_next ? p.a1 : p.a0;                                   // the agent whose window is rendered
    const bool render = p.obs_next ? p.a1 < p.A : (!(p.flags & SGW_STEP_NO_OBS) && p.a0 < p.a1);
    const bool after = p.obs_next != 0;                                        // it sees the grid AFTER the move

    // ---- the move: decided here from reads only (wave-uniform); its writes come LAST, behind every gather load -- an
    // observation of the mover itself (the plain per-agent step) is the grid BEFORE the move
    uint32_t old_cell = 0xFFFFFFFFu, new_cell = 0xFFFFFFFFu, my_type = 0;      // changed cells (offsets in the agent layer), if it moved
    uint32_t new_pos = 0;
    double val = 0.0;
    int st = 0;
    if (mover) {
        const int a = p.a0;
        uint32_t yx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + a];
        if ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W) { yx = 0; st |= SGW_STATUS_BAD_POS; }
        const int y = (int)(yx & 0xFFu), x = (int)(yx >> 8);
        const uint32_t act = p.actions[env * p.A + a];
        my_type = p.agent_state ? p.agent_state[env * p.A + a] : gtab->agent_type[a];
        const bool act_ok = act < (uint32_t)p.nact;
        const int dy = act_ok ? (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1 : 0;
        const int dx = act_ok ? (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1 : 0;
        const int ty = y + dy, tx = x + dx;
        const bool inb = act_ok && (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
        const uint32_t t = inb ? (uint32_t)g[p.zA * HW + ty * W + tx] : 0xFFu;
        const bool tok = inb && t < (uint32_t)p.T;
        val = tok ? gtab->value[t & 31u] : 0.0;                                // reward read BEFORE the move
        const bool pass = tok && ((p.pass_mask >> (t & 31u)) & 1u);
        st |= !act_ok ? SGW_STATUS_BAD_ACTION : (!inb ? SGW_STATUS_OOB_MOVE : (!tok ? SGW_STATUS_BAD_TYPE : 0));
        if (pass) {
            old_cell = (uint32_t)(y * W + x);
            new_cell = (uint32_t)(ty * W + tx);
            new_pos = (uint32_t)ty | ((uint32_t)tx << 8);
        }
    }
    auto commit = [&]() {
        if (!mover) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // every gather load of this wave has returned
        if (lane == 0) {
            if (new_cell != 0xFFFFFFFFu) {
                g[p.zA * HW + new_cell] = (uint8_t)my_type;
                g[p.zA * HW + old_cell] = (uint8_t)p.default_type;
                reinterpret_cast<uint16_t*>(p.pos)[env * p.A + p.a0] = (uint16_t)new_pos;
            }
            p.rewards[env * p.A + p.a0] = (float)val;
            p.total[env] += val;                                               // float64, agent order (agent.py:172)
            if (st) atomicOr(p.status, st);
        }
    };
    if (!render) {
        commit();
        return;
    }

    // ---- the window of agent `ra` (visual_field.py:9-101): lane = window cell, bytes straight from global memory
    uint32_t pyx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + ra];   // ra != the mover when `after` (ra = a1 > a0)
    if ((pyx & 0xFFu) >= (uint32_t)H || (pyx >> 8) >= (uint32_t)W) {
        pyx = 0;
        if (lane == 0) atomicOr(p.status, SGW_STATUS_BAD_POS);
    }
    const int y = (int)(pyx & 0xFFu), x = (int)(pyx >> 8);
    const int64_t obase = ((env * p.A + ra) * (int64_t)C) * VV;
    constexpr int NW = 4;
    const int nw = (C + 3) >> 2;
    gsync<1>();                                                                // table words visible to every lane
    for (int w = lane; w < VV; w += 64) {
        const int i = w / V, j = w - i * V;
        const int gy = y - r + i, gx = x - r + j;
        const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const uint32_t cellz = (uint32_t)(gy * W + gx);
        if constexpr (ONEHOT) {
            uint32_t cnt[NW] = {0u, 0u, 0u, 0u};
            if (inb) {
                for (int z = 0; z < L; ++z) {
                    uint32_t t = g[z * HW + cellz];
                    if (after && z == p.zA) {                                  // the move, applied to the gathered byte
                        if (cellz == old_cell) t = p.default_type;
                        if (cellz == new_cell) t = my_type;
                    }
                    t &= 31u;
#pragma unroll
                    for (int q = 0; q < NW; ++q)
                        if (q < nw) cnt[q] += wdelta[q * 32 + t];
                }
            } else {
#pragma unroll
                for (int q = 0; q < NW; ++q) cnt[q] = p.fill_delta[q];
            }
#pragma unroll
            for (int q = 0; q < NW; ++q) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int c = 4 * q + b;
                    if (c < C) {
                        const uint32_t v = (cnt[q] >> (8 * b)) & 0xFFu;
                        if (p.obs_u8) reinterpret_cast<uint8_t*>(p.obs)[obase + c * VV + w] = (uint8_t)v;
                        else p.obs[obase + c * VV + w] = (float)v;
                    }
                }
            }
        } else {
            uint32_t tz[SGW_MAX_LAYERS];
            if (inb) {
                for (int z = 0; z < L; ++z) {
                    uint32_t t = g[z * HW + cellz];
                    if (after && z == p.zA) {
                        if (cellz == old_cell) t = p.default_type;
                        if (cellz == new_cell) t = my_type;
                    }
                    tz[z] = t & 31u;
                }
            }
            for (int c = 0; c < C; ++c) {
                double acc;
                if (inb) {   // np.sum over layers: left to right, float64 (visual_field.py:51)
                    acc = wapp[tz[0]][c];
                    for (int z = 1; z < L; ++z) acc += wapp[tz[z]][c];
                } else {
                    acc = wapp[p.fill_type][c];
                }
                p.obs[obase + c * VV + w] = obs_finish(acc, p.obs_post);
            }
        }
    }
    commit();
}

// ---------------------------------------------------------------- reset kernel
// create_world + populate_environment (gridworld.py:47-65, treasurehunt/env.py:114-147).
template <int WPE>
__global__ __launch_bounds__(kBlock) void reset_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int G = WPE * kWave;
    constexpr int EPB = kBlock / G;
    const int tid = threadIdx.x;
    const int sub = tid / G;
    const int gtid = tid - sub * G;
    {
        const uint4* s = reinterpret_cast<const uint4*>(p.tab);
        uint4* d = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < (p.tab_bytes >> 4); i += kBlock) d[i] = s[i];
    }
    __syncthreads();
    const DevTables* tab = reinterpret_cast<const DevTables*>(smem);
    uint8_t* slice = smem + p.tab_bytes + sub * p.env_lds;
    uint8_t* lg = slice;
    uint8_t* s_pos = slice + p.cells_pad + kPosOff;
    const int HW = p.H * p.W;
    const int zoff = p.zA * HW;

    for (int64_t env = (int64_t)blockIdx.x * EPB + sub; env < p.E; env += (int64_t)gridDim.x * EPB) {
        const uint32_t env_id = p.first_env + (uint32_t)env;
        // layers: fill + border
        for (int i = gtid; i < p.cells_pad; i += G) {
            uint8_t v = 0xFF;
            if (i < p.cells) {
                const int z = i / HW;
                const int rem = i - z * HW;
                const int y = rem / p.W, x = rem - y * p.W;
                v = tab->layer_fill[z];
                const uint8_t b = tab->layer_border[z];
                if (b != SGW_NO_BORDER && (y == 0 || y == p.H - 1 || x == 0 || x == p.W - 1)) v = b;
            }
            lg[i] = v;
        }
        gsync<WPE>();
        // optional dense pre-seeding of the agent layer's interior
        if (p.dense_count > 0 && p.dense_thr > 0) {
            const int d0 = zoff >> 2, d1 = (zoff + HW + 3) >> 2;
            for (int d = d0 + gtid; d < d1; d += G) {
                const U4 w = philox4x32_10((uint32_t)d, 0u, env_id, (p.epoch << 4) | SGW_STREAM_DENSE, p.seed_lo, p.seed_hi);
                uint32_t hits = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = 4 * d + b - zoff;
                    if (i < 0 || i >= HW) continue;
                    const int y = i / p.W, x = i - y * p.W;
                    if (y == 0 || y == p.H - 1 || x == 0 || x == p.W - 1) continue;
                    if ((uint64_t)word_of(w, b) < p.dense_thr) hits |= 1u << b;
                }
                if (hits == 0) continue;
                const U4 k = philox4x32_10((uint32_t)d, 0u, env_id, (p.epoch << 4) | SGW_STREAM_DENSE_KIND, p.seed_lo, p.seed_hi);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if ((hits >> b) & 1u)
                        lg[4 * d + b] = tab->dense_choice[(uint32_t)(((uint64_t)word_of(k, b) * (uint32_t)p.dense_count) >> 32)];
            }
            gsync<WPE>();
        }
        // agent placement: sequential sampling without replacement, done by the
        // lanes of the first wave (lane a draws its own u32; lane 0 resolves)
        uint32_t u = 0;
        if (gtid < p.A) {
            const U4 w = philox4x32_10((uint32_t)gtid >> 2, 0u, env_id, (p.epoch << 4) | SGW_STREAM_PLACE, p.seed_lo, p.seed_hi);
            u = word_of(w, gtid & 3);
        }
        uint32_t* s_u = reinterpret_cast<uint32_t*>(slice + p.cells_pad + kRewOff);
        if (gtid < p.A) s_u[gtid] = u;
        gsync<WPE>();
        if (gtid == 0) {
            const int n = (p.H - 2) * (p.W - 2);
            const int iw = p.W - 2;
            // s_u[j], j < i, is reused as the ascending list of taken indices
            for (int i = 0; i < p.A; ++i) {
                int d = (int)(((uint64_t)s_u[i] * (uint32_t)(n - i)) >> 32);
                for (int j = 0; j < i; ++j)
                    if (d >= (int)s_u[j]) ++d;
                int j = i;
                while (j > 0 && (int)s_u[j - 1] > d) {
                    s_u[j] = s_u[j - 1];
                    --j;
                }
                s_u[j] = (uint32_t)d;
                const int y = 1 + d / iw, x = 1 + d - (d / iw) * iw;
                s_pos[2 * i] = (uint8_t)y;
                s_pos[2 * i + 1] = (uint8_t)x;
                lg[zoff + y * p.W + x] = p.agent_state ? p.agent_state[env * p.A + i] : tab->agent_type[i];
            }
            p.total[env] = 0.0;
        }
        gsync<WPE>();
        store_grid<G>(p, p.grid + env * p.env_stride, lg, gtid);
        if (gtid < p.A)
            reinterpret_cast<uint16_t*>(p.pos)[env * p.A + gtid] = reinterpret_cast<const uint16_t*>(s_pos)[gtid];
        gsync<WPE>();
    }
}

// ---------------------------------------------------------------- small kernels
__global__ void random_actions_kernel(const Params p) {
    const int64_t n = p.E * p.A;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t env = i / p.A;
        const int a = (int)(i - env * p.A);
        const U4 w = philox4x32_10((uint32_t)a >> 2, p.turn, p.first_env + (uint32_t)env,
                                   (p.epoch << 4) | SGW_STREAM_ACTION, p.seed_lo, p.seed_hi);
        p.actions[i] = (uint8_t)(((uint64_t)word_of(w, a & 3) * (uint32_t)p.nact) >> 32);
    }
}

// sgw_init_agent_state: configured types; Tag draws the initial "it" agent of every env
__global__ void init_agent_state_kernel(const Params p) {
    for (int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; env < p.E; env += (int64_t)gridDim.x * blockDim.x) {
        uint8_t* st = p.agent_state + env * p.A;
        if (p.agent_rule == SGW_AGENT_RULE_TAG) {
            const U4 w = philox4x32_10(0u, 0u, p.first_env + (uint32_t)env, SGW_STREAM_TAG_INIT, p.seed_lo, p.seed_hi);
            const uint32_t it = __umulhi(w.x, (uint32_t)p.A);
            for (int a = 0; a < p.A; ++a) st[a] = (uint8_t)((uint32_t)a == it ? p.tag_it : p.tag_notit);
        } else {
            for (int a = 0; a < p.A; ++a) st[a] = p.tab->agent_type[a];
        }
    }
}

constexpr int kRedBlocks = 256;

// stage 1: block b sums elements b*256+t, stride 65536, in a fixed order
__global__ __launch_bounds__(kBlock) void reduce_stage1(const double* __restrict__ x, int64_t n, double* __restrict__ part) {
    __shared__ double s[kBlock], s2[kBlock];
    double a = 0.0, a2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)kRedBlocks * kBlock) {
        const double v = x[i];
        a += v;
        a2 += v * v;
    }
    s[threadIdx.x] = a;
    s2[threadIdx.x] = a2;
    __syncthreads();
    for (int k = kBlock / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            s[threadIdx.x] += s[threadIdx.x + k];
            s2[threadIdx.x] += s2[threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x] = s[0];
        part[kRedBlocks + blockIdx.x] = s2[0];
    }
}

__global__ __launch_bounds__(kBlock) void reduce_stage2(const double* __restrict__ part, int64_t n, double* __restrict__ out) {
    __shared__ double s[kBlock], s2[kBlock];
    s[threadIdx.x] = part[threadIdx.x];
    s2[threadIdx.x] = part[kRedBlocks + threadIdx.x];
    __syncthreads();
    for (int k = kBlock / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            s[threadIdx.x] += s[threadIdx.x + k];
            s2[threadIdx.x] += s2[threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = s[0];
        out[1] = s2[0];
        out[2] = (double)n;
        out[3] = 0.0;
    }
}
static_assert(kRedBlocks == kBlock, "stage 2 assumes one partial per thread");

// ---------------------------------------------------------------- host side
thread_local char g_err[512] = "";

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

}  // namespace

struct sgw_engine {
    sgw_config cfg;
    Params base;          // everything except per-call fields
    DevTables* d_tab = nullptr;
    int* d_status = nullptr;
    double* d_part = nullptr;
    int obs_format = SGW_OBS_F32;
    uint8_t* agent_state = nullptr;    // caller-owned, bound with sgw_bind_agent_state
    uint8_t* state_at_pov = nullptr;
    uint8_t* agent_dir = nullptr;      // caller-owned, bound with sgw_bind_agent_dir
    int wpe = 1;          // waves per env
    int group = 64;       // generic step kernel: threads per env (16 / 32: several envs share a wave)
    bool onehot = true;
    bool fast = false;    // step_fast specialisation applies
    bool big = false;     // step_big (workgroup per env, pipelined agents) applies
    void (*step_fn)(const Params) = nullptr;
    void (*step_fn_plain)(const Params) = nullptr;   // run-time-shape STAGE kernels: the direct-store variant for calls that cannot be staged
    void (*step_fn_multi)(const Params) = nullptr;   // step_fast<..., MULTI>: sgw_rollout's turns in one launch
    const char* kernel_name_multi = "-";
    const char* kernel_name_plain = "?";
    int stage_agents = 0;      // agents per staged chunk (STAGE kernels)
    bool phase_ok = false;     // the phase kernel applies (plain moves)
    bool multi_turn = false;   // the step kernel in use runs sgw_rollout's turns in one launch
    void (*reset_fn)(const Params) = nullptr;
    size_t lds_bytes = 0;       // reset / generic step
    size_t step_lds_bytes = 0;  // step kernel actually launched
    bool fast_rules = false;   // the RULES variant of step_fast applies
    int step_env_lds = 0;
    int obs_stage = 0;     // bytes of LDS observation staging per wave (step_fast, one-hot)
    int fast_tab_bytes = 0;
    int grid_blocks = 1;
    int fast_wg_cap = 5;   // step_fast workgroups per CU when writing large float32 observations of a large batch (0: no cap)
    int wg_per_cu = 0;     // sgw_set_wg_per_cu: 0 = the automatic rule above, 1..8 = forced, -1 = never capped
    const char* kernel_name = "?";
    uint32_t auto_max_turns = 0;       // sgw_set_auto_reset
    double* episode_return = nullptr;  // caller-owned
    int reset_blocks = 1;
    int num_cus = 256;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev0, ev1;
    int ev_used = 0;
    double ms_acc = 0.0;
    int64_t launches = 0;
    std::vector<float> series;   // per-launch durations since the last sgw_get_step_times_ms
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

template <typename K>
int occupancy_blocks(K kernel, size_t lds, int num_cus, int* out) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds));
    if (per_cu < 1) per_cu = 1;
    *out = per_cu * num_cus;
    return SGW_OK;
}

using StepFn = void (*)(const Params);

#define PICK(...)                 \
    do {                          \
        *name = #__VA_ARGS__;     \
        return __VA_ARGS__;       \
    } while (0)

template <int G>
StepFn pick_step_g(bool onehot, int L, int C, int rule, const char** name) {
    if (rule == SGW_AGENT_RULE_CLEANUP) {
        if (onehot) PICK(step_kernel<G, true, 0, 0, SGW_AGENT_RULE_CLEANUP>);
        PICK(step_kernel<G, false, 0, 0, SGW_AGENT_RULE_CLEANUP>);
    }
    if (rule == SGW_AGENT_RULE_TAG) {
        if (onehot && L == 1 && C == 4) PICK(step_kernel<G, true, 1, 4, SGW_AGENT_RULE_TAG>);   // the Tag example's tables
        if (onehot) PICK(step_kernel<G, true, 0, 0, SGW_AGENT_RULE_TAG>);
        PICK(step_kernel<G, false, 0, 0, SGW_AGENT_RULE_TAG>);
    }
    if (onehot && L == 2 && C == 6) PICK(step_kernel<G, true, 2, 6>);                             // Treasurehunt-shaped tables
    if (onehot) PICK(step_kernel<G, true>);
    PICK(step_kernel<G, false>);
}

StepFn pick_step(int group, bool onehot, int L, int C, int rule, const char** name) {
    if (group == 16) return pick_step_g<16>(onehot, L, C, rule, name);
    if (group == 32) return pick_step_g<32>(onehot, L, C, rule, name);
    if (group == 64) return pick_step_g<64>(onehot, L, C, rule, name);
    return pick_step_g<256>(onehot, L, C, rule, name);
}
// the MULTI (turn-loop) instantiations of step_fast that sgw_rollout launches; nullptr: no such variant, the rollout is
// a loop of single-turn launches
StepFn pick_fast_multi(bool onehot, int L, int C, int r, int H, int W, bool tag, bool rules, bool stage, const char** name) {
    if (onehot && rules && stage) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true, true, true>);    // layered rule sets (Cleanup)
    if (!onehot || tag || rules || L != 2 || C != 6) return nullptr;
    if (r == 3 && H == 32 && W == 32) PICK(step_fast<true, 2, 6, 3, 32, 32, false, false, false, true>);
    if (r == 2 && H == 16 && W == 16) PICK(step_fast<true, 2, 6, 2, 16, 16, false, false, false, true>);
    if (stage) PICK(step_fast<true, 2, 6, 0, 0, 0, false, false, true, true>);
    return nullptr;
}

StepFn pick_reset(int wpe) { return wpe == 1 ? reset_kernel<1> : reset_kernel<4>; }

StepFn pick_big(bool onehot, int L, int C, int r, const char** name) {
    if (!onehot) PICK(step_big<false, 0, 0, 0>);
    if (L == 2 && C == 6 && r == 5) PICK(step_big<true, 2, 6, 5>);   // BASELINE config 5
    PICK(step_big<true, 0, 0, 0>);
}

bool fixed_fast_shape(int L, int C, int r, int H, int W) {
    return L == 2 && C == 6 && ((r == 3 && H == 32 && W == 32) || (r == 2 && H == 16 && W == 16));
}

StepFn pick_big_multi(bool onehot, int L, int C, int r, const char** name) {
    if (!onehot) PICK(step_big<false, 0, 0, 0, true>);
    if (L == 2 && C == 6 && r == 5) PICK(step_big<true, 2, 6, 5, true>);
    PICK(step_big<true, 0, 0, 0, true>);
}

StepFn pick_fast(bool onehot, int L, int C, int r, int H, int W, bool tag, bool rules, bool stage, const char** name) {
    if (rules) {
        if (onehot && stage) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true, true>);
        if (onehot) PICK(step_fast<true, 0, 0, 0, 0, 0, false, true>);
        PICK(step_fast<false, 0, 0, 0, 0, 0, false, true>);
    }
    if (tag) {
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
    if (stage) PICK(step_fast<true, 0, 0, 0, 0, 0, false, false, true>);
    PICK(step_fast<true, 0, 0, 0, 0, 0>);
}

int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr size_t kSeriesCap = (size_t)1 << 20;

// Waits for the recorded event pairs and folds them into the running sum and the per-launch series.
int time_drain(sgw_engine* e) {
    if (e->ev_used == 0) return SGW_OK;
    HIP_TRY(hipEventSynchronize(e->ev1[e->ev_used - 1]));
    for (int i = 0; i < e->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->ev0[i], e->ev1[i]));
        e->ms_acc += ms;
        if (e->series.size() < kSeriesCap) e->series.push_back(ms);
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

const char* sgw_version(void) { return "sgw 0.1 (gfx950)"; }

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
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const sgw_config& c = e->cfg;

    // ---- tables
    DevTables h;
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

    // ---- static launch parameters
    Params& p = e->base;
    memset(&p, 0, sizeof(p));
    p.H = c.height; p.W = c.width; p.L = c.layers; p.A = c.num_agents; p.r = c.vision_radius;
    p.V = 2 * c.vision_radius + 1; p.VV = p.V * p.V; p.C = c.num_channels; p.T = c.num_types;
    p.nact = c.num_actions; p.zA = c.agent_layer;
    p.cells = c.layers * c.height * c.width;
    p.cells_pad = (p.cells + 15) & ~15;
    p.env_stride = c.grid_env_stride > 0 ? c.grid_env_stride : p.cells;
    p.env_lds = p.cells_pad + kAgentLds;
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
    p.nturns = 1;
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
    p.has_become = 0;
    p.become_mask = 0;
    for (int t = 0; t < c.num_types; ++t)
        if (c.type_rule[t] == SGW_RULE_BECOME_IF) {
            p.has_become = 1;
            p.become_mask |= 1u << t;
        }
    p.kind_pack = 0;
    for (int a = 0; a < c.num_actions; ++a) p.kind_pack |= (uint32_t)(c.action_kind[a] & 3u) << (2 * a);
    p.beam_radius = c.beam_radius;
    p.clean_beam = c.clean_beam_type;
    p.zap_beam = c.zap_beam_type;
    p.beam_block_mask = c.beam_block_mask;
    p.total_factor = c.reward_total_factor > 0 ? c.reward_total_factor : 1;

    // ---- group geometry: one wave per env while a slice stays small, else a workgroup per env
    e->wpe = (p.cells_pad <= 4096) ? 1 : 4;
    const int epb = kBlock / (e->wpe * kWave);
    e->lds_bytes = (size_t)p.tab_bytes + (size_t)epb * p.env_lds;
    const bool plain_move = c.agent_rule == SGW_AGENT_RULE_MOVE;   // step_big implements MovingAgent.act only
    const bool simple_rules = !p.has_become && c.agent_rule != SGW_AGENT_RULE_CLEANUP;   // else: generic kernel
    const bool vec16 = (p.env_stride & 15) == 0 && p.env_stride >= p.cells_pad;   // 16-byte loads/stores per env are legal
    e->fast = e->wpe == 1 && vec16 && (p.cells_pad >> 4) <= 64 * kMaxUnits && nspawn <= 1 && p.VV <= 128 && simple_rules;   // MovingAgent.act and TagAgent.act
    // the layered rule set on the wave-per-env kernel (RULES variant): any spawners, BECOME_IF rules, Cleanup or plain agents
    e->fast_rules = !e->fast && e->wpe == 1 && vec16 && (p.cells_pad >> 4) <= 64 * kMaxUnits && p.VV <= 128 &&
                    c.agent_rule != SGW_AGENT_RULE_TAG;
    if (const char* f = getenv("SGW_NO_FAST_RULES")) { if (f[0] == '1') e->fast_rules = false; }   // test hook: generic kernel instead
    e->fast = e->fast || e->fast_rules;
    // fast kernel: wave-private LDS = [one-hot counter words | appearance table][grid]
    e->fast_tab_bytes = onehot ? 4 * SGW_MAX_TYPES * 4 : SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8;
    bool agents_impassable = true;
    for (int a = 0; a < c.num_agents; ++a) agents_impassable = agents_impassable && !c.type_passable[c.agent_type[a]];
    e->big = e->wpe == 4 && vec16 && nspawn <= 1 && p.VV <= 128 && agents_impassable && plain_move && simple_rules;
    bool stage_kernel = false;   // a run-time-shape STAGE kernel applies
    {   // LDS staging of one-hot observations
        const int ob_elems = c.num_agents * c.num_channels * p.VV;
        const int per_agent = c.num_channels * p.VV;
        const bool fixed_shape = !e->fast_rules && c.agent_rule != SGW_AGENT_RULE_TAG &&
                                 fixed_fast_shape(c.layers, c.num_channels, c.vision_radius, c.height, c.width);   // = pick_fast's fixed-shape kernels
        e->obs_stage = 0;
        if (e->fast && onehot && fixed_shape) {
            // whole envs of a multiple of 4 elements, at most 4 KiB of byte counts
            if ((ob_elems & 3) == 0 && ob_elems <= 4096) e->obs_stage = (ob_elems + 15) & ~15;
        } else if (e->fast && onehot) {
            // run-time shapes: as many agents per burst as fit the wave's share of LDS at full occupancy (8 workgroups
            // of 4 waves per CU, 1 KiB granules: 5 120 bytes per wave); if not even one agent fits, at 5 workgroups per CU
            const int base = e->fast_tab_bytes + (e->fast_rules ? kRuleLds : 0) + p.cells_pad;
            int budget = (int)(kLdsPerCu / 8 / 4) - base - 16;
            if (budget < per_agent) budget = (int)((kLdsPerCu / 5 - 1024) / 4) - base - 16;
            if (const char* f = getenv("SGW_STAGE_BYTES")) budget = atoi(f);                 // A/B hook
            int apc = budget >= per_agent ? std::min(c.num_agents, budget / per_agent) : 0;
            if (const char* f = getenv("SGW_STAGE_AGENTS")) apc = std::min(c.num_agents, atoi(f));   // A/B hook
            if (apc > 0) {
                e->stage_agents = apc;
                e->obs_stage = (apc * per_agent + 3 + 15) & ~15;    // + 3: the chunk's misalignment in global memory
                stage_kernel = true;
            }
        }
        if (const char* f = getenv("SGW_NO_STAGE")) { if (f[0] == '1') { e->obs_stage = 0; stage_kernel = false; } }   // test / tuning hook
    }
    if (const char* f = getenv("SGW_FORCE_GENERIC")) {   // test hook: exercise the generic kernel on shapes the specialised ones would take
        if (f[0] == '1') e->fast = e->big = e->fast_rules = false;
    }
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
    // against 16-19 us packed).  SGW_GROUP = 16 / 32 forces a packing, 64 forbids it (A/B hook).
    e->group = e->wpe * kWave;
    if (e->wpe == 1) {
        const int64_t avv = (int64_t)c.num_agents * p.VV;
        auto enough = [&](int G) { return c.num_agents <= G && (int64_t)c.num_envs * G / kWave >= 12288; };
        int g = 0;
        if (c.agent_rule == SGW_AGENT_RULE_TAG) g = enough(32) ? 32 : 0;
        else if (c.agent_rule == SGW_AGENT_RULE_MOVE && !p.has_become) {
            if (avv <= 100 && p.cells_pad <= 1024 && enough(16)) g = 16;
            else if (avv <= 200 && enough(32)) g = 32;
        }
        if (const char* f = getenv("SGW_GROUP")) g = atoi(f);
        const bool fits = (g == 16 || g == 32) && c.num_agents <= g &&
                          (c.agent_rule != SGW_AGENT_RULE_CLEANUP || 3 * c.beam_radius <= g);
        if (fits) {
            e->group = g;
            e->fast = e->fast_rules = false;
        }
    }
    if (!e->fast) { e->obs_stage = 0; stage_kernel = false; }
    const int epb_step = (e->fast || e->big) ? epb : kBlock / e->group;   // envs per workgroup of the step kernel
    e->step_env_lds = e->fast ? e->fast_tab_bytes + (e->fast_rules ? kRuleLds : 0) + p.cells_pad + e->obs_stage : p.env_lds;
    e->step_lds_bytes = e->fast ? (size_t)epb * e->step_env_lds : (size_t)p.tab_bytes + (size_t)epb_step * e->step_env_lds;
    p.big_pitch = c.width;
    if (e->big) {
        // padded LDS rows (bank-conflict-free window gather) where a row is whole 16-byte units and the image still
        // leaves four workgroups per CU (LDS is handed out in 1 KiB granules)
        const size_t fixed = (size_t)e->fast_tab_bytes + kBigAgentLds;
        const size_t padded = fixed + (size_t)c.layers * c.height * (c.width + 16);
        if ((c.width & 15) == 0 && (p.cells & 15) == 0 && ((padded + 1023) & ~(size_t)1023) * 4 <= kLdsPerCu) p.big_pitch = c.width + 16;
        if (const char* f = getenv("SGW_BIG_NO_PAD")) { if (f[0] == '1') p.big_pitch = c.width; }   // A/B hook
        e->step_lds_bytes = fixed + (p.big_pitch == c.width ? (size_t)p.cells_pad : (size_t)c.layers * c.height * p.big_pitch);
    }
    const size_t lds_cap = prop.sharedMemPerBlock > 0 ? prop.sharedMemPerBlock : 65536;
    const size_t lds_max = 160 * 1024;
    if (e->lds_bytes > lds_max) {
        delete e;
        return fail(SGW_EINVAL, "world of %d bytes per env does not fit the %zu-byte LDS-resident path", p.cells, lds_max);
    }

    hipError_t err = hipMalloc(&e->d_tab, sizeof(DevTables));
    if (err == hipSuccess) err = hipMemcpy(e->d_tab, &h, sizeof(DevTables), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMalloc(&e->d_status, 4 * sizeof(int));
    if (err == hipSuccess) err = hipMemset(e->d_status, 0, 4 * sizeof(int));
    if (err == hipSuccess) err = hipMalloc(&e->d_part, 2 * kRedBlocks * sizeof(double));
    if (err != hipSuccess) {
        sgw_destroy(e);
        return fail(SGW_EHIP, "device allocation failed: %s", hipGetErrorString(err));
    }
    p.tab = e->d_tab;
    p.status = e->d_status;

    if (e->fast) e->step_fn_plain = pick_fast(e->onehot, c.layers, c.num_channels, c.vision_radius, c.height, c.width, c.agent_rule == SGW_AGENT_RULE_TAG, e->fast_rules, false, &e->kernel_name_plain);
    // Worlds above 4 KiB only: there, gathering one window from global memory beats staging 32 KiB through LDS (config 5:
    // 14.9 against 24.4 us per phase launch); a 2 KiB env is staged with four coalesced 16-byte loads per lane and the
    // byte gather from global is the slower way (config 3: 62.9 against 46.1 us).  SGW_NO_PHASE_KERNEL = 1 / 0 forces.
    e->phase_ok = c.agent_rule == SGW_AGENT_RULE_MOVE && e->wpe == 4;
    if (const char* f = getenv("SGW_NO_PHASE_KERNEL")) e->phase_ok = c.agent_rule == SGW_AGENT_RULE_MOVE && f[0] == '0';
    p.stage_agents = e->stage_agents;
    StepFn sk = e->fast  ? pick_fast(e->onehot, c.layers, c.num_channels, c.vision_radius, c.height, c.width, c.agent_rule == SGW_AGENT_RULE_TAG, e->fast_rules, stage_kernel, &e->kernel_name)
                : e->big ? pick_big(e->onehot, c.layers, c.num_channels, c.vision_radius, &e->kernel_name)
                         : pick_step(e->group, e->onehot, c.layers, c.num_channels, c.agent_rule, &e->kernel_name);
    StepFn rk = pick_reset(e->wpe);
    if (const char* f = getenv("SGW_FAST_WG_PER_CU")) e->fast_wg_cap = atoi(f);   // tuning hook
    e->step_fn = sk;
    e->reset_fn = rk;
    if (e->fast)
        e->step_fn_multi = pick_fast_multi(e->onehot, c.layers, c.num_channels, c.vision_radius, c.height, c.width,
                                           c.agent_rule == SGW_AGENT_RULE_TAG, e->fast_rules, stage_kernel, &e->kernel_name_multi);
    if (e->big) e->step_fn_multi = pick_big_multi(e->onehot, c.layers, c.num_channels, c.vision_radius, &e->kernel_name_multi);
    e->multi_turn = (e->fast || e->big) ? e->step_fn_multi != nullptr : true;   // kernels with sgw_rollout's turn loop
    if (std::max(e->lds_bytes, e->step_lds_bytes) > std::min<size_t>(lds_cap, 65536)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds_bytes);
        if (err == hipSuccess && e->step_fn_plain && e->step_fn_plain != sk)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(e->step_fn_plain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds_bytes);
        if (err == hipSuccess && e->step_fn_multi)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(e->step_fn_multi), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->step_lds_bytes);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(rk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
        if (err != hipSuccess) {
            sgw_destroy(e);
            return fail(SGW_EHIP, "cannot reserve %zu bytes of LDS: %s", e->lds_bytes, hipGetErrorString(err));
        }
    }
    int nb = 0;
    if (int rc = occupancy_blocks(sk, e->step_lds_bytes, e->num_cus, &nb)) { sgw_destroy(e); return rc; }
    // generic kernel: persistent grid; fast kernel: one env per wave, the dispatcher balances
    (void)nb;
    e->grid_blocks = (int)ceil_div(p.E, (e->fast || e->big) ? epb : epb_step);   // every step kernel: one env per group, the dispatcher balances
    if (int rc = occupancy_blocks(rk, e->lds_bytes, e->num_cus, &nb)) { sgw_destroy(e); return rc; }
    e->reset_blocks = (int)std::min<int64_t>(ceil_div(p.E, epb), nb);
    *out = e;
    return SGW_OK;
}

void sgw_destroy(sgw_engine* e) {
    if (!e) return;
    for (hipEvent_t ev : e->ev0) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev1) (void)hipEventDestroy(ev);
    if (e->d_tab) (void)hipFree(e->d_tab);
    if (e->d_status) (void)hipFree(e->d_status);
    if (e->d_part) (void)hipFree(e->d_part);
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

static int launch_step(sgw_engine* e, Params& p, hipStream_t s) {
    if (int rc = time_begin(e, s)) return rc;
    p.agent_state = e->agent_state;
    p.state_at_pov = e->state_at_pov;
    p.agent_dir = e->agent_dir;
    if (p.agent_rule == SGW_AGENT_RULE_CLEANUP && p.do_move && !p.agent_dir)
        return fail(SGW_EINVAL, "SGW_AGENT_RULE_CLEANUP needs sgw_bind_agent_dir");
    p.obs_u8 = e->obs_format == SGW_OBS_U8 ? 1 : 0;
    if (p.agent_rule == SGW_AGENT_RULE_TAG && p.do_move && !p.agent_state)
        return fail(SGW_EINVAL, "SGW_AGENT_RULE_TAG needs sgw_bind_agent_state");
    p.env_lds = e->step_env_lds;
    if (e->fast || e->big) p.tab_bytes = e->fast_tab_bytes;
    p.obs_stage = (e->fast && p.obs && (reinterpret_cast<uintptr_t>(p.obs) & 15) == 0) ? e->obs_stage : 0;
    if (p.spawn_mask == 0 && !p.has_become) p.flags &= ~SGW_STEP_SWEEP;   // nothing transitions
    // Occupancy cap (an LDS request that fits 5 workgroups per CU = 5 waves per SIMD) for whole-turn float32
    // observation writes of 8 KiB or more per env in large batches, where fewer concurrent waves mean fewer
    // half-written lines open in HBM:
    //  - the unstaged path (Cleanup, 21x31x3 at 65 536 envs: 893 -> 801 us);
    //  - the staged path only when the grids of the batch no longer fit the caches (262 144 envs of config 3:
    //    662 -> 578 us, 524 288: 1375 -> 1146 us).  While they do fit (configs 3/4: 65 536 envs, 134 MB) the staged
    //    emit with its streaming full-line stores is fastest at full occupancy (124 us at 8 and 7 per CU, 126 at 6,
    //    131 at 5; 131 072 envs: 248 vs 281 us).
    // The uint8 format, small batches and the shapes with small windows, which are latency-bound (Tag 11x11, 6.5 KB
    // per env: 164 us at full occupancy, 192 us capped), are not capped.
    // The policy is explicit (sgw_set_wg_per_cu, include/sgw.h): 0 = this automatic rule, 1..8 = forced, -1 = never.
    size_t lds = e->step_lds_bytes;
    int cap = 0;
    if (e->fast && e->wg_per_cu > 0) cap = e->wg_per_cu;
    else if (e->fast && e->wg_per_cu == 0 && e->fast_wg_cap > 0 && p.obs && !(p.flags & SGW_STEP_NO_OBS) && !p.obs_u8 &&
             (!p.obs_stage || (size_t)p.E * (size_t)p.env_stride > kCacheResidentGrid) && p.a1 == p.A && p.a0 == 0 &&
             (size_t)p.A * p.C * p.VV * 4 >= 8192 && p.E >= (int64_t)e->num_cus * 32 * 2)
        cap = e->fast_wg_cap;
    if (cap > 0) lds = std::max(lds, (size_t)(kLdsPerCu / cap - 1024) & ~(size_t)511);   // 1 KiB below the share: LDS is handed out in 1 KiB granules
    // A policy-driven phase (at most one agent moves, at most one window is rendered, no sweep, plain moves): the phase
    // kernel, which does not stage the env (SGW_NO_PHASE_KERNEL=1: A/B and test hook).
    if (e->phase_ok && p.nturns == 1 && !(p.flags & SGW_STEP_SWEEP) && p.a1 - p.a0 <= 1 && (p.do_move || p.a1 - p.a0 == 1)) {
        Params q = p;
        q.env_lds = e->onehot ? 4 * SGW_MAX_TYPES * 4 : SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8;
        hipLaunchKernelGGL(e->onehot ? phase_kernel<true> : phase_kernel<false>, dim3((unsigned)ceil_div(p.E, 4)), dim3(kBlock),
                           (size_t)4 * q.env_lds, s, q);
        HIP_TRY(hipGetLastError());
        return time_end(e, s);
    }
    // a run-time-shape STAGE kernel has no direct-store path: calls it cannot serve take the plain variant
    StepFn fn = e->step_fn;
    if (e->fast && e->stage_agents > 0 && e->step_fn_plain &&
        (p.obs_stage == 0 || p.a0 != 0 || p.a1 != p.A || p.obs_next || (p.flags & SGW_STEP_NO_OBS)))
        fn = e->step_fn_plain;
    if (p.nturns > 1 && (e->fast || e->big)) fn = e->step_fn_multi;   // sgw_rollout made sure it exists and the call qualifies
    hipLaunchKernelGGL(fn, dim3(e->grid_blocks), dim3(e->big ? kBigThreads : kBlock), lds, s, p);
    HIP_TRY(hipGetLastError());
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
    if (flags & SGW_STEP_OBS_NEXT) {   // the stepped agents' own observations are not written
        p.obs_next = 1;
        p.flags |= SGW_STEP_NO_OBS;
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
    *count = n;
    e->series.clear();
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
    snprintf(buf, (size_t)capacity, "%s group=%d threads=%d lds=%zu env_lds=%d obs_stage=%d stage_agents=%d grid=%d wg_per_cu=%s%d", e->kernel_name,
             (e->fast || e->big) ? e->wpe * kWave * (e->big ? kBigWaves / 4 : 1) : e->group,
             e->big ? kBigThreads : kBlock, e->step_lds_bytes, e->step_env_lds, e->obs_stage, e->stage_agents, e->grid_blocks,
             e->wg_per_cu == 0 ? "auto:" : "", e->wg_per_cu == 0 ? e->fast_wg_cap : e->wg_per_cu);
    return SGW_OK;
}

}  // extern "C"

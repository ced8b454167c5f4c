// phase.h -- part of the single translation unit sgw.hip (included inside its anonymous namespace).
// phase_kernel<ONEHOT>: one policy-driven phase of a world above 4 KiB without staging the env.
#pragma once

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
    const int H = p.H, W = p.W, HW = H * W, L = p.L, C = p.C, V = p.V, VV = p.VV, r = p.r;
    uint8_t* g = p.grid + env * p.env_stride;
    const bool mover = p.do_move && p.a0 < p.a1;
    const int ra = p.obs_next ? p.a1 : p.a0;                                   // the agent whose window is rendered
    const bool render = p.obs_next ? p.a1 < p.A : (!(p.flags & SGW_STEP_NO_OBS) && p.a0 < p.a1);
    const bool after = p.obs_next != 0;                                        // it sees the grid AFTER the move

    // ---- level 1: every load that depends on nothing but the env index, issued together
    uint32_t yx = 0, act = 0, my_type = 0, pyx = 0;
    if (mover) {
        yx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + p.a0];
        act = p.actions[env * p.A + p.a0];
        my_type = p.agent_state ? p.agent_state[env * p.A + p.a0] : gtab->agent_type[p.a0];
    }
    if (render) pyx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + ra];   // ra is not the mover when `after` (ra = a1 > a0)
    // wave-private tables: [one-hot counter words | appearance][value f64 x 32]
    uint8_t* wl = smem + sub * p.env_lds;
    constexpr int kTab = ONEHOT ? 4 * SGW_MAX_TYPES * 4 : SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8;
    double* wval = reinterpret_cast<double*>(wl + kTab);
    if constexpr (ONEHOT) {
        uint32_t* wd = reinterpret_cast<uint32_t*>(wl);
        wd[lane] = reinterpret_cast<const uint32_t*>(gtab->delta)[lane];
        wd[lane + 64] = reinterpret_cast<const uint32_t*>(gtab->delta)[lane + 64];
    } else {
        double* wa = reinterpret_cast<double*>(wl);
        for (int i = lane; i < SGW_MAX_TYPES * SGW_MAX_CHANNELS; i += 64) wa[i] = reinterpret_cast<const double*>(gtab->appearance)[i];
    }
    if (lane < SGW_MAX_TYPES) wval[lane] = gtab->value[lane];
    const uint32_t* wdelta = reinterpret_cast<const uint32_t*>(wl);
    const double(*wapp)[SGW_MAX_CHANNELS] = reinterpret_cast<const double(*)[SGW_MAX_CHANNELS]>(wl);

    // ---- level 2: the target byte and the first 64 window cells' bytes (addresses from level 1), issued together
    int st = 0;
    if (mover && ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W)) { yx = 0; st |= SGW_STATUS_BAD_POS; }
    const int my = (int)(yx & 0xFFu), mx = (int)(yx >> 8);
    const bool act_ok = act < (uint32_t)p.nact;
    const int dy = (mover && act_ok) ? (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1 : 0;
    const int dx = (mover && act_ok) ? (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1 : 0;
    const int ty = my + dy, tx = mx + dx;
    const bool tinb = mover && act_ok && (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
    const uint32_t t = tinb ? (uint32_t)g[p.zA * HW + ty * W + tx] : 0xFFu;
    if (render && ((pyx & 0xFFu) >= (uint32_t)H || (pyx >> 8) >= (uint32_t)W)) {
        pyx = 0;
        st |= SGW_STATUS_BAD_POS;
    }
    const int y = (int)(pyx & 0xFFu), x = (int)(pyx >> 8);
    // window cell of this lane in pass `w0 / 64`: in-bounds flag, cell offset in a layer, the L type ids packed 8 bits each
    auto gather = [&](const int w, bool& inb, uint32_t& cellz, uint32_t& lo, uint32_t& hi) {
        const int i = w / V, j = w - i * V;
        const int gy = y - r + i, gx = x - r + j;
        inb = w < VV && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        cellz = (uint32_t)(gy * W + gx);
        lo = hi = 0;
        if (inb)
            for (int z = 0; z < L; ++z) {
                const uint32_t tz = g[z * HW + cellz];
                if (z < 4) lo |= (tz & 31u) << (8 * z);
                else hi |= (tz & 31u) << (8 * (z - 4));
            }
    };
    bool inb0 = false;
    uint32_t cell0 = 0, lo0 = 0, hi0 = 0;
    if (render) gather(lane, inb0, cell0, lo0, hi0);

    // ---- the move: decided from reads only (wave-uniform); its writes come LAST, behind every gather load -- an
    // observation of the mover itself (the plain per-agent step) is the grid BEFORE the move
    gsync<1>();                                                                // table words visible to every lane
    uint32_t old_cell = 0xFFFFFFFFu, new_cell = 0xFFFFFFFFu, new_pos = 0;      // changed cells (offsets in the agent layer), if it moved
    double val = 0.0;
    if (mover) {
        const bool tok = tinb && t < (uint32_t)p.T;
        val = tok ? wval[t & 31u] : 0.0;                                       // reward read BEFORE the move
        const bool pass = tok && ((p.pass_mask >> (t & 31u)) & 1u);
        st |= !act_ok ? SGW_STATUS_BAD_ACTION : (!tinb ? SGW_STATUS_OOB_MOVE : (!tok ? SGW_STATUS_BAD_TYPE : 0));
        if (pass) {
            old_cell = (uint32_t)(my * W + mx);
            new_cell = (uint32_t)(ty * W + tx);
            new_pos = (uint32_t)ty | ((uint32_t)tx << 8);
        }
    }
    auto commit = [&]() {
        if (lane == 0 && st) atomicOr(p.status, st);
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
        }
    };
    if (!render) {
        commit();
        return;
    }

    // ---- the window of agent `ra` (visual_field.py:9-101): lane = window cell
    const int64_t obase = ((env * p.obs_A + (ra - p.obs_a0)) * (int64_t)C) * VV;
    constexpr int NW = 4;
    const int nw = (C + 3) >> 2;
    const int zsh = 8 * (p.zA & 3);
    auto emit = [&](const int w, const bool inb, const uint32_t cellz, uint32_t lo, uint32_t hi) {
        if (after && inb && (cellz == old_cell || cellz == new_cell)) {        // the move, applied to the gathered bytes
            const uint32_t nv = cellz == new_cell ? (my_type & 31u) : (p.default_type & 31u);
            if (p.zA < 4) lo = (lo & ~(0xFFu << zsh)) | (nv << zsh);
            else hi = (hi & ~(0xFFu << zsh)) | (nv << zsh);
        }
        if constexpr (ONEHOT) {
            uint32_t cnt[NW] = {0u, 0u, 0u, 0u};
            if (inb) {
                for (int z = 0; z < L; ++z) {
                    const uint32_t tz = z < 4 ? (lo >> (8 * z)) & 31u : (hi >> (8 * (z - 4))) & 31u;
#pragma unroll
                    for (int q = 0; q < NW; ++q)
                        if (q < nw) cnt[q] += wdelta[q * 32 + tz];
                }
            } else {                                                           // fill entity's appearance, once (visual_field.py:89-94)
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
            for (int c = 0; c < C; ++c) {
                double acc;
                if (inb) {   // np.sum over layers: left to right, float64 (visual_field.py:51)
                    acc = wapp[lo & 31u][c];
                    for (int z = 1; z < L; ++z) acc += wapp[z < 4 ? (lo >> (8 * z)) & 31u : (hi >> (8 * (z - 4))) & 31u][c];
                } else {
                    acc = wapp[p.fill_type][c];
                }
                p.obs[obase + c * VV + w] = obs_finish(acc, p.obs_post);
            }
        }
    };
    if (lane < VV) emit(lane, inb0, cell0, lo0, hi0);
    for (int w = lane + 64; w < VV; w += 64) {                                 // windows wider than 64 cells
        bool inb;
        uint32_t cellz, lo, hi;
        gather(w, inb, cellz, lo, hi);
        emit(w, inb, cellz, lo, hi);
    }
    commit();
}

// ---------------------------------------------------------------- row-load window kernels (round 3)
// Windows of one-hot worlds WITHOUT staging the env, for worlds of ANY size, small ones included (where phase_kernel's
// byte gather -- 49 of 64 lanes, a byte load per cell and layer -- lost to staging the whole env):
//   * a LANE owns a window ROW: G = 4 / 8 / 16 lanes per window (V <= 4 / 8 / 16), so a wave carries 16 / 8 / 4 windows and
//     the instruction stream is shared by that many (a small world's phase is bound by issue and by the number of waves
//     in flight, not by bytes);
//   * a row of a layer arrives with ONE unaligned 8-byte load per 8 columns (global_load_dwordx2 at a byte address; the
//     start is clamped into the env and the value shifted, so that no load ever leaves the env's bytes); columns outside
//     the map are replaced by the fill entity through a byte mask -- no per-cell bounds test;
//   * a lane emits its V consecutive floats per channel with 16-byte stores at 4-byte-aligned addresses.
// L and the number of counter words NW = ceil(C / 4) are compile-time (loops over cells / layers / words fully unrolled:
// everything stays in registers); C, H, W are run-time.
// Two kernels share the pieces below:
//   phase_rows<L, NW, R>     one policy-driven phase: MovingAgent.act of agent a0 and / or ONE window per env (the 1 + A launch
//                            protocol of sgw_step with SGW_STEP_OBS_NEXT; sorrel/agents/agent.py:155-173)
//   observe_rows<L, NW, R>   the windows of a RANGE of agents of every env into per-agent destinations (sgw_observe_rows)
typedef float vf4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float vf2u __attribute__((ext_vector_type(2), aligned(4)));

// per-agent window destinations (sgw_observe_rows / sgw_act): agent a's window of env e starts at p[a] + e * stride elements
struct RowPtrs {
    void* p[SGW_MAX_AGENTS];
    int64_t stride;
    // sgw_act only: where the acting agent's action comes from and where else its outputs go (all optional)
    const void* agent_action;   // [E] of the agent's own actions (the policy's output tensor as it is); NULL: actions[E][A]
    int action_kind;            // SGW_ACT_U8 / _I32 / _I64
    float* reward_row;          // [E]: a second copy of the rewards (the row of the agent's replay buffer)
    int64_t* action_row;        // [E]: the actions as int64 (the row of the agent's replay buffer)
};

__device__ __forceinline__ uint64_t load8_unaligned(const uint8_t* q) {
    uint64_t v;
    __builtin_memcpy(&v, q, 8);
    return v;
}

template <int NW>
struct RowsTab {
    static constexpr int kTabW = 34;                     // counter words per q: 32 types + [32] the fill entity + [33] zero
    static constexpr int kBytes = NW * kTabW * 4;
    // one wave's copy of the counter words (wave-private: no workgroup barrier)
    static __device__ __forceinline__ void fill(uint32_t* wd, const DevTables* gtab, const Params& p, int lane) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            if (lane < 32) wd[q * kTabW + lane] = gtab->delta[q][lane];
            if (lane == 32) wd[q * kTabW + 32] = p.fill_delta[q];
            if (lane == 33) wd[q * kTabW + 33] = 0u;
        }
    }
};

// the window row `gy` of an agent at column x: L x NCH 64-bit values, byte j of chunk ch = type id at column x - R + 8 ch + j
template <int L, int R>
__device__ __forceinline__ void rows_load(uint64_t (&row)[L][(2 * R + 1 + 7) / 8], const uint8_t* g, const bool rowinb, const int gy,
                                          const int x, const int W, const int HW, const int cells) {
    constexpr int NCH = (2 * R + 1 + 7) / 8;
#pragma unroll
    for (int z = 0; z < L; ++z)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            row[z][ch] = 0;
            if (rowinb) {
                const int s = z * HW + gy * W + (x - R) + 8 * ch;             // first byte wanted (may lie outside the env)
                const int sc = min(max(s, 0), cells - 8);                     // first byte loaded
                const uint64_t raw = load8_unaligned(g + sc);
                const int d = min(max(s - sc, -7), 7);                        // |d| > 7: nothing of this chunk is on the map
                row[z][ch] = d >= 0 ? raw >> (8 * d) : raw << (8 * -d);
            }
        }
}

// rows -> per cell (table index) << 2: cells off the map look up the fill entity once (layer 0) and zeros (other layers)
template <int L, int R>
__device__ __forceinline__ void rows_index(uint64_t (&row)[L][(2 * R + 1 + 7) / 8], const bool rowinb, const int x, const int W) {
    constexpr int V = 2 * R + 1, NCH = (V + 7) / 8;
    const int jlo = max(0, R - x), jhi = min(V, W + R - x);                   // columns on the map: j in [jlo, jhi)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int blo = min(max(jlo - 8 * ch, 0), 8), bhi = min(max(jhi - 8 * ch, 0), 8);
        const int n = bhi - blo;
        const uint64_t mask = (rowinb && n > 0) ? ((~0ull >> (64 - 8 * n)) << (8 * blo)) : 0ull;
#pragma unroll
        for (int z = 0; z < L; ++z) {
            const uint64_t off = z == 0 ? 0x2020202020202020ull : 0x2121212121212121ull;
            row[z][ch] = (((row[z][ch] & 0x1F1F1F1F1F1F1F1Full) & mask) | (off & ~mask)) << 2;
        }
    }
}

// Table look-ups and layer sums of the lane's row, then the wave's windows leave TOGETHER: the one-hot counts are staged
// as bytes in the wave's LDS area, window after window in each window's final [C][V][V] order, and streamed out with
// consecutive lanes on consecutive elements.  `mode` (host: rows_mode()):
//   kRowsFlat     the wave's live windows are ONE contiguous, 16-byte aligned run in global memory starting at `o` of the
//                 wave's first group (all agents of consecutive envs in the [E][A][C][V][V] tensor; or consecutive envs of
//                 one agent in that agent's own [E][C][V][V] destination): streaming float4 stores of whole lines, as the
//                 headline kernel's emit -- observe_rows at config 3: 188 us with per-lane 16-byte pieces, 145 with
//                 per-window float2 runs, ... flat;
//   kRowsPair     every destination 8-byte aligned and C*V*V even: per window, float2 runs;
//   kRowsSingle   anything else: per window, single floats.
//   idx     this lane's row (rows_index)          o       the window's destination (the same in every lane of a group)
//   wd      the wave's counter words              stage   the wave's staging area, WPW * C*V*V bytes (+ 3)
//   act     this lane renders a row of a live window
constexpr int kRowsSingle = 1, kRowsPair = 2, kRowsFlat = 4;
template <int L, int NW, int R>
__device__ __forceinline__ void rows_emit(const uint64_t (&idx)[L][(2 * R + 1 + 7) / 8], const uint32_t* wd, uint8_t* stage, float* o,
                                          const int C, const int lane, const bool act, const int mode) {
    constexpr int V = 2 * R + 1, VV = V * V, kTabW = RowsTab<NW>::kTabW;
    constexpr int G = V <= 4 ? 4 : (V <= 8 ? 8 : 16), WPW = 64 / G;
    const int N = C * VV;
    const uint8_t* wdb = reinterpret_cast<const uint8_t*>(wd);
    const int gl = lane & (G - 1);
    uint8_t* mine = stage + (lane / G) * N + gl * V;
    if (act) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            uint32_t cnt[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q) cnt[q] = 0;
#pragma unroll
            for (int z = 0; z < L; ++z) {
                const uint32_t half = (j & 7) < 4 ? (uint32_t)idx[z][j >> 3] : (uint32_t)(idx[z][j >> 3] >> 32);
                const uint32_t t4 = (half >> (8 * (j & 3))) & 0xFFu;
#pragma unroll
                for (int q = 0; q < NW; ++q) cnt[q] += *reinterpret_cast<const uint32_t*>(wdb + q * kTabW * 4 + t4);
            }
#pragma unroll
            for (int q = 0; q < NW; ++q)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int c = 4 * q + b;
                    if (c < C) mine[c * VV + j] = (uint8_t)(cnt[q] >> (8 * b));
                }
        }
    }
    gsync<1>();
    const uint32_t o_lo = (uint32_t)reinterpret_cast<uintptr_t>(o), o_hi = (uint32_t)(reinterpret_cast<uintptr_t>(o) >> 32);
    const uint64_t livemask = __ballot(act);
    if (mode == kRowsFlat) {
        // live windows are a prefix of the wave's groups; their bytes are contiguous in LDS and in global memory
        const int nlive = __popcll(livemask & (G == 4 ? 0x1111111111111111ull : (G == 8 ? 0x0101010101010101ull : 0x0001000100010001ull)));
        const int total = nlive * N;
        float* ow = reinterpret_cast<float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)o_hi) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)o_lo));
        typedef float vfloat4 __attribute__((ext_vector_type(4)));
        const uint32_t* s4 = reinterpret_cast<const uint32_t*>(stage);
        for (int k = lane; 4 * k + 3 < total; k += 64) {
            const uint32_t b = s4[k];
            vfloat4 v;
            v.x = (float)(b & 0xFFu);
            v.y = (float)((b >> 8) & 0xFFu);
            v.z = (float)((b >> 16) & 0xFFu);
            v.w = (float)(b >> 24);
            __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(ow) + k);
        }
        const int rem = total & 3;                                            // a last, partial wave only
        if (lane < rem) ow[(total & ~3) + lane] = (float)stage[(total & ~3) + lane];
        return;
    }
#pragma unroll
    for (int w = 0; w < WPW; ++w) {
        if (!((livemask >> (w * G)) & 1ull)) continue;                        // (lane 0 of a group always renders row 0 of a live window)
        float* ow = reinterpret_cast<float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)o_hi, w * G) << 32) |
                                             (uint32_t)__builtin_amdgcn_readlane((int)o_lo, w * G));
        const uint8_t* sw = stage + w * N;
        if (mode == kRowsPair) {
            for (int k = lane; 2 * k < N; k += 64) {
                const uint32_t b2 = (uint32_t)sw[2 * k] | ((uint32_t)sw[2 * k + 1] << 8);
                vf2u v2 = {(float)(b2 & 0xFFu), (float)(b2 >> 8)};
                *reinterpret_cast<vf2u*>(ow + 2 * k) = v2;
            }
        } else {
            for (int k = lane; k < N; k += 64) ow[k] = (float)sw[k];
        }
    }
}

// MovingAgent.act of agent `a` of env `env` (agent.py:215-225, gridworld.py:95-122), computed by every lane that calls it
// (same-address loads; vector instructions cost the same for 1 or 64 lanes); `writer` lanes store.  Returns the status
// bits; (old_y, old_x) -> default type and (new_y, new_x) -> my_type are the changed cells of the agent layer if it moved
// (old_y = -1 otherwise).  The stores are issued behind an s_waitcnt vmcnt(0): every load this wave issued before has
// returned, so a window row loaded earlier shows the grid BEFORE this move.
struct MoveOut {
    int old_y, old_x, new_y, new_x;
    uint32_t my_type;
};
__device__ __forceinline__ int move_one(const Params& p, const DevTables* gtab, uint8_t* g, const int64_t env, const int a,
                                        const double* wval, const bool writer, MoveOut& mo, const RowPtrs* io = nullptr) {
    const int H = p.H, W = p.W, HW = H * W;
    int st = 0;
    uint32_t yx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + a];
    uint32_t act;
    int64_t act_raw;
    if (io && io->agent_action) {                        // the policy's own output tensor: one action per env
        act_raw = io->action_kind == SGW_ACT_I64 ? reinterpret_cast<const int64_t*>(io->agent_action)[env]
                : io->action_kind == SGW_ACT_I32 ? (int64_t)reinterpret_cast<const int32_t*>(io->agent_action)[env]
                                                    : (int64_t)reinterpret_cast<const uint8_t*>(io->agent_action)[env];
        act = (act_raw < 0 || act_raw > 255) ? 255u : (uint32_t)act_raw;     // out of range either way: SGW_STATUS_BAD_ACTION
    } else {
        act = p.actions[env * p.A + a];
        act_raw = act;
    }
    const uint32_t my_type = p.agent_state ? p.agent_state[env * p.A + a] : gtab->agent_type[a];
    const double tot = p.total[env];
    if ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W) { yx = 0; st |= SGW_STATUS_BAD_POS; }
    const int my = (int)(yx & 0xFFu), mx = (int)(yx >> 8);
    const bool act_ok = act < (uint32_t)p.nact;
    const int dy = act_ok ? (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1 : 0;
    const int dx = act_ok ? (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1 : 0;
    const int ty = my + dy, tx = mx + dx;
    const bool tinb = act_ok && (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
    const uint32_t t = tinb ? (uint32_t)g[p.zA * HW + ty * W + tx] : 0xFFu;
    const bool tok = tinb && t < (uint32_t)p.T;
    const double val = tok ? (wval ? wval[t & 31u] : gtab->value[t & 31u]) : 0.0;   // reward read BEFORE the move
    const bool pass = tok && ((p.pass_mask >> (t & 31u)) & 1u);
    st |= !act_ok ? SGW_STATUS_BAD_ACTION : (!tinb ? SGW_STATUS_OOB_MOVE : (!tok ? SGW_STATUS_BAD_TYPE : 0));
    mo.old_y = -1; mo.old_x = 0; mo.new_y = -1; mo.new_x = 0; mo.my_type = my_type;
    if (pass) { mo.old_y = my; mo.old_x = mx; mo.new_y = ty; mo.new_x = tx; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (writer) {
        if (pass) {
            g[p.zA * HW + ty * W + tx] = (uint8_t)my_type;
            g[p.zA * HW + my * W + mx] = (uint8_t)p.default_type;
            reinterpret_cast<uint16_t*>(p.pos)[env * p.A + a] = (uint16_t)((uint32_t)ty | ((uint32_t)tx << 8));
        }
        p.rewards[env * p.A + a] = (float)val;
        p.total[env] = tot + val;                                              // float64, agent order (agent.py:172)
        if (io) {
            if (io->agent_action) p.actions[env * p.A + a] = (uint8_t)act;     // the record of what was taken
            if (io->reward_row) io->reward_row[env] = (float)val;
            if (io->action_row) io->action_row[env] = act_raw;
        }
    }
    return st;
}

template <int L, int NW, int R>
__global__ __launch_bounds__(kBlock, 8) void phase_rows(const Params p) {
    constexpr int V = 2 * R + 1, VV = V * V;
    constexpr int G = V <= 4 ? 4 : (V <= 8 ? 8 : 16);   // lanes per env; lane gl < V owns window row gl
    constexpr int EPW = 64 / G;                          // envs per wave
    constexpr int NCH = (V + 7) / 8;                     // 8-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kWaveLds = RowsTab<NW>::kBytes + SGW_MAX_TYPES * 8 + ((EPW * p.C * VV + 15) & ~15);   // = what sgw_create reserves
    const int64_t env0 = ((int64_t)blockIdx.x * 4 + sub) * EPW;
    if (env0 >= p.E) return;                             // whole wave
    const int gl = lane & (G - 1);
    int64_t env = env0 + (lane / G);
    const bool live = env < p.E;
    if (!live) env = p.E - 1;                            // addresses stay valid; nothing is stored
    const DevTables* gtab = p.tab;
    const int H = p.H, W = p.W, HW = H * W, C = p.C, cells = p.cells;
    uint8_t* g = p.grid + env * p.env_stride;
    const bool mover = p.do_move && p.a0 < p.a1;
    const int ra = p.obs_next ? p.a1 : p.a0;
    const bool render = p.obs_next ? p.a1 < p.A : (!(p.flags & SGW_STEP_NO_OBS) && p.a0 < p.a1);
    const bool after = p.obs_next != 0;

    // wave-private tables: [NW][34] counter words, value f64 x 32
    uint32_t* wd = reinterpret_cast<uint32_t*>(smem + sub * kWaveLds);
    double* wval = reinterpret_cast<double*>(smem + sub * kWaveLds + RowsTab<NW>::kBytes);
    RowsTab<NW>::fill(wd, gtab, p, lane);
    if (lane < SGW_MAX_TYPES) wval[lane] = gtab->value[lane];

    // ---- this lane's window row (issued before the move decision: the loads overlap)
    int st = 0;
    uint32_t pyx = 0;
    if (render) pyx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + ra];   // ra is not the mover when `after` (ra = a1 > a0)
    if (render && ((pyx & 0xFFu) >= (uint32_t)H || (pyx >> 8) >= (uint32_t)W)) { pyx = 0; st |= SGW_STATUS_BAD_POS; }
    const int y = (int)(pyx & 0xFFu), x = (int)(pyx >> 8);
    const int gy = y - R + gl;                           // the map row this lane renders
    const bool rowinb = render && gl < V && (unsigned)gy < (unsigned)H;
    uint64_t row[L][NCH];
    rows_load<L, R>(row, g, rowinb, gy, x, W, HW, cells);

    // ---- the move: decided from reads only; its stores wait for every row load of this wave
    gsync<1>();                                                                // table words visible to every lane
    MoveOut mo{-1, 0, -1, 0, 0u};
    if (mover) st |= move_one(p, gtab, g, env, p.a0, wval, live && gl == 0, mo);
    if (st && live && gl == 0) atomicOr(p.status, st);
    if (!render) return;

    // ---- the window of agent `ra` (visual_field.py:9-101)
    if (after && mo.old_y >= 0) {                                              // the move, applied to the loaded rows
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int z = 0; z < L; ++z)
                if (z == p.zA) {
                    uint64_t v = row[z][ch];
                    if (mo.old_y == gy) {
                        const int j = mo.old_x - (x - R) - 8 * ch;
                        if ((unsigned)j < 8u) v = (v & ~(0xFFull << (8 * j))) | ((uint64_t)(p.default_type & 31u) << (8 * j));
                    }
                    if (mo.new_y == gy) {
                        const int j = mo.new_x - (x - R) - 8 * ch;
                        if ((unsigned)j < 8u) v = (v & ~(0xFFull << (8 * j))) | ((uint64_t)(mo.my_type & 31u) << (8 * j));
                    }
                    row[z][ch] = v;
                }
    }
    rows_index<L, R>(row, rowinb, x, W);
    float* o = p.obs + ((env * p.obs_A + (ra - p.obs_a0)) * (int64_t)C) * VV;
    rows_emit<L, NW, R>(row, wd, smem + sub * kWaveLds + RowsTab<NW>::kBytes + SGW_MAX_TYPES * 8, o, C, lane, live && gl < V, p.rows_mode);
}

// The windows of agents [a0, a1) of every env, each into its own destination (rp.p[a] + env * rp.stride): what every
// agent's pov() reads at the start of a policy-driven turn (sorrel/agents/agent.py:167), rendered once; sgw_act then
// repairs the cells that earlier agents' moves change.  Consecutive groups of a wave render consecutive agents of one
// env (their rows share cache lines).
template <int L, int NW, int R>
__global__ __launch_bounds__(kBlock, 8) void observe_rows(const Params p, const RowPtrs rp) {
    constexpr int V = 2 * R + 1, VV = V * V;
    constexpr int G = V <= 4 ? 4 : (V <= 8 ? 8 : 16);
    constexpr int WPW = 64 / G;                          // windows per wave
    constexpr int NCH = (V + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kWaveLds = RowsTab<NW>::kBytes + SGW_MAX_TYPES * 8 + ((WPW * p.C * VV + 15) & ~15);   // the same layout as phase_rows
    const int nA = p.a1 - p.a0;
    const int gl = lane & (G - 1);
    int64_t env;
    int a;
    bool live;
    const int64_t v = (int64_t)blockIdx.x * 4 + sub;                           // wave index
    if (p.rows_by_agent) {
        // per-agent destinations: a wave renders WPW consecutive envs of ONE agent (contiguous in that agent's destination)
        const int64_t nchunk = (p.E + WPW - 1) / WPW;
        if (v >= nchunk * nA) return;
        a = p.a0 + (int)(v / nchunk);
        env = (v % nchunk) * WPW + (lane / G);
        live = env < p.E;
        if (!live) env = p.E - 1;
    } else {
        // consecutive windows = consecutive agents of an env, env after env (contiguous in the [E][A][C][V][V] tensor)
        const int64_t nwin = p.E * nA;
        const int64_t w0 = v * WPW;
        if (w0 >= nwin) return;
        int64_t wi = w0 + (lane / G);
        live = wi < nwin;
        if (!live) wi = nwin - 1;
        env = wi / nA;
        a = p.a0 + (int)(wi - env * nA);
    }
    const int H = p.H, W = p.W, HW = H * W;
    const uint8_t* g = p.grid + env * p.env_stride;
    uint32_t* wd = reinterpret_cast<uint32_t*>(smem + sub * kWaveLds);
    RowsTab<NW>::fill(wd, p.tab, p, lane);
    uint32_t pyx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + a];
    if ((pyx & 0xFFu) >= (uint32_t)H || (pyx >> 8) >= (uint32_t)W) {
        pyx = 0;
        if (live && gl == 0) atomicOr(p.status, SGW_STATUS_BAD_POS);
    }
    const int y = (int)(pyx & 0xFFu), x = (int)(pyx >> 8);
    const int gy = y - R + gl;
    const bool rowinb = gl < V && (unsigned)gy < (unsigned)H;
    uint64_t row[L][NCH];
    rows_load<L, R>(row, g, rowinb, gy, x, W, HW, p.cells);
    gsync<1>();
    rows_index<L, R>(row, rowinb, x, W);
    float* o = reinterpret_cast<float*>(rp.p[a]) + env * rp.stride;
    rows_emit<L, NW, R>(row, wd, smem + sub * kWaveLds + RowsTab<NW>::kBytes + SGW_MAX_TYPES * 8, o, p.C, lane, live && gl < V, p.rows_mode);
}

// ---------------------------------------------------------------- sgw_act: one agent moves, later agents' windows are repaired
// MovingAgent.act of agent a (= p.a0) of every env, and -- instead of rendering the next agent's window again -- the at
// most two cells the move changed are rewritten in the window of every LATER agent that sees them (the windows of a turn
// were rendered once, after the sweep: sgw_observe_rows / sgw_observe).  Agent j's window then shows the grid after the
// moves of agents < j: exactly what its pov() reads in the reference (agent.py:155-173).  Any appearance
// table, float32 or uint8 windows.  G lanes per env (a power of two >= A), lane j = agent j.
template <int G>
__global__ __launch_bounds__(kBlock, 8) void act_patch(const Params p, const RowPtrs rp) {
    constexpr int EPW = 64 / G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t env0 = ((int64_t)blockIdx.x * 4 + sub) * EPW;
    if (env0 >= p.E) return;
    const int j = lane & (G - 1);
    int64_t env = env0 + (lane / G);
    const bool live = env < p.E;
    if (!live) env = p.E - 1;
    const DevTables* gtab = p.tab;
    const int H = p.H, W = p.W, HW = H * W, C = p.C, V = p.V, VV = p.VV, r = p.r, L = p.L, a = p.a0;
    uint8_t* g = p.grid + env * p.env_stride;
    uint32_t pj = 0;
    const bool later = live && j > a && j < p.A;
    if (later) pj = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + j];
    MoveOut mo;
    const int st = move_one(p, gtab, g, env, a, nullptr, live && j == 0, mo, &rp);
    if (st && live && j == 0) atomicOr(p.status, st);
    if (!later || mo.old_y < 0 || rp.p[j] == nullptr) return;
    const int yj = (int)(pj & 0xFFu), xj = (int)(pj >> 8);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int cy = k ? mo.new_y : mo.old_y, cx = k ? mo.new_x : mo.old_x;
        const uint32_t nt = k ? mo.my_type : p.default_type;
        const int di = cy - yj + r, dj = cx - xj + r;
        if ((unsigned)di >= (unsigned)V || (unsigned)dj >= (unsigned)V) continue;
        const int64_t o = env * rp.stride + di * V + dj;
        if (p.onehot) {                                   // one-hot tables: byte counters
            uint32_t cnt[4] = {0u, 0u, 0u, 0u};
            for (int z = 0; z < L; ++z) {
                const uint32_t tz = (z == p.zA ? nt : (uint32_t)g[z * HW + cy * W + cx]) & 31u;
#pragma unroll
                for (int q = 0; q < 4; ++q) cnt[q] += gtab->delta[q][tz];
            }
            for (int c = 0; c < C; ++c) {
                const uint32_t v = (cnt[c >> 2] >> (8 * (c & 3))) & 0xFFu;
                if (p.obs_u8) reinterpret_cast<uint8_t*>(rp.p[j])[o + (int64_t)c * VV] = (uint8_t)v;
                else reinterpret_cast<float*>(rp.p[j])[o + (int64_t)c * VV] = (float)v;
            }
        } else {                                          // np.sum over layers: left to right, float64 (visual_field.py:51)
            for (int c = 0; c < C; ++c) {
                double acc = 0.0;
                for (int z = 0; z < L; ++z) {
                    const uint32_t tz = (z == p.zA ? nt : (uint32_t)g[z * HW + cy * W + cx]) & 31u;
                    acc = z == 0 ? gtab->appearance[tz][c] : acc + gtab->appearance[tz][c];
                }
                reinterpret_cast<float*>(rp.p[j])[o + (int64_t)c * VV] = obs_finish(acc, p.obs_post);
            }
        }
    }
}

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

// (struct RowPtrs -- the per-agent window destinations of sgw_observe_rows / sgw_act -- lives in common.h: step_fast_rows takes it too)

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
//   kRowsRun      (round 4) any other destination: per window, the elements up to the first 16-byte boundary one by one, then
//                 float4 streaming stores on 16-byte boundaries (the staged bytes re-aligned with v_alignbyte), then the rest one by
//                 one -- rows with a tail behind the window (Cleanup: 1 089 + 12 elements per env) have no common alignment;
//   kRowsPair     every destination 8-byte aligned and C*V*V even: per window, float2 runs (kept as a test / A-B path);
//   kRowsSingle   per window, single floats (likewise).
//   idx     this lane's row (rows_index)          o       the window's destination (the same in every lane of a group)
//   wd      the wave's counter words              stage   the wave's staging area, WPW * C*V*V bytes (+ 3)
//   act     this lane renders a row of a live window
constexpr int kRowsSingle = 1, kRowsPair = 2, kRowsRun = 3, kRowsFlat = 4;
template <int L, int NW, int R>
__device__ __forceinline__ void rows_emit(const uint64_t (&idx)[L][(2 * R + 1 + 7) / 8], const uint32_t* wd, uint8_t* stage, float* o,
                                          const int C, const int lane, const bool act, const int mode, const bool staged = false) {   // staged: the bytes are in the staging area already (a second destination)
    constexpr int V = 2 * R + 1, VV = V * V, kTabW = RowsTab<NW>::kTabW;
    constexpr int G = V <= 4 ? 4 : (V <= 8 ? 8 : 16), WPW = 64 / G;
    const int N = C * VV;
    const uint8_t* wdb = reinterpret_cast<const uint8_t*>(wd);
    const int gl = lane & (G - 1);
    uint8_t* mine = stage + (lane / G) * N + gl * V;
    if (act && !staged) {
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
        // lane 0 of every store on a 128-byte line (a run starts on a 16-byte boundary, config 3's on a 64-byte one: a streaming
        // store that covers part of a line costs 10-20 % of the write rate, tools/micro/region_writer.hip)
        const int mis = (int)((reinterpret_cast<uintptr_t>(ow) >> 4) & 7u);
        for (int k = lane - mis; 4 * k + 3 < total; k += 64) {
            if (k < 0) continue;
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
        if (mode == kRowsRun) {
            typedef float vfloat4 __attribute__((ext_vector_type(4)));
            const int h = min((int)(((16u - ((uint32_t)reinterpret_cast<uintptr_t>(ow) & 15u)) & 15u) >> 2), N);   // elements before the first 16-byte boundary
            if (lane < h) ow[lane] = (float)sw[lane];
            const int nb = (N - h) >> 2;
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(stage);
            const int base = w * N + h;                                       // staging byte of the body's first element
            for (int k = lane; k < nb; k += 64) {
                const int off = base + 4 * k;
                const uint32_t lo = s4[off >> 2], hi = (off & 3) ? s4[(off >> 2) + 1] : 0u;
                const uint32_t b = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(off & 3));
                vfloat4 v;
                v.x = (float)(b & 0xFFu);
                v.y = (float)((b >> 8) & 0xFFu);
                v.z = (float)((b >> 16) & 0xFFu);
                v.w = (float)(b >> 24);
                __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(ow + h) + k);
            }
            const int done = h + 4 * nb;
            if (lane < N - done) ow[done + lane] = (float)sw[done + lane];
        } else if (mode == kRowsPair) {
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
    uint32_t found;      // what the target cell held before the agent entered it
    uint32_t left;       // what the agent's own cell held (its type, unless the caller's grid and agent types disagree)
    uint64_t col_old, col_new;   // ActIO.want_cols: every layer's byte of the two cells (byte z = layer z), loaded with the rest
};
struct ActIO {                   // sgw_act's optional extras (see RowPtrs), passed by value
    const void* agent_action = nullptr;
    int action_kind = 0;
    float* reward_row = nullptr;
    int64_t* action_row = nullptr;
    const TurnState* ts = nullptr;
    bool want_cols = false;      // sgw_act: the whole columns of the mover's two cells (window repairs need the other layers)
    // sgw_act (round 6): the act's inputs, loaded by the caller BEFORE its table staging and barrier (addresses that need only the launch arguments)
    bool pre = false;
    uint32_t yx_pre = 0, type_pre = 0;
    int64_t act_pre = 0;
    double tot_pre = 0.0;
};
// The caller's action of agent a in env: its tensor's element, or for SGW_ACT_QF32 the first index of the maximum of the env's row of
// action values (np.argmax; NaN = maximum, as np / torch have it) -- and, under the turn protocol, with probability epsilon[a] the
// engine's own uniform draw for (env, turn, agent) instead, the action SGW_STEP_RANDOM_ACTIONS takes (sorrel/models/pytorch/iqn.py:294-309: the two branches of take_action_from_policy).
// (shared with sgw_choose_actions: the speculative turn's batched form of the same choice)
__device__ __forceinline__ int argmax_explore(const float* q, const int nact, const uint64_t thr, const uint32_t env_id, const uint32_t turn,
                                              const uint32_t ep4, const int a, const uint32_t seed_lo, const uint32_t seed_hi) {
    float best = q[0];
    int arg = 0;
    for (int i = 1; i < nact; ++i) {
        const float v = q[i];
        if (best == best && (v > best || v != v)) { best = v; arg = i; }
    }
    if (thr) {
        const U4 u = philox4x32_10((uint32_t)a >> 2, turn, env_id, ep4 | SGW_STREAM_EXPLORE, seed_lo, seed_hi);
        if ((uint64_t)word_of(u, a & 3) < thr) {
            const U4 w = philox4x32_10((uint32_t)a >> 2, turn, env_id, ep4 | SGW_STREAM_ACTION, seed_lo, seed_hi);
            arg = (int)(((uint64_t)word_of(w, a & 3) * (uint32_t)nact) >> 32);
        }
    }
    return arg;
}
__device__ __forceinline__ int64_t read_action(const Params& p, const void* agent_action, const int kind, const TurnState* ts,
                                               const int64_t env, const int a) {
    if (kind == SGW_ACT_I64) return reinterpret_cast<const int64_t*>(agent_action)[env];
    if (kind == SGW_ACT_I32) return (int64_t)reinterpret_cast<const int32_t*>(agent_action)[env];
    if (kind == SGW_ACT_U8) return (int64_t)reinterpret_cast<const uint8_t*>(agent_action)[env];
    const float* q = reinterpret_cast<const float*>(agent_action) + env * p.nact;
    const uint64_t thr = ts ? ts->eps_thr[a] : 0ull;
    return (int64_t)argmax_explore(q, p.nact, thr, p.first_env + (uint32_t)env, ts ? ts->turn + 1u : 0u, ts ? ts->epoch << 4 : 0u, a, p.seed_lo, p.seed_hi);
}
__device__ __forceinline__ int move_one(const Params& p, const DevTables* gtab, uint8_t* g, const int64_t env, const int a,
                                        const double* wval, const bool writer, MoveOut& mo, const ActIO io = ActIO{}) {
    const int H = p.H, W = p.W, HW = H * W;
    int st = 0;
    uint32_t yx = io.pre ? io.yx_pre : (uint32_t)reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + a];
    uint32_t act;
    int64_t act_raw;
    if (io.pre) {
        act_raw = io.act_pre;
        act = (act_raw < 0 || act_raw > 255) ? 255u : (uint32_t)act_raw;
    } else if (io.agent_action) {                        // the policy's own output tensor: one action per env
        act_raw = read_action(p, io.agent_action, io.action_kind, io.ts, env, a);
        act = (act_raw < 0 || act_raw > 255) ? 255u : (uint32_t)act_raw;     // out of range either way: SGW_STATUS_BAD_ACTION
    } else {
        act = p.actions[env * p.A + a];
        act_raw = act;
    }
    const uint32_t my_type = io.pre ? io.type_pre : (p.agent_state ? p.agent_state[env * p.A + a] : gtab->agent_type[a]);
    const double tot = io.pre ? io.tot_pre : p.total[env];
    if ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W) { yx = 0; st |= SGW_STATUS_BAD_POS; }
    const int my = (int)(yx & 0xFFu), mx = (int)(yx >> 8);
    const bool act_ok = act < (uint32_t)p.nact;
    const int dy = act_ok ? (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1 : 0;
    const int dx = act_ok ? (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1 : 0;
    const int ty = my + dy, tx = mx + dx;
    const bool tinb = act_ok && (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
    const uint32_t t = tinb ? (uint32_t)g[p.zA * HW + ty * W + tx] : 0xFFu;
    const uint32_t here = g[p.zA * HW + my * W + mx];                               // (issued with the target's byte)
    mo.col_old = mo.col_new = 0ull;
    if (io.want_cols) {                                                             // (... and so are the other layers of both cells)
#pragma unroll
        for (int z = 0; z < SGW_MAX_LAYERS; ++z)
            if (z < p.L) {
                mo.col_old |= (uint64_t)g[z * HW + my * W + mx] << (8 * z);
                if (tinb) mo.col_new |= (uint64_t)g[z * HW + ty * W + tx] << (8 * z);
            }
    }
    const bool tok = tinb && t < (uint32_t)p.T;
    const double val = tok ? (wval ? wval[t & 31u] : gtab->value[t & 31u]) : 0.0;   // reward read BEFORE the move
    const bool pass = tok && ((p.pass_mask >> (t & 31u)) & 1u);
    st |= !act_ok ? SGW_STATUS_BAD_ACTION : (!tinb ? SGW_STATUS_OOB_MOVE : (!tok ? SGW_STATUS_BAD_TYPE : 0));
    mo.old_y = -1; mo.old_x = 0; mo.new_y = -1; mo.new_x = 0; mo.my_type = my_type; mo.found = t; mo.left = here;
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
        if (io.agent_action) p.actions[env * p.A + a] = (uint8_t)act;         // the record of what was taken
        if (io.reward_row) io.reward_row[env] = (float)val;
        if (io.action_row) io.action_row[env] = act_raw;
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
    MoveOut mo{-1, 0, -1, 0, 0u, 0u, 0u};
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
    if (rp.dual && rp.ts) {      // the same windows into the agents' replay rows of the turn in flight (by the device's row count): a recorded turn
        const TurnState* ts = rp.ts;   // needs no copy of the windows afterwards
        const bool keep = ts->cap[a] > 0 && ts->states[a] != nullptr;
        float* o2 = keep ? static_cast<float*>(ts->states[a]) + (ts->row[a] * p.E + env) * ts->row_elems[a] : o;
        if (keep && live && gl == 0 && ts->dones[a]) ts->dones[a][ts->row[a] * p.E + env] = 0.f;   // done is False inside an epoch
        // (flat mode: a wave carries consecutive envs of ONE agent, so `keep` is wave-uniform; otherwise dead windows are masked below)
        if (__ballot(keep) != 0ull)
            rows_emit<L, NW, R>(row, wd, smem + sub * kWaveLds + RowsTab<NW>::kBytes + SGW_MAX_TYPES * 8, o2, p.C, lane, live && gl < V && keep, rp.rows_mode2, true);
    }
    if (p.tail_kind != SGW_TAIL_NONE && live) {      // what pov() appends behind the flattened window
        float* t = o + p.C * VV;
        float* t2 = nullptr;                         // ... and behind its second copy in the replay row (a recorded turn), where the row has room
        if (rp.dual && rp.ts && rp.ts->cap[a] > 0 && rp.ts->states[a] && rp.ts->row_elems[a] >= p.C * VV + p.tail_len)
            t2 = static_cast<float*>(rp.ts->states[a]) + (rp.ts->row[a] * p.E + env) * rp.ts->row_elems[a] + p.C * VV;
        if (p.tail_kind == SGW_TAIL_AGENT_IS_IT) {   // TagAgent.pov: [self.it]
            if (gl == 0) {
                const float it = (p.agent_state && p.agent_state[env * p.A + a] == p.tag_it) ? 1.f : 0.f;
                t[0] = it;
                if (t2) t2[0] = it;
            }
        } else {                                     // CleanupObservation.observe: the positional code of the agent's cell
            const float* src = p.tail_table + ((int64_t)y * W + x) * p.tail_len;
            for (int k = gl; k < p.tail_len; k += G) {
                const float f = src[k];
                t[k] = f;
                if (t2) t2[k] = f;
            }
        }
    }
}

// ---------------------------------------------------------------- sgw_act: one agent acts, later agents' windows are repaired
// Agent.act of agent a (= p.a0) of every env -- MovingAgent.act (agent.py:215-225), TagAgent.act (examples/tag/agents.py:
// 76-106) or CleanupAgent.act (examples/cleanup/agents.py:93-177), RULE -- and, instead of rendering the next agent's window
// again, every cell the act changed is rewritten in the window of every LATER agent that sees it (the windows of a turn
// were rendered once, after the sweep: sgw_observe_rows / SGW_STEP_NO_MOVE).  Agent j's window then shows the grid after
// the acts of agents < j: exactly what its pov() reads in the reference (agent.py:155-173).  An act changes
// few cells: the mover's two; for Tag the tagger's and its victim's; for Cleanup up to 3 R beam cells on the layer above.
// Any appearance table, float32 or uint8 windows.  G lanes per env, lane j = agents j, j + G, ... (NJ of them: G * NJ >= A;
// 9..16 agents share 8 lanes two by two -- Cleanup's ten agents: half the waves of a 16-lane group, an act without repairs
// 23 -> ... us at 65 536 envs); every lane of a group evaluates the act from same-address loads (vector instructions
// cost the same for 1 or 64 lanes), lane 0 of the group stores.  A changed cell is recomputed from the grid column with the changed layer's NEW type substituted (the
// stores of this launch are never read back by it).
template <int G, int NJ, int RULE, bool ONEHOT>
__global__ __launch_bounds__(kBlock, RULE == SGW_AGENT_RULE_CLEANUP ? 4 : 8) void act_patch(const Params p, const RowPtrs rp) {   // (Cleanup: registers spilled at 64 and at 96)
    constexpr int EPW = 64 / G;
    // the appearance tables a repair looks up: in LDS (a repair is then ONE global round trip -- the cell's other layers --
    // instead of a chain of dependent table loads from global memory: Cleanup's crowded 11x11 windows made an act 35-90 us)
    __shared__ uint8_t s_chan[SGW_MAX_TYPES];             // one-hot tables: the channel a type lights (0xFF: none -- EmptyEntity's all-zero row)
    __shared__ double s_app[ONEHOT ? 1 : SGW_MAX_TYPES * SGW_MAX_CHANNELS];
    __shared__ double s_value[SGW_MAX_TYPES];            // Entity.value: keeps a dependent global load out of every act
    const int tid = threadIdx.x;
    const DevTables* gtab = p.tab;
    const int lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t env0 = ((int64_t)blockIdx.x * 4 + sub) * EPW;
    const bool wave_live = env0 < p.E;
    const int j = lane & (G - 1);
    int64_t env = env0 + (lane / G);
    const bool live = wave_live && env < p.E;
    if (!live) env = p.E - 1;
    const int H = p.H, W = p.W, HW = H * W, C = p.C, V = p.V, VV = p.VV, r = p.r, L = p.L, a = p.a0;
    // Everything below up to the barrier is loads whose addresses need nothing but the launch arguments: they fly while the tables
    // travel to LDS (a recorded turn at <= 1 024 envs is a chain of such launches, each as long as its dependent round trips).
    uint32_t pjv[NJ];                                                        // where this lane's agents stand (0xFFFFFFFF: no such agent)
    float* ring[NJ];                                                         // a recorded turn: this lane's agents' replay rows of the turn in flight
#pragma unroll
    for (int n = 0; n < NJ; ++n) {
        const int jj = j + n * G;
        pjv[n] = (live && jj < p.A) ? (uint32_t)reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + jj] : 0xFFFFFFFFu;
        ring[n] = nullptr;
        if (rp.dual && rp.ts && live && jj > a && jj < p.A && rp.ts->cap[jj] > 0 && rp.ts->states[jj])
            ring[n] = static_cast<float*>(rp.ts->states[jj]) + (rp.ts->row[jj] * p.E + env) * rp.ts->row_elems[jj];
    }
    float* reward_row = rp.reward_row;
    int64_t* action_row = rp.action_row;
    if (rp.ts && rp.ts->cap[a] > 0) {        // the replay rows of the turn in flight, by the engine's own count
        if (rp.ts->rewards[a]) reward_row = rp.ts->rewards[a] + rp.ts->row[a] * p.E;
        if (rp.ts->actions[a]) action_row = rp.ts->actions[a] + rp.ts->row[a] * p.E;
    }
    // ... and the act's own inputs (the acting agent's cell, type, action, the env's total): a policy's fresh output comes from the memory side,
    // ~2 us away -- behind the barrier below it headed the act's chain of dependent loads (profiles/r06_act_after_probe.txt)
    const uint32_t yx_pre = (uint32_t)reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + a];
    const uint32_t type_pre = p.agent_state ? (uint32_t)p.agent_state[env * p.A + a] : (uint32_t)gtab->agent_type[a];
    const double tot_pre = p.total[env];
    int64_t act_pre;
    if (rp.agent_action) act_pre = read_action(p, rp.agent_action, rp.action_kind, rp.ets, env, a);
    else act_pre = (int64_t)p.actions[env * p.A + a];
    if (tid >= 128 && tid < 128 + SGW_MAX_TYPES) s_value[tid - 128] = gtab->value[tid - 128];
    if constexpr (ONEHOT) {
        if (tid < SGW_MAX_TYPES) {
            uint32_t ch = 0xFFu;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t w = gtab->delta[q][tid];
                if (w) ch = 4u * q + ((uint32_t)__builtin_ctz(w) >> 3);
            }
            s_chan[tid] = (uint8_t)ch;
        }
    } else {
        for (int i = tid; i < SGW_MAX_TYPES * SGW_MAX_CHANNELS; i += kBlock) s_app[i] = reinterpret_cast<const double*>(gtab->appearance)[i];
    }
    __syncthreads();
    if (!wave_live) return;
    uint8_t* g = p.grid + env * p.env_stride;
    const bool writer = live && j == 0;

    // Cell (cy, cx) held type `ot` on layer `zc` and now holds `nt`: it is rewritten in the window of every later agent that
    // contains it -- only the channels whose value changes (every store here is a lone 4-byte write into a tensor far bigger than
    // the caches: they are what a repair costs).  `col`: the cell's bytes on every layer (byte z = layer z) as they were before
    // this act, where the caller has them in registers already; kNoCol: the other layers are read here.
    // One-hot tables (a type lights one channel or none): WHAT changes is a property of the cell, not of the window -- at most
    // two (channel, new value) pairs, `cell_update` packs them into one dword (channel 0xFF: nothing) -- so it is computed once
    // per cell (for Cleanup's beams by the lane that placed the beam, in parallel) and `apply` is a window test + <= 2 stores.
    // The per-window form -- s_delta counters for every channel group, 81 (cell, agent) pairs deep in a serial loop -- made a
    // Cleanup act with every agent firing 28 us at 1 024 envs.
    constexpr uint64_t kNoCol = ~0ull;
    constexpr uint32_t kNoUpdate = 0x00FF00FFu;
    auto load_col = [&](const int cy, const int cx) {
        uint64_t col = 0ull;
#pragma unroll
        for (int z = 0; z < SGW_MAX_LAYERS; ++z)
            if (z < L) col |= (uint64_t)g[z * HW + cy * W + cx] << (8 * z);
        return col;
    };
    auto cell_update = [&](const int zc, const uint32_t ot, const uint32_t nt, const uint64_t col) -> uint32_t {
        const uint32_t co = s_chan[ot & 31u], cn = s_chan[nt & 31u];
        if (ot == nt || co == cn) return kNoUpdate;
        uint32_t vo = 0u, vn = 1u;
#pragma unroll
        for (int z = 0; z < SGW_MAX_LAYERS; ++z)
            if (z < L && z != zc) {
                const uint32_t c = s_chan[(uint32_t)(col >> (8 * z)) & 31u];
                vo += c == co ? 1u : 0u;
                vn += c == cn ? 1u : 0u;
            }
        return co | (vo << 8) | (cn << 16) | (vn << 24);
    };
    auto apply = [&](const int cy, const int cx, const uint32_t u) {
      if (u == kNoUpdate) return;
#pragma unroll
      for (int n = 0; n < NJ; ++n) {
        const int jj = j + n * G;
        if (!(live && jj > a && jj < p.A) || rp.p[jj] == nullptr) continue;  // a later agent with a window to keep current
        const int di = cy - (int)(pjv[n] & 0xFFu) + r, dj = cx - (int)((pjv[n] >> 8) & 0xFFu) + r;
        if ((unsigned)di >= (unsigned)V || (unsigned)dj >= (unsigned)V) continue;
        const int64_t o = env * rp.stride + di * V + dj;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t c = (u >> (16 * h)) & 0xFFu, v = (u >> (16 * h + 8)) & 0xFFu;
            if (c == 0xFFu) continue;
            if (p.obs_u8) reinterpret_cast<uint8_t*>(rp.p[jj])[o + (int64_t)c * VV] = (uint8_t)v;
            else reinterpret_cast<float*>(rp.p[jj])[o + (int64_t)c * VV] = (float)v;
            if (ring[n]) ring[n][di * V + dj + (int64_t)c * VV] = (float)v;       // ... and in the agent's replay row of this turn
        }
      }
    };
    // any appearance table: np.sum over layers, left to right, float64 (visual_field.py:51), per window
    auto patch_general = [&](const int cy, const int cx, const int zc, const uint32_t ot, const uint32_t nt, const uint64_t col) {
      if (ot == nt) return;
#pragma unroll
      for (int n = 0; n < NJ; ++n) {
        const int jj = j + n * G;
        if (!(live && jj > a && jj < p.A) || rp.p[jj] == nullptr) continue;
        const int di = cy - (int)(pjv[n] & 0xFFu) + r, dj = cx - (int)((pjv[n] >> 8) & 0xFFu) + r;
        if ((unsigned)di >= (unsigned)V || (unsigned)dj >= (unsigned)V) continue;
        const int64_t o = env * rp.stride + di * V + dj;
        uint32_t tz[SGW_MAX_LAYERS];
#pragma unroll
        for (int z = 0; z < SGW_MAX_LAYERS; ++z)                             // the column's other layers: loads issued together
            tz[z] = z < L ? ((z == zc ? nt : (col != kNoCol ? (uint32_t)(col >> (8 * z)) : (uint32_t)g[z * HW + cy * W + cx])) & 31u) : 0u;
        for (int c = 0; c < C; ++c) {
            if (s_app[(ot & 31u) * SGW_MAX_CHANNELS + c] == s_app[(nt & 31u) * SGW_MAX_CHANNELS + c]) continue;
            double acc = s_app[tz[0] * SGW_MAX_CHANNELS + c];
#pragma unroll
            for (int z = 1; z < SGW_MAX_LAYERS; ++z)
                if (z < L) acc += s_app[tz[z] * SGW_MAX_CHANNELS + c];
            reinterpret_cast<float*>(rp.p[jj])[o + (int64_t)c * VV] = obs_finish(acc, p.obs_post);
            if (ring[n]) ring[n][di * V + dj + (int64_t)c * VV] = obs_finish(acc, p.obs_post);
        }
      }
    };
    auto patch = [&](const int cy, const int cx, const int zc, const uint32_t ot, const uint32_t nt, const uint64_t col) {
        if constexpr (ONEHOT) {
            if (ot == nt) return;
            apply(cy, cx, cell_update(zc, ot, nt, col != kNoCol ? col : load_col(cy, cx)));
        } else {
            patch_general(cy, cx, zc, ot, nt, col);
        }
    };

    if constexpr (RULE == SGW_AGENT_RULE_MOVE) {
        MoveOut mo;
        ActIO io{rp.agent_action, rp.action_kind, reward_row, action_row, rp.ets, true};
        io.pre = true; io.yx_pre = yx_pre; io.type_pre = type_pre; io.act_pre = act_pre; io.tot_pre = tot_pre;
        const int st = move_one(p, gtab, g, env, a, s_value, writer, mo, io);
        if (st && writer) atomicOr(p.status, st);
        if (mo.old_y < 0) return;
        patch(mo.old_y, mo.old_x, p.zA, mo.left, p.default_type, mo.col_old);
        patch(mo.new_y, mo.new_x, p.zA, mo.found, mo.my_type, mo.col_new);
        return;
    } else {
        // ---- inputs of the act (same-address loads in every lane of the group)
        int st = 0;
        uint32_t yx = yx_pre;                                             // (loaded ahead of the barrier, above)
        const int64_t act_raw = act_pre;
        const uint32_t act = rp.agent_action ? ((act_raw < 0 || act_raw > 255) ? 255u : (uint32_t)act_raw) : (uint32_t)act_raw;
        const uint32_t my_type = type_pre;
        const double tot = tot_pre;
        if ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W) { yx = 0; st |= SGW_STATUS_BAD_POS; }
        const int y = (int)(yx & 0xFFu), x = (int)(yx >> 8);
        const bool act_ok = act < (uint32_t)p.nact;
        uint64_t col_own = 0ull;             // every layer of the agent's own cell (its type on the agent layer, unless the caller's grid and agent types disagree)
#pragma unroll
        for (int zl = 0; zl < SGW_MAX_LAYERS; ++zl)
            if (zl < L) col_own |= (uint64_t)g[zl * HW + y * W + x] << (8 * zl);
        double reward = 0.0, total_add = 0.0;
        bool pass = false;
        int ny = y, nx = x;
        uint32_t found = 0u;                 // what the cell the agent moved onto held
        uint64_t col_new = kNoCol;           // ... and that cell's column
        if constexpr (RULE == SGW_AGENT_RULE_CLEANUP) {
            // CleanupAgent.act: a move action turns the agent (even if the move fails) and moves it; clean / zap place a
            // beam on the layer above; the reward is the value summed over ALL layers of the target, read before the move
            uint32_t dir = p.agent_dir[env * p.A + a] & 3u;
            const uint32_t kind = act_ok ? (p.kind_pack >> (2 * (act & 15u))) & 3u : 0u;
            if (!act_ok) {
                st |= SGW_STATUS_BAD_ACTION;
            } else {
                if (kind == SGW_ACTION_MOVE) {
                    const int dy = (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1, dx = (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1;
                    ny = y + dy; nx = x + dx;
                    dir = (dy == -1 && dx == 0) ? 0u : (dy == 1 && dx == 0) ? 2u : (dy == 0 && dx == -1) ? 3u : (dy == 0 && dx == 1) ? 1u : dir;
                }
                // the target column's bytes are loaded BEFORE the beams are placed (one round trip for both: beams never land on
                // the target of the same act -- a firing agent's target is its own cell, the beam cells start next to it)
                const bool tin = (unsigned)ny < (unsigned)H && (unsigned)nx < (unsigned)W;
                uint32_t tl[SGW_MAX_LAYERS];
#pragma unroll
                for (int zl = 0; zl < SGW_MAX_LAYERS; ++zl) tl[zl] = (tin && zl < L) ? (uint32_t)g[zl * HW + ny * W + nx] : 0u;
                if (kind != SGW_ACTION_MOVE && p.zA + 1 < L) {
                    // The beam cells (1..R ahead; 0..R-1 ahead of the right / left neighbours) are spread over the lanes of the
                    // group: lane k tests and writes cells k, k + G, ... (independent loads, one wait) and reads the cell's other
                    // layers in the same round trip; a ballot tells every lane which cells took a beam, and each later agent's lane
                    // repairs those inside its window from the placing lane's registers (no load in the repair loop).
                    const int fy = dir == 0 ? -1 : dir == 2 ? 1 : 0, fx = dir == 1 ? 1 : dir == 3 ? -1 : 0;
                    const int ry = dir == 1 ? 1 : dir == 3 ? -1 : 0, rx = dir == 0 ? 1 : dir == 2 ? -1 : 0;
                    const uint32_t beam = kind == SGW_ACTION_CLEAN ? p.clean_beam : p.zap_beam;
                    const int nb = 3 * p.beam_radius;
                    auto beam_cell = [&](const int b, int& by, int& bx) {
                        const int arm = b / p.beam_radius, i = b - arm * p.beam_radius;
                        const int step = arm == 0 ? i + 1 : i, side = arm == 0 ? 0 : (arm == 1 ? 1 : -1);
                        by = y + side * ry + step * fy;
                        bx = x + side * rx + step * fx;
                        return (unsigned)by < (unsigned)H && (unsigned)bx < (unsigned)W;
                    };
                    for (int b0 = 0; b0 < nb; b0 += G) {                    // (uniform over the wave: nb is a launch constant)
                        int by, bx;
                        const int b = b0 + j;
                        bool placed = false;
                        uint32_t was = 0u;                                      // what the beam cell held
                        uint32_t col_lo = 0u, col_hi = 0u;                      // the cell's column (SGW_MAX_LAYERS = 7 bytes: two dwords for the shuffles)
                        uint32_t upd = kNoUpdate, cyx = 0u;                     // one-hot: what changes in the cell, and where it is
                        if (b < nb && beam_cell(b, by, bx)) {
                            const int boff = (p.zA + 1) * HW + by * W + bx;
                            const uint64_t col = load_col(by, bx);
                            was = (uint32_t)(col >> (8 * (p.zA + 1))) & 31u;
                            col_lo = (uint32_t)col; col_hi = (uint32_t)(col >> 32);
                            placed = !((p.beam_block_mask >> was) & 1u);
                            if (placed && live) g[boff] = (uint8_t)beam;        // (a cell is visited at most once per act)
                            if constexpr (ONEHOT) {
                                if (placed) upd = cell_update(p.zA + 1, was, beam, col);
                                cyx = (uint32_t)by | ((uint32_t)bx << 8);
                            }
                        }
                        const unsigned long long all = __ballot(ONEHOT ? (placed && upd != kNoUpdate) : placed);
                        uint32_t mine_grp = (uint32_t)(all >> (lane & ~(G - 1))) & (G == 64 ? 0xFFFFFFFFu : ((1u << (G & 31)) - 1u));
                        if constexpr (G == 64) {
                            unsigned long long m = all;
                            while (m) {
                                const int k = __builtin_ctzll(m);
                                m &= m - 1ull;
                                if constexpr (ONEHOT) {
                                    const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)cyx, k);
                                    apply((int)(at & 0xFFu), (int)(at >> 8), (uint32_t)__builtin_amdgcn_readlane((int)upd, k));
                                } else {
                                    int cy, cx;
                                    beam_cell(b0 + k, cy, cx);
                                    const uint64_t ck = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)col_lo, k) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)col_hi, k) << 32);
                                    patch_general(cy, cx, p.zA + 1, (uint32_t)__builtin_amdgcn_readlane((int)was, k), beam, ck);
                                }
                            }
                        } else {
                            while (mine_grp) {                                  // (divergent across the envs of a wave: each lane walks its own group's bits)
                                const int k = __ffs(mine_grp) - 1;
                                mine_grp &= mine_grp - 1u;
                                const int src = (lane & ~(G - 1)) + k;
                                if constexpr (ONEHOT) {
                                    const uint32_t at = (uint32_t)__shfl((int)cyx, src);
                                    apply((int)(at & 0xFFu), (int)(at >> 8), (uint32_t)__shfl((int)upd, src));
                                } else {
                                    int cy, cx;
                                    beam_cell(b0 + k, cy, cx);
                                    const uint64_t ck = (uint64_t)(uint32_t)__shfl((int)col_lo, src) | ((uint64_t)(uint32_t)__shfl((int)col_hi, src) << 32);
                                    patch_general(cy, cx, p.zA + 1, (uint32_t)__shfl((int)was, src), beam, ck);
                                }
                            }
                        }
                    }
                }
                if (!tin) {
                    st |= SGW_STATUS_OOB_MOVE;
                    ny = y; nx = x;
                } else {
                    col_new = 0ull;
#pragma unroll
                    for (int zl = 0; zl < SGW_MAX_LAYERS; ++zl)
                        if (zl < L) {
                            reward += s_value[tl[zl] & 31u];                  // every layer of the target, read BEFORE the move
                            col_new |= (uint64_t)tl[zl] << (8 * zl);
                        }
                    total_add = reward * (double)(p.total_factor - 1);       // the extra add inside act() (agents.py:172)
                    uint32_t t = 0xFFu;
#pragma unroll
                    for (int zl = 0; zl < SGW_MAX_LAYERS; ++zl)
                        if (zl == p.zA) t = tl[zl];
                    pass = t < (uint32_t)p.T && ((p.pass_mask >> (t & 31u)) & 1u);
                    found = t;
                }
            }
            if (writer) p.agent_dir[env * p.A + a] = (uint8_t)dir;
        } else {
            // TagAgent.act: the move of MovingAgent.act without its reward ...
            if (!act_ok) {
                st |= SGW_STATUS_BAD_ACTION;
            } else {
                const int ty = y + (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1, tx = x + (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1;
                if ((unsigned)ty >= (unsigned)H || (unsigned)tx >= (unsigned)W) {
                    st |= SGW_STATUS_OOB_MOVE;
                } else {
                    const uint32_t t = g[p.zA * HW + ty * W + tx];
                    if (t >= (uint32_t)p.T) st |= SGW_STATUS_BAD_TYPE;
                    else if ((p.pass_mask >> (t & 31u)) & 1u) { pass = true; ny = ty; nx = tx; found = t; }
                }
            }
        }
        const uint32_t left = (uint32_t)(col_own >> (8 * p.zA)) & 0xFFu;      // what the agent's own cell holds on the agent layer
        if (!pass) { ny = y; nx = x; col_new = col_own; }
        // ---- Tag: an agent that is "it" hands the flag to the first NotIt neighbour of the cell it now stands on
        uint32_t mine_now = my_type;
        int vy = -1, vx = -1;
        if constexpr (RULE == SGW_AGENT_RULE_TAG) {
            if (my_type == p.tag_it) {
#pragma unroll
                for (int d = 3; d >= 0; --d) {          // Location.adjacent order: up, right, down, left; the first match wins
                    const int ay = ny + (d == 0 ? -1 : d == 2 ? 1 : 0), ax = nx + (d == 1 ? 1 : d == 3 ? -1 : 0);
                    if ((unsigned)ay >= (unsigned)H || (unsigned)ax >= (unsigned)W) continue;
                    // (the one cell this act has already changed is the agent's old cell: default type now, never NotIt)
                    const uint32_t nt = (pass && ay == y && ax == x) ? p.default_type : (uint32_t)g[p.zA * HW + ay * W + ax];
                    if (nt == p.tag_notit) { vy = ay; vx = ax; }
                }
                if (vy >= 0) mine_now = p.tag_notit;
            }
            reward = mine_now != p.tag_it ? p.tag_reward : 0.0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every read of the grid above has returned: now the stores
        if (writer) {
            if (pass) {
                g[p.zA * HW + y * W + x] = (uint8_t)p.default_type;
                reinterpret_cast<uint16_t*>(p.pos)[env * p.A + a] = (uint16_t)((uint32_t)ny | ((uint32_t)nx << 8));
            }
            if (pass || mine_now != my_type) g[p.zA * HW + ny * W + nx] = (uint8_t)mine_now;
            if (vy >= 0) g[p.zA * HW + vy * W + vx] = (uint8_t)p.tag_it;
            if (RULE == SGW_AGENT_RULE_TAG) {
                p.agent_state[env * p.A + a] = (uint8_t)mine_now;
                if (p.state_at_pov) p.state_at_pov[env * p.A + a] = (uint8_t)my_type;      // what TagAgent.pov appended
            }
            p.rewards[env * p.A + a] = (float)reward;
            p.total[env] = (tot + total_add) + reward;               // float64, in the reference's order of additions
            if (rp.agent_action) p.actions[env * p.A + a] = (uint8_t)act;
            if (reward_row) reward_row[env] = (float)reward;
            if (action_row) action_row[env] = act_raw;
            if (st) atomicOr(p.status, st);
        }
        if (RULE == SGW_AGENT_RULE_TAG && vy >= 0 && live) {
#pragma unroll
            for (int n = 0; n < NJ; ++n)
                if (j + n * G != a && pjv[n] == ((uint32_t)vy | ((uint32_t)vx << 8))) {
                    const int jj = j + n * G;
                    p.agent_state[env * p.A + jj] = (uint8_t)p.tag_it;   // the agent standing on the victim's cell
                    // ... and, if it has yet to observe this turn, the "it" flag behind its window (TagAgent.pov reads self.it)
                    if (jj > a && rp.p[jj] != nullptr && p.tail_kind == SGW_TAIL_AGENT_IS_IT && !p.obs_u8) {
                        reinterpret_cast<float*>(rp.p[jj])[env * rp.stride + (int64_t)C * VV] = 1.f;
                        if (ring[n] && rp.ts->row_elems[jj] > (int64_t)C * VV) ring[n][(int64_t)C * VV] = 1.f;
                    }
                }
        }
        // (the own cell's and the target's columns are in registers; the victim's -- Tag worlds with more than one layer -- is read in the repair)
        if (pass) patch(y, x, p.zA, left, p.default_type, col_own);
        if (pass || mine_now != my_type) patch(ny, nx, p.zA, pass ? found : left, mine_now, pass && RULE == SGW_AGENT_RULE_CLEANUP ? col_new : (pass ? kNoCol : col_own));
        if (vy >= 0) patch(vy, vx, p.zA, p.tag_notit, p.tag_it, kNoCol);
    }
}

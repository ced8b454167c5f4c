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

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


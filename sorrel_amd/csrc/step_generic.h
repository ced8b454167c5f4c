// step_generic.h -- part of the single translation unit sgw.hip (included inside its anonymous namespace).
// step_kernel<G, ONEHOT, L, C, RULE, r, H, W>: every shape and rule; 256 / 64 / 32 / 16 lanes per env (small envs share a
// wave; a workgroup per env runs its agent phases wave by wave behind a ticket).
#pragma once

// ---------------------------------------------------------------- step kernel
// Per-env LDS slice: [grid cells_pad][pos 2 x AC][act AC][rew f32 x AC][type AC][pov type AC][facing AC][ticket + total 16], AC = Params::agent_cap:
// 64, or 128 for engines with more than 64 agents (round 6) -- every other engine keeps the 656 bytes it had
__host__ __device__ constexpr int agent_lds_bytes(int cap) { return 10 * cap + 16; }

#ifndef SGW_GENERIC_WAVES
#define SGW_GENERIC_WAVES 6
#endif
// G = threads per environment: 256 (a workgroup per env, worlds above 4 KiB), 64 (a wave per env) or, for small worlds,
// 32 / 16 lanes of a wave -- two or four envs share a wave and its instruction stream.  The kernel keeps every piece of
// per-env state in the group's LDS slice and uses no cross-lane instruction, so a sub-wave group needs nothing but the
// wave-level ordering of DS instructions; what it buys is that the per-env instruction count, which bounds small worlds
// (a 21x21x2 world keeps 29 of 64 lanes busy in the sweep and 25 in the window gather), is shared by 2 or 4 envs.
// TR: compile-time vision radius (0 = run-time) for the example defaults: the window size becomes a constant, so the
// channel planes of an observation are immediate store offsets and the cell -> (i, j) split needs no division.
// TH, TW: compile-time world size on top of that (the examples' own maps).
// MULTI: sgw_rollout's instance (the turn loop; nturns > 1).  The single-turn instances are compiled without the loop: its
// loop-carried state cost the small-world kernels a fifth of their speed when it was added to the one kernel (round 2: 16x16 / 4
// agents, 32 lanes per env, 54 -> 67 us; found in round 3 by bisecting tools/group_sweep.py).
// AC: entries of the per-agent LDS arrays -- 64, or 128 for engines with more than 64 agents (round 6; workgroup-per-env instances only: a group of G < 256
// threads holds at most G agents).  A compile-time constant: as a launch parameter it cost the small-world instances 2-4 VGPRs, three of them a wave of occupancy.
// ROWS (round 6): the instance behind sgw_sweep_observe_rows on the worlds this kernel serves (small worlds packed two or four to a wave, rule worlds above 8 KiB,
// more than 64 agents) -- agent a's window of env e goes to rp.p[a] + e * rp.stride, the bound row tail behind it; specialised in-process only.  Every instance
// takes the row pointers as its second argument (read by ROWS instances only).
template <int G, bool ONEHOT, int TL = 0, int TC = 0, int RULE = SGW_AGENT_RULE_MOVE, int TR = 0, int TH = 0, int TW = 0, bool MULTI = false, int AC = 64, bool ROWS = false>
__global__ __launch_bounds__(kBlock, SGW_GENERIC_WAVES) void step_kernel(const Params p, [[maybe_unused]] const RowPtrs rp) {
    static_assert(AC == 64 || (AC == SGW_MAX_AGENTS && G == 256), "more than 64 agents: the workgroup-per-env instances");
    static_assert(!ROWS || !MULTI, "ROWS: a single-turn instance");
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
    uint8_t* s_pos = slice + p.cells_pad;             // [A][2]
    uint8_t* s_act = s_pos + 2 * AC;                  // [A]
    float* s_rew = reinterpret_cast<float*>(s_act + AC);
    uint8_t* s_type = s_act + 5 * AC;                   // [A] current entity type of each agent
    uint8_t* s_pov = s_type + AC;                       // [A] its type when it observed
    uint8_t* s_dir = s_pov + AC;                        // [A] its facing
    // G = 256 (worlds above 4 KiB): the load, the sweep and the write-back use the whole workgroup, but an AGENT PHASE is the
    // work of one wave, and the four waves take the agents in turn behind a ticket in LDS: wave (a - a0) mod 4 waits
    // until agent a - 1 has acted, captures agent a's window bytes in registers, acts (Tag / Cleanup / move: the same code
    // as below, its synchronisation now wave-level), passes the ticket on and only then turns the captured bytes into
    // stores -- so the sequential chain is capture + act, and the observation work of four agents overlaps.  Before,
    // all 256 threads rendered one agent (121 cells: half of them idle) and met at ~5 workgroup barriers per agent.
    constexpr bool kTicket = G == 256;
    constexpr int GA = kTicket ? kWave : G;            // threads that cooperate on one agent
    constexpr int WPA = kTicket ? 1 : WPE;             // waves that synchronise inside an agent phase
    const int atid = kTicket ? (tid & 63) : gtid;      // index inside that group
    volatile uint32_t* s_ticket = reinterpret_cast<volatile uint32_t*>(s_dir + AC);     // G = 256: whose turn it is (u32), the env's running total (f64 at + 8)
    double* s_tot = reinterpret_cast<double*>(s_dir + AC + 8);

    const int r = TR ? TR : p.r, V = TR ? 2 * TR + 1 : p.V, VV = TR ? (2 * TR + 1) * (2 * TR + 1) : p.VV;
    const int H = TH ? TH : p.H, W = TW ? TW : p.W;
    // window cell(s) this thread renders: fixed for the whole kernel
    int wi[kMaxPass], wj[kMaxPass];
#pragma unroll
    for (int k = 0; k < kMaxPass; ++k) {
        const int w = atid + k * GA;
        wi[k] = w / V;
        wj[k] = w - wi[k] * V;
    }
    const bool write_obs = !(p.flags & SGW_STEP_NO_OBS);
    const bool dirty = (p.flags & SGW_STEP_SWEEP) || (p.do_move && p.a1 > p.a0);
    const int zoff = p.zA * H * W;
    const int HW = H * W;
    uint32_t turn0 = p.turn, ep4 = p.epoch << 4;     // kernel arguments, or (sgw_turn_*) the engine's device-side count
    if (p.ts) { turn0 = p.ts->turn + 1u; ep4 = p.ts->epoch << 4; }   // (ts->turn: turns completed)

    // one env per group and launch (no persistent loop: nothing stays live from one env to the next, and the
    // dispatcher balances the workgroups)
    const int64_t env = (int64_t)blockIdx.x * EPB + sub;
    if (env < p.E) {
        const uint32_t env_id = p.first_env + (uint32_t)env;
        uint8_t* ggrid = p.grid + env * p.env_stride;
        load_grid<G>(p, ggrid, lg, gtid);
        double tot = 0.0;
        if (gtid == 0 && p.do_move) {
            tot = p.total[env];
            if constexpr (kTicket) *s_tot = tot;
        }
        uint32_t yx0 = 0;                     // this thread's agent: position at the start of the call
        if (gtid < p.A) {
            uint16_t yx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + gtid];
            if ((yx & 0xFF) >= H || (yx >> 8) >= W) {   // garbage in: stay inside this env's LDS slice, and say so
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
        const uint32_t nturns = MULTI ? p.nturns : 1u;
        for (uint32_t tix = 0; tix < nturns; ++tix) {
        const uint32_t turn = turn0 + tix;
        if (gtid < p.A && p.do_move && gtid >= p.a0 && gtid < p.a1) {
            uint8_t* acts = p.actions + tix * p.ts_act;
            uint32_t act;
            if (p.flags & SGW_STEP_RANDOM_ACTIONS) {
                const U4 w = philox4x32_10((uint32_t)gtid >> 2, turn, env_id,
                                           ep4 | SGW_STREAM_ACTION, p.seed_lo, p.seed_hi);
                act = (uint32_t)(((uint64_t)word_of(w, gtid & 3) * (uint32_t)p.nact) >> 32);
                acts[env * p.A + gtid] = (uint8_t)act;
            } else {
                act = acts[env * p.A + gtid];
            }
            s_act[gtid] = (uint8_t)act;
        }
        if constexpr (kTicket)
            if (gtid == 0) s_ticket[0] = s_ticket[1] = (uint32_t)p.a0;
        gsync<WPE>();
        if (p.flags & SGW_STEP_SWEEP) {
            if (p.has_become) {
                sweep_ordered<WPE, G>(p, tab, lg, env_id, gtid, turn, ep4, p.L, HW);
            } else {
                if (p.single_spawner) sweep_single<G>(p, lg, env_id, gtid, turn, ep4);
                else sweep<G>(p, tab, lg, env_id, gtid, turn, ep4);
                gsync<WPE>();
            }
        }

        // ROWS: what pov() appends behind the flattened window (phase.h, observe_rows: the same two kinds) -- by the `n` threads that render the window
        [[maybe_unused]] auto row_tail = [&](float* row, const int a, const int y, const int x, const int me, const int n) {
            if (p.tail_kind == SGW_TAIL_NONE) return;
            float* t = row + (TC ? TC : p.C) * VV;
            if (p.tail_kind == SGW_TAIL_AGENT_IS_IT) {
                if (me == 0) t[0] = s_type[a] == p.tag_it ? 1.f : 0.f;
            } else {
                const float* src = p.tail_table + ((int64_t)y * W + x) * p.tail_len;
                for (int k = me; k < p.tail_len; k += n) t[k] = src[k];
            }
        };
        const int a_end = (p.obs_next && p.a1 < p.A) ? p.a1 + 1 : p.a1;   // OBS_NEXT: one extra, observe-only iteration
        if constexpr (!kTicket) {
        // ---- a wave (or part of one) per env: the agents one after the other, the whole group on each
            for (int a = p.a0; a < a_end; ++a) {
                const int y = s_pos[2 * a], x = s_pos[2 * a + 1];
                // ---- pov: egocentric window (visual_field.py:9-101)
                if (p.obs_next ? a == p.a1 : write_obs) {
                    float* obase;
                    if constexpr (ROWS) {
                        obase = static_cast<float*>(rp.p[a]) + env * rp.stride;
                        row_tail(obase, a, y, x, gtid, G);
                    } else {
                        obase = p.obs + tix * p.ts_obs + ((env * p.obs_A + (a - p.obs_a0)) * (int64_t)p.C) * VV;
                    }
                    auto render = [&](const int w, const int i, const int j) {
                        const int gy = y - r + i, gx = x - r + j;
                        const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                        const int off = gy * W + gx;
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
                                        if (p.obs_u8) reinterpret_cast<uint8_t*>(p.obs)[(o - p.obs) + c * VV] = (uint8_t)v;
                                        else o[c * VV] = (float)v;
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
                                o[c * VV] = obs_finish(acc, p.obs_post);
                            }
                        }
                    };
    #pragma unroll
                    for (int k = 0; k < kMaxPass; ++k) {
                        const int w = gtid + k * G;
                        if (w < VV) render(w, wi[k], wj[k]);
                    }
                    {   // further passes (small groups, wide windows): (i, j) advance by G cells, no division
                        int i = wi[kMaxPass - 1], j = wj[kMaxPass - 1];
                        for (int w = gtid + kMaxPass * G; w < VV; w += G) {
                            j += G;
                            while (j >= V) { j -= V; ++i; }
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
                        if ((unsigned)by < (unsigned)H && (unsigned)bx < (unsigned)W) {
                            const int boff = (p.zA + 1) * HW + by * W + bx;
                            if (!((p.beam_block_mask >> (lg[boff] & 31u)) & 1u))
                                lg[boff] = (uint8_t)(kind == SGW_ACTION_CLEAN ? p.clean_beam : p.zap_beam);
                        }
                    }
                    gsync<WPE>();
                    const bool inb = act_ok && (unsigned)ny < (unsigned)H && (unsigned)nx < (unsigned)W;
                    double val = 0.0;
                    uint32_t t = 0xFFu;
                    if (inb) {
                        for (int zl = 0; zl < p.L; ++zl) val += tab->value[lg[zl * HW + ny * W + nx] & 31u];   // all layers, BEFORE the move
                        t = lg[zoff + ny * W + nx];
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
                            lg[zoff + ny * W + nx] = (uint8_t)my_type;
                            lg[zoff + y * W + x] = (uint8_t)p.default_type;
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
                const bool inb = act_ok && (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
                const int taddr = zoff + ty * W + tx;
                const int oaddr = zoff + y * W + x;
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
                    const int own = zoff + cy * W + cx;
                    if (my_type == p.tag_it)         // only the agent that is "it" looks around (one in A: the other agents skip four LDS round trips)
    #pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int ay = cy + (d == 0 ? -1 : d == 2 ? 1 : 0);
                        const int ax = cx + (d == 1 ? 1 : d == 3 ? -1 : 0);
                        const bool ain = (unsigned)ay < (unsigned)H && (unsigned)ax < (unsigned)W;
                        const uint32_t nt = ain ? lg[zoff + ay * W + ax] : 0xFFu;
                        if (mine_now == p.tag_it && nt == p.tag_notit) {
                            mine_now = p.tag_notit;
                            if (gtid == 0) {
                                lg[own] = (uint8_t)p.tag_notit;
                                lg[zoff + ay * W + ax] = (uint8_t)p.tag_it;
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
        } else {
        // ---- G = 256: an agent phase is the work of ONE wave; the waves take the agents in turn (see kTicket above)
            auto agent_phase = [&](const int a) {
                // what does not depend on the other agents is read BEFORE the wait for the ticket: an agent's position only
                // changes by its own act, its action is fixed for the turn
                const uint32_t yx_a = reinterpret_cast<const uint16_t*>(s_pos)[a];
                const int y = (int)(yx_a & 0xFFu), x = (int)(yx_a >> 8);
                const uint32_t act_a = (p.do_move && a < p.a1) ? (uint32_t)s_act[a] : 0u;
                double pend[2];                              // G = 256: this agent's additions to total_reward, applied in agent order below
                int npend = 0;
                auto add_total = [&](const double v) {       // world.total_reward += reward, float64, agent order (agent.py:172)
                    pend[npend++] = v;
                };
                const bool observe = p.obs_next ? a == p.a1 : write_obs;
                float* obase0_;
                if constexpr (ROWS) obase0_ = static_cast<float*>(rp.p[a]) + env * rp.stride;
                else obase0_ = p.obs + tix * p.ts_obs + ((env * p.obs_A + (a - p.obs_a0)) * (int64_t)p.C) * VV;
                float* const obase0 = obase0_;
                if constexpr (ROWS) if (observe) row_tail(obase0, a, y, x, atid, GA);   // (nobody acts in a ROWS launch: the agent's type and cell are the turn's)
                uint32_t clo[kMaxPass], chi[kMaxPass];      // G = 256: the captured window bytes (layers 0-3 | 4-6, five bits each)
                bool cin[kMaxPass];
                int coff[kMaxPass];
                // the geometry of the move (agent.py:187-225; Cleanup: only a move action moves) and of the window: pure
                // arithmetic on this agent's own position and action, so it is done before the wait as well
                const bool act_ok = act_a < (uint32_t)p.nact;
                uint32_t kind = SGW_ACTION_MOVE;
                if constexpr (RULE == SGW_AGENT_RULE_CLEANUP) kind = act_ok ? (p.kind_pack >> (2 * act_a)) & 3u : 0u;
                const bool moves = act_ok && kind == SGW_ACTION_MOVE;
                const int dy = moves ? (int)((p.dy_pack >> (2 * act_a)) & 3u) - 1 : 0;
                const int dx = moves ? (int)((p.dx_pack >> (2 * act_a)) & 3u) - 1 : 0;
                const int ty = y + dy, tx = x + dx;
                const bool inb = act_ok && (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
                const int taddr = zoff + ty * W + tx;
                const int oaddr = zoff + y * W + x;
                if (observe) {
#pragma unroll
                    for (int k = 0; k < kMaxPass; ++k) {
                        const int w = atid + k * GA;
                        const int gy = y - r + wi[k], gx = x - r + wj[k];
                        cin[k] = w < VV && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                        coff[k] = cin[k] ? gy * W + gx : 0;
                    }
                }
                if (p.do_move) {                             // (sgw_observe: nothing changes the grid, no order to keep)
                    while (s_ticket[0] != (uint32_t)a) {}    // (s_sleep 1 / 2 / 4 between looks: 125.3 / 126.2 / 127.4 us against 125.5)
                    asm volatile("" ::: "memory");
                }
                // G = 256: pass the ticket on, then turn the captured bytes into stores (visual_field.py:41-55, 89-94)
                auto after_act = [&]() {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS writes have landed (LDS only: no wait for stores)
                    if (atid == 0) s_ticket[0] = (uint32_t)(a + 1);
                    // the float64 total: same agent order, its own ticket, so that the read-modify-write in LDS is not part
                    // of the act chain (the two chains run side by side)
                    if (atid == 0 && p.do_move && a < p.a1) {
                        while (s_ticket[1] != (uint32_t)a) {}
                        asm volatile("" ::: "memory");
                        double t = *s_tot;
                        for (int i = 0; i < npend; ++i) t += pend[i];
                        *s_tot = t;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        s_ticket[1] = (uint32_t)(a + 1);
                    }
                    if (!observe) return;
#pragma unroll
                    for (int k = 0; k < kMaxPass; ++k) {
                        const int w = atid + k * GA;
                        if (w >= VV) continue;
                        float* o = obase0 + w;
                        const int Cn = TC ? TC : p.C, Ln = TL ? TL : p.L;
                        if constexpr (ONEHOT) {
                            constexpr int NWq = TC ? (TC + 3) / 4 : 4;
                            uint32_t cnt[NWq];
#pragma unroll
                            for (int q = 0; q < NWq; ++q) cnt[q] = 0u;
                            const int nw = (Cn + 3) >> 2;
                            if (cin[k]) {
                                for (int z = 0; z < Ln; ++z) {
                                    const uint32_t t = z < 4 ? (clo[k] >> (8 * z)) & 31u : (chi[k] >> (8 * (z - 4))) & 31u;
#pragma unroll
                                    for (int q = 0; q < NWq; ++q)
                                        if (q < nw) cnt[q] += tab->delta[q][t];
                                }
                            } else {
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
                                        if (p.obs_u8) reinterpret_cast<uint8_t*>(p.obs)[(o - p.obs) + c * VV] = (uint8_t)v;
                                        else o[c * VV] = (float)v;
                                    }
                                }
                            }
                        } else {
                            for (int c = 0; c < Cn; ++c) {
                                double acc;
                                if (cin[k]) {   // np.sum over layers: left to right, float64 (visual_field.py:51)
                                    acc = tab->appearance[clo[k] & 31u][c];
                                    for (int z = 1; z < Ln; ++z)
                                        acc += tab->appearance[z < 4 ? (clo[k] >> (8 * z)) & 31u : (chi[k] >> (8 * (z - 4))) & 31u][c];
                                } else {
                                    acc = tab->appearance[p.fill_type][c];
                                }
                                o[c * VV] = obs_finish(acc, p.obs_post);
                            }
                        }
                    }
                };
                // ---- pov: egocentric window (visual_field.py:9-101)
                if (observe) {
                    float* obase = obase0;
                    auto render = [&](const int w, const int i, const int j) {
                        const int gy = y - r + i, gx = x - r + j;
                        const bool inb = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                        const int off = gy * W + gx;
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
                                        if (p.obs_u8) reinterpret_cast<uint8_t*>(p.obs)[(o - p.obs) + c * VV] = (uint8_t)v;
                                        else o[c * VV] = (float)v;
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
                                o[c * VV] = obs_finish(acc, p.obs_post);
                            }
                        }
                    };
                    // the first kMaxPass cells of this lane: bytes now, stores after the act
#pragma unroll
                    for (int k = 0; k < kMaxPass; ++k) {
                        clo[k] = chi[k] = 0;
                        for (int z = 0; z < (TL ? TL : p.L); ++z) {      // (out-of-window lanes read cell 0 and ignore it)
                            const uint32_t t = lg[z * HW + coff[k]] & 31u;
                            if (z < 4) clo[k] |= t << (8 * z);
                            else chi[k] |= t << (8 * (z - 4));
                        }
                    }
                    {   // windows wider than that (more than 128 cells): the rest is rendered here, before the act
                        int i = wi[kMaxPass - 1], j = wj[kMaxPass - 1];
                        for (int w = atid + kMaxPass * GA; w < VV; w += GA) {
                            j += GA;
                            while (j >= V) { j -= V; ++i; }
                            render(w, i, j);
                        }
                    }
                }
                if (!p.do_move || a >= p.a1) { after_act(); return; }
                if constexpr (RULE == SGW_AGENT_RULE_CLEANUP) {
                    // ---- CleanupAgent.act (sorrel/examples/cleanup/agents.py:146-177).  Every thread evaluates the
                    // same LDS bytes, so all control flow here is uniform; single threads do the writes.
                    const uint32_t my_type = s_type[a];
                    const int ny = ty, nx = tx;
                    const uint32_t facing = s_dir[a] & 3u;
                    gsync<WPA>();
                    if (act_ok && kind != SGW_ACTION_MOVE && p.zA + 1 < p.L)
                      for (int bc = atid; bc < 3 * p.beam_radius; bc += GA) {
                        // beam cells on the layer above: 1..R ahead; 0..R-1 ahead of the right / left neighbours
                        const int arm = bc / p.beam_radius, i = bc - arm * p.beam_radius;
                        const int fy = facing == 0 ? -1 : facing == 2 ? 1 : 0, fx = facing == 1 ? 1 : facing == 3 ? -1 : 0;
                        const int ry = facing == 1 ? 1 : facing == 3 ? -1 : 0, rx = facing == 0 ? 1 : facing == 2 ? -1 : 0;
                        const int step = arm == 0 ? i + 1 : i, side = arm == 0 ? 0 : (arm == 1 ? 1 : -1);
                        const int by = y + side * ry + step * fy, bx = x + side * rx + step * fx;
                        if ((unsigned)by < (unsigned)H && (unsigned)bx < (unsigned)W) {
                            const int boff = (p.zA + 1) * HW + by * W + bx;
                            if (!((p.beam_block_mask >> (lg[boff] & 31u)) & 1u))
                                lg[boff] = (uint8_t)(kind == SGW_ACTION_CLEAN ? p.clean_beam : p.zap_beam);
                        }
                    }
                    gsync<WPA>();
                    double val = 0.0;
                    uint32_t t = 0xFFu;
                    if (inb) {
                        for (int zl = 0; zl < p.L; ++zl) val += tab->value[lg[zl * HW + ny * W + nx] & 31u];   // all layers, BEFORE the move
                        t = lg[zoff + ny * W + nx];
                    }
                    const bool pass = inb && t < (uint32_t)p.T && ((p.pass_mask >> (t & 31u)) & 1u);
                    gsync<WPA>();
                    if (atid == 0) {
                        s_pov[a] = (uint8_t)my_type;
                        if (act_ok && kind == SGW_ACTION_MOVE) {            // movement() turns the agent even if the move fails
                            if (dy == -1 && dx == 0) s_dir[a] = 0;
                            else if (dy == 1 && dx == 0) s_dir[a] = 2;
                            else if (dy == 0 && dx == -1) s_dir[a] = 3;
                            else if (dy == 0 && dx == 1) s_dir[a] = 1;
                        }
                        if (pass) {
                            lg[zoff + ny * W + nx] = (uint8_t)my_type;
                            lg[zoff + y * W + x] = (uint8_t)p.default_type;
                            s_pos[2 * a] = (uint8_t)ny;
                            s_pos[2 * a + 1] = (uint8_t)nx;
                        }
                        s_rew[a] = (float)val;
                        add_total(val * (double)(p.total_factor - 1));       // the extra add inside act() (agents.py:172) ...
                        add_total(val);                                  // ... and Agent.transition's own (agent.py:172)
                        st_bits |= (!act_ok ? SGW_STATUS_BAD_ACTION : 0) | ((act_ok && !inb) ? SGW_STATUS_OOB_MOVE : 0);
                    }
                    gsync<WPA>();
                    after_act();
                    return;
                }
                // ---- act: MovingAgent.movement / act, Gridworld.move (agent.py:187-225, gridworld.py:95-122)
                const uint32_t my_type = s_type[a];
                const uint32_t t = inb ? lg[taddr] : 0xFFu;
                const bool tok = t < (uint32_t)p.T;
                double val = (inb && tok && RULE == SGW_AGENT_RULE_MOVE) ? tab->value[t & 31u] : 0.0;   // reward read BEFORE the move
                const bool pass = inb && tok && ((p.pass_mask >> (t & 31u)) & 1u);
                const int cy = pass ? ty : y, cx = pass ? tx : x;   // where the agent stands after the move
                gsync<WPA>();   // every thread has read s_type / the target before thread 0 rewrites them
                if (atid == 0) {
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
                    gsync<WPA>();
                    uint32_t mine_now = my_type;
                    const int own = zoff + cy * W + cx;
                    if (my_type == p.tag_it)         // only the agent that is "it" looks around
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int ay = cy + (d == 0 ? -1 : d == 2 ? 1 : 0);
                        const int ax = cx + (d == 1 ? 1 : d == 3 ? -1 : 0);
                        const bool ain = (unsigned)ay < (unsigned)H && (unsigned)ax < (unsigned)W;
                        const uint32_t nt = ain ? lg[zoff + ay * W + ax] : 0xFFu;
                        if (mine_now == p.tag_it && nt == p.tag_notit) {
                            mine_now = p.tag_notit;
                            if (atid == 0) {
                                lg[own] = (uint8_t)p.tag_notit;
                                lg[zoff + ay * W + ax] = (uint8_t)p.tag_it;
                                s_type[a] = (uint8_t)p.tag_notit;
                            }
                            // the neighbour's slot: the agent standing on (ay, ax) (more than 64 agents: a second round of the wave's lanes)
                            for (int b = atid; b < p.A; b += GA)
                                if (b != a && s_pos[2 * b] == ay && s_pos[2 * b + 1] == ax) s_type[b] = (uint8_t)p.tag_it;
                        }
                    }
                    val = mine_now != p.tag_it ? p.tag_reward : 0.0;
                }
                if (atid == 0) {
                    s_rew[a] = (float)val;
                    add_total(val);
                }
                gsync<WPA>();
                after_act();
            };
            for (int a = p.a0 + (tid >> 6); a < a_end; a += kBlock / kWave) agent_phase(a);
            __syncthreads();            // every agent has acted: rewards / positions / the grid are final
        }
        if (p.do_move && gtid >= p.a0 && gtid < p.a1) {      // this turn's rewards (and what TagAgent.pov appends)
            p.rewards[tix * p.ts_rew + env * p.A + gtid] = s_rew[gtid];
            if (p.state_at_pov) p.state_at_pov[env * p.A + gtid] = s_pov[gtid];
        }
        }   // turns

        if (dirty) {
            if (RULE == SGW_AGENT_RULE_MOVE && !(p.flags & SGW_STEP_SWEEP) && nturns == 1) {   // (a rollout's earlier turns moved other cells too)
                // a policy-driven phase (no sweep, plain moves): only the movers' two cells changed -- write those bytes,
                // not the whole grid (with agents i < j both touching a cell, both write its FINAL content: no race)
                if (gtid >= p.a0 && gtid < p.a1) {
                    const uint32_t now = reinterpret_cast<const uint16_t*>(s_pos)[gtid];
                    if (now != yx0) {
                        const int o0 = zoff + (int)(yx0 & 0xFFu) * W + (int)(yx0 >> 8), o1 = zoff + (int)(now & 0xFFu) * W + (int)(now >> 8);
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
            if (gtid == 0) p.total[env] = kTicket ? *s_tot : tot;
            if (atid == 0 && st_bits) atomicOr(p.status, st_bits);      // (G = 256: lane 0 of every wave kept its own bits)
        }
    }
}


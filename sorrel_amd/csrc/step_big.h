// step_big.h -- part of the single translation unit sgw.hip (included inside its anonymous namespace).
// step_big<...>: a 512-thread workgroup per env, worlds above 4 KiB (config 5).
#pragma once

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
// Eight waves per workgroup (measured 115 us per config-5 launch against 123 us with four waves per workgroup and 143 us with two: round 2,
// same box, interleaved A/B).  Workgroups per CU: the single-turn one-env-per-workgroup instances are compiled for SGW_BIG_WAVES = 8 waves per
// SIMD (50 VGPRs, no scratch) = FOUR per CU (round 5; rounds 2-4 said four here but compiled for 6-7 waves per SIMD: three); the walking
// and rollout variants (80 VGPRs) hold three.  What the fourth buys (profiles/r05_c5_occupancy_ab.txt): 1 024 envs 47.8 -> 43.1 us; nothing
// from two rounds of workgroups on (2 048 envs 114 against the walking variant's 91; 4 096 / 8 192 staged 187 / 358 at either occupancy).
#ifndef SGW_BIG_THREADS
#define SGW_BIG_THREADS 512
#endif
constexpr int kBigThreads = SGW_BIG_THREADS;
#ifndef SGW_WALK_WAVES
#define SGW_WALK_WAVES 6   // waves per SIMD the WALK variant is compiled for (8 = 64 VGPRs: spills the prefetched units)
#endif
#ifndef SGW_BIG_WAVES
#define SGW_BIG_WAVES 8    // waves per SIMD the single-turn one-env-per-workgroup variant is compiled for
#endif
constexpr int kBigWaves = kBigThreads / 64;
constexpr int kBigAgentLds = 64 * 4 * 3 + 64;   // ta, oa, npos | agent types  (round 3: the f64 rewards of a turn and the value table live in wave 0's
                                                // registers -- with the observation staging, config 5's image then fits a CU four times)
constexpr int kBigTagLds = 64 * 4;                                       // ... and, for Tag, the tag journal behind them
constexpr int big_agent_lds(bool tag) { return kBigAgentLds + (tag ? kBigTagLds : 0); }

// MULTI: sgw_rollout's variant -- a turn loop around sweep / moves / observations with the env's 32 KiB resident in LDS
// (later turns sweep the units read back from LDS; only the last turn is followed by the write-back).
// WALK: the single-turn variant for batches larger than the chip holds at once -- as many workgroups as are resident,
// each walking envs blockIdx.x, + gridDim.x, ...  A wave cannot end before its stores are acknowledged, so with one env
// per workgroup the second round of workgroups (whose first 17 us are the sweep: pure VALU) only starts once the first
// round's 190 MB of observation stores have drained; a workgroup that carries on with its next env sweeps while they
// drain.  vmcnt counts loads and stores together on gfx9, so the next env's grid (64 B per thread), positions, actions
// and total are loaded during phase M of the current env -- before this env's stores are issued -- and waited for
// there; the loop then contains no load that would have to wait behind the observation stores.
// TAG (round 3): TagAgent.act (sorrel/examples/tag/agents.py:76-106) on this kernel.  A Tag agent moves exactly like a plain
// mover (agents are impassable), so phase M resolves the moves as before; what is sequential on top is the "it" token:
// an agent that is "it" WHEN ITS TURN COMES (at the start of the turn, or tagged by an earlier agent of the same turn)
// hands the flag to the first NotIt agent next to the cell it now stands on.  Wave 0 walks only those agents (typically
// one per env and turn): lane b knows where agent b stands at that moment (its new cell if it has already acted, its old
// one if not), four ballots find the neighbours in Location.adjacent order.  The journal of an agent then also says what
// type it carried when it acted and whom it tagged, and phase R undoes tags along with moves, latest first.
// BT: threads per workgroup.  512 (eight waves) for worlds with many agents (config 5: 64); 256 for up to 32 agents, where eight waves
// have one or two windows each and mostly wait at the barriers (round 3, 8 192 envs: 90x90x2 / 16 agents 122 -> 103 us, 100x100x2 / 8
// agents / 11x11 104 -> 88, Tag 128x128 / 32 agents 145 -> 116; config 5 itself 352 -> 394: it keeps 512).
// ROWS (round 6): the instance behind sgw_sweep_observe_rows -- agent a's window of env e goes to rp.p[a] + e * rp.stride (its own row, e.g. of its replay buffer),
// followed by the bound row tail; a separate instantiation (specialised in-process only): in the ordinary instances the extra addressing cost the walking variant
// 37 spilled scalars and 8 bytes of scratch per lane.  Every instance takes the row pointers as its second argument (read by ROWS instances only).
template <bool ONEHOT, int TL, int TC, int TR, bool MULTI = false, bool WALK = false, bool TAG = false, int BT = kBigThreads, bool ROWS = false>
__global__ __launch_bounds__(BT, WALK ? SGW_WALK_WAVES : (MULTI ? 6 : SGW_BIG_WAVES)) void step_big(const Params p, [[maybe_unused]] const RowPtrs rp) {
    static_assert(!ROWS || (!MULTI && !WALK), "ROWS: the plain single-turn variant");
    constexpr int kBT = BT, kBW = BT / 64;   // threads / waves of this instance
    static_assert(!(MULTI && WALK), "a rollout keeps one env per workgroup");
    // Philox key schedule per block (common.h): config 5's share on the walking variant 94 -> 90 us; the plain variant is indifferent at
    // config 5's shape and loses 3.6 % on a 48x48 world, so it keeps the shared keys
    constexpr bool kBigOwnKeys = WALK;
    static_assert(!(TAG && (MULTI || WALK)), "Tag: single-turn, one env per workgroup");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid0 = threadIdx.x;
    // (Round 6 measured the transposed assignment -- workgroup b plays env (b % 8) * (blocks / 8) + b / 8, so that the workgroups of an XCD write one contiguous
    // eighth of the round's windows: + 1-3 % for the staged windows on one card, - 3-6 % on the next, - 1-3 % for the direct stores; not kept: profiles/r06_c5_remap_ab.txt)
    int64_t env = blockIdx.x;
#ifdef SGW_STAMPS
    unsigned long long tprev_ = 0;
#define STAMPB(i)                                                                                            \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long t_;                                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (threadIdx.x == 0 && (i) > 0 && env < kStampEnvs) g_stamps[env * 8 + (i)-1] = t_ - tprev_;         \
        tprev_ = t_;                                                                                         \
    } while (0)
    STAMPB(0);
    if (threadIdx.x == 0 && env < kStampEnvs) {
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
    uint8_t* s_atype = reinterpret_cast<uint8_t*>(s_np + 64);                         // agent_type[64]
    [[maybe_unused]] uint32_t* s_rm = reinterpret_cast<uint32_t*>(s_atype + 64);      // TAG: the tag journal -- victim's cell (y, x) | type the agent carried when it acted << 16 | tagged << 24
    uint8_t* lg = smem + p.tab_bytes + big_agent_lds(TAG);
    uint4* lg16 = reinterpret_cast<uint4*>(lg);
    const DevTables* gtab = p.tab;

    const bool write_obs = !(p.flags & SGW_STEP_NO_OBS);
    const bool do_sweep = (p.flags & SGW_STEP_SWEEP) != 0;
    const bool dirty = do_sweep || (p.do_move && p.a1 > p.a0);
    const bool rnd = (p.flags & SGW_STEP_RANDOM_ACTIONS) != 0;
    uint32_t turn0 = p.turn, ep4 = p.epoch << 4;     // kernel arguments, or (sgw_turn_*) the engine's device-side count
    if (p.ts) { turn0 = p.ts->turn + 1u; ep4 = p.ts->epoch << 4; }   // (ts->turn: turns completed)

    // ---- tables -> LDS
    if constexpr (ONEHOT) {
        uint32_t* wd = reinterpret_cast<uint32_t*>(smem);
        if (tid0 < (p.tab_bytes >> 2)) wd[tid0] = reinterpret_cast<const uint32_t*>(gtab->delta)[tid0];   // (the counter words of the channels in use)
    } else {
        double* wa = reinterpret_cast<double*>(smem);
        for (int i = tid0; i < SGW_MAX_TYPES * SGW_MAX_CHANNELS; i += kBT) wa[i] = reinterpret_cast<const double*>(gtab->appearance)[i];
    }
    const double vtab = gtab->value[tid0 & 31];   // lane t (of wave 0): value[t], f64 (keeps global loads out of the chain)
    if (tid0 >= 64 && tid0 < 128) s_atype[tid0 - 64] = gtab->agent_type[tid0 - 64];
    const uint32_t* wdelta = reinterpret_cast<const uint32_t*>(smem);
    const double(*wapp)[SGW_MAX_CHANNELS] = reinterpret_cast<const double(*)[SGW_MAX_CHANNELS]>(smem);

    // per-agent state of wave 0 (lane a = agent a), carried from turn to turn of a rollout
    uint32_t yx = 0;
    int st_lane = 0;
    uint32_t ta_v = 0xFFFFFFFFu, npos_v = 0, oaddr_v = 0, jr = 0;
    const uint32_t nturns = MULTI ? p.nturns : 1u;
    // WALK: what an env needs from global memory is loaded one env ahead (the first env's: here)
    uint4 nu[4] = {};
    uint32_t nyx = 0, nact = 0;
    double ntot = 0.0;
    auto prefetch = [&](const int64_t e, const int t) {
        if (e >= p.E) return;
        const uint4* nsrc = reinterpret_cast<const uint4*>(p.grid + e * p.env_stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = k * kBT + t;
            if (idx < nunits) nu[k] = nsrc[idx];
        }
        if (t < p.A) {
            nyx = reinterpret_cast<const uint16_t*>(p.pos)[e * p.A + t];
            if (!rnd && p.do_move && t >= p.a0 && t < p.a1) nact = p.actions[e * p.A + t];
        }
        if (t == 0 && p.do_move) ntot = p.total[e];
    };
    if constexpr (WALK) prefetch(env, tid0);
    // WALK: every workgroup plays `walk_static` envs of its own (blockIdx + k * gridDim), then takes envs off a shared counter until
    // they run out.  (Round 4: the eight XCDs of a chip finish the same share of config 5 up to 15 % apart -- stamps per XCD: a
    // workgroup's life 27.5 ... 31.6 us -- and with a purely static split the launch waits for the slowest.)  Exactly
    // (E - static part) + gridDim numbers are drawn per launch (every workgroup draws one that is out of range), so whoever draws the
    // last one resets the counter for the next launch; no other draw is outstanding then.
    [[maybe_unused]] int walk_k = 1;                 // envs of the static share this workgroup has started (wave-uniform: a scalar register)
    [[maybe_unused]] uint32_t next_u = 0;            // the env this workgroup plays next (0xFFFFFFFF: none), wave-uniform
    do {
    // WALK: the thread index is re-derived per env behind an opaque copy, so that nothing computed from it is hoisted
    // out of the env loop and kept in registers across it (the hoisted version needed 80 VGPRs + 27 spilled)
    const int tid = WALK ? (int)opaque((uint32_t)tid0) : tid0;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mine = tid >= p.a0 && tid < p.a1 && tid < p.A;
    const uint32_t env_id = p.first_env + (uint32_t)env;
    if (wv == 0 && tid < p.A) {
        if constexpr (WALK) yx = nyx;
        else yx = reinterpret_cast<const uint16_t*>(p.pos)[env * p.A + tid];
        if ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W) {   // garbage in: stay inside the LDS grid, and say so
            yx = 0;
            st_lane |= SGW_STATUS_BAD_POS;
        }
    }
    double tot = 0.0;
    if (tid == 0 && p.do_move) {
        if constexpr (WALK) tot = ntot;
        else tot = p.total[env];
    }
    for (uint32_t tix = 0; tix < nturns; ++tix) {
    const uint32_t turn = turn0 + tix;
    // ---- grid -> LDS (first turn), sweep on the registers, 4 units per thread per round
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.grid + env * p.env_stride);
        for (int base = 0; base < nunits; base += 4 * kBT) {
            uint4 u[4];
            uint32_t hits[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + k * kBT + tid;
                if constexpr (WALK) u[k] = nu[k];   // (nunits <= 4 * kBT: host)
                else if (idx < nunits) u[k] = (MULTI && tix > 0) ? lg16[lunit(idx)] : src[idx];
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
                const int idx = base + k * kBT + tid;
                hits[k] = 0;
                if (idx < nunits) {
                    if (!(MULTI && tix > 0)) lg16[lunit(idx)] = u[k];
                    if (do_sweep) hits[k] = sweep_hits<kBigOwnKeys>(u[k], (uint32_t)idx, p, env_id, turn, ep4);
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
                        const uint32_t off = (uint32_t)(base + k * kBT + tid) * 16u + cell;
                        const U4 kw = philox4x32_10<kBigOwnKeys>(opaque(off >> 2), turn, env_id, ep4 | SGW_STREAM_SPAWN_KIND,
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
                    const U4 w = philox4x32_10<kBigOwnKeys>(opaque((uint32_t)tid >> 2), turn, env_id, ep4 | SGW_STREAM_ACTION,
                                               p.seed_lo, p.seed_hi);
                    act = __umulhi(word_of(w, tid & 3), (uint32_t)p.nact);
                    p.actions[tix * p.ts_act + env * p.A + tid] = (uint8_t)act;
                } else {
                    if constexpr (WALK) act = nact;
                    else act = p.actions[tix * p.ts_act + env * p.A + tid];
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
    if constexpr (WALK) {
        if (tid == 0) {          // which env this workgroup plays next
            uint32_t nx;
            if (walk_k < p.walk_static) {
                nx = (uint32_t)(env + gridDim.x);
            } else {
                const uint32_t first = (uint32_t)p.walk_static * gridDim.x;      // the envs behind the static shares
                const uint32_t tail = p.E > first ? (uint32_t)p.E - first : 0u;
                const uint32_t got = atomicAdd(p.walk_ctr, 1u);
                if (got == tail + gridDim.x - 1u) *p.walk_ctr = 0u;              // the last draw of the launch
                nx = got < tail ? first + got : 0xFFFFFFFFu;
            }
            *reinterpret_cast<volatile uint32_t*>(smem + p.walk_word) = nx;
        }
    }
    __syncthreads();             // grid (+ sweep patches) and tables visible to every wave
    STAMPB(1);                   // load + sweep done
    if constexpr (WALK) {
        next_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile uint32_t*>(smem + p.walk_word));
        ++walk_k;
        prefetch(next_u == 0xFFFFFFFFu ? p.E : (int64_t)next_u, tid);   // the next env's inputs: issued now, complete by the end of phase M (before any observation store)
    }

    // ---- phase M: the strictly sequential part, by wave 0 alone, entirely in registers.
    // All targets are read from the pre-move grid in ONE LDS round trip (lane a = agent a).  What
    // agent a finds on its target when its turn comes differs from that only if an earlier mover
    // left from or entered that very cell; the scalar loop below finds the latest such mover with
    // two ballots (no LDS access inside the loop).  The grid is patched afterwards in two ordered
    // batches: every mover's old cell <- default, then every mover's new cell <- its type (a cell
    // can be left and then entered in one turn, never the other way round: an agent moves once).
    if (wv == 0 && p.do_move) {
        uint32_t atype_v = s_atype[lane];
        if constexpr (TAG)
            if (lane < p.A) atype_v = p.agent_state[env * p.A + lane];   // the agent's CURRENT type: "it" or not (survives resets)
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
        uint32_t type_v = atype_v;        // TAG: the agent's type as the turn proceeds (ends as its final type)
        uint32_t pov_v = atype_v;         // TAG: its type when its own turn came (what TagAgent.pov appends; what its move carries)
        uint32_t tagj = 0u;               // TAG: journal word of this agent's tag, 0 = tagged nobody
        if constexpr (TAG) {
            const uint32_t cur0 = passed_v ? npos_v : yx;                 // where this agent stands once it has acted
            unsigned long long todo = ~0ull << p.a0;                      // agents whose turn is still to come
            if (p.a1 < 64) todo &= (1ull << p.a1) - 1ull;
            while (true) {
                const unsigned long long its = __ballot(lane < p.A && type_v == p.tag_it) & todo;
                if (!its) break;
                const int i = __builtin_ctzll(its);                       // the next agent that acts as "it"
                todo &= ~0ull << (i + 1);
                const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cur0, i);
                const int cy = (int)(ci & 0xFFu), cx = (int)(ci >> 8);
                const uint32_t here = lane < i ? cur0 : yx;               // where agent `lane` stands at that moment (later agents have not moved yet)
                const bool cand = lane < p.A && lane != i && type_v == p.tag_notit;
                int victim = -1;
#pragma unroll
                for (int d = 3; d >= 0; --d) {                            // Location.adjacent order: up, right, down, left; the FIRST match wins
                    const int ay = cy + (d == 0 ? -1 : d == 2 ? 1 : 0), ax = cx + (d == 1 ? 1 : d == 3 ? -1 : 0);
                    const bool ain = (unsigned)ay < (unsigned)H && (unsigned)ax < (unsigned)W;
                    const unsigned long long hit = __ballot(cand && ain && here == ((uint32_t)ay | ((uint32_t)ax << 8)));
                    if (hit) victim = __builtin_ctzll(hit);
                }
                if (lane == i) pov_v = p.tag_it;
                if (victim >= 0) {
                    const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)here, victim);
                    if (lane == i) { type_v = p.tag_notit; tagj = (cj & 0xFFFFu) | (1u << 24); }
                    if (lane == victim) type_v = p.tag_it;
                }
            }
        }
        if (passed_v) lg[oaddr_v] = (uint8_t)p.default_type;
        gsync<1>();
        if constexpr (TAG) {
            if (lane < p.A) lg[passed_v ? ta_v : oaddr_v] = (uint8_t)type_v;   // every agent's cell: its FINAL type (a tag flips agents that did not move, too)
        } else {
            if (passed_v) lg[ta_v] = (uint8_t)atype_v;
        }
        double val = __shfl(vtab, (int)(jr & 31u));                     // reward = value of the target BEFORE the move
        if (!(jr & 0x100u)) val = 0.0;
        if constexpr (TAG) {
            // TagAgent.act: reward_per_turn for not being "it" once its own act is over (agents.py:100-106)
            const uint32_t after = tagj ? p.tag_notit : pov_v;
            val = (mine && after != p.tag_it) ? p.tag_reward : 0.0;
            s_rm[lane] = tagj | ((pov_v & 0xFFu) << 16);
            if (lane < p.A) p.agent_state[env * p.A + lane] = (uint8_t)type_v;
            if (mine && p.state_at_pov) p.state_at_pov[env * p.A + lane] = (uint8_t)pov_v;
        }
        s_ta[lane] = jr;                                                // journal for the render phase
        if (jr & 0x400u) st_lane |= SGW_STATUS_BAD_TYPE;
        if (mine) p.rewards[tix * p.ts_rew + env * p.A + tid] = (float)val;   // this turn's rewards
        gsync<1>();
        {   // float64, agent order (agent.py:172): the same sum in every lane, lane 0's is kept
            const uint32_t v_lo = (uint32_t)__double_as_longlong(val), v_hi = (uint32_t)(__double_as_longlong(val) >> 32);
            for (int a = p.a0; a < p.a1; ++a) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v_lo, a), hi = (uint32_t)__builtin_amdgcn_readlane((int)v_hi, a);
                tot += __longlong_as_double(((long long)hi << 32) | lo);
            }
        }
    }
    __syncthreads();
    STAMPB(2);                   // phase M done
    if constexpr (WALK) {       // a use of every prefetched register: the compiler waits for those loads HERE
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(nu[k].x), "v"(nu[k].y), "v"(nu[k].z), "v"(nu[k].w));
        asm volatile("" ::"v"(nyx), "v"(nact), "v"(ntot));
    }

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
        const uint32_t jb = s_ta[lane], srcb = s_oa[lane], dstb = s_np[lane];
        uint32_t atb = s_atype[lane];
        [[maybe_unused]] uint32_t tgb = 0u;              // TAG: victim's cell | type carried << 16 | tagged << 24
        if constexpr (TAG) {
            tgb = p.do_move ? s_rm[lane] : 0u;
            if (p.do_move) atb = (tgb >> 16) & 0xFFu;    // a move carries the type the agent had when its turn came
        }
        const bool movedb = p.do_move && (jb & 0x200u) && lane >= p.a0 && lane < p.a1;
        [[maybe_unused]] const bool taggedb = TAG && p.do_move && ((tgb >> 24) & 1u) && lane >= p.a0 && lane < p.a1;
        const int zsh = 8 * (p.zA & 3), zw = p.zA >> 2;
        // SGW_STEP_OBS_NEXT: only agent a1, which sees the grid after ALL moves of this call (nothing to undo)
        const int r_lo = p.obs_next ? p.a1 : p.a0, r_hi = p.obs_next ? (p.a1 < p.A ? p.a1 + 1 : p.a1) : p.a1;
        // (Starting each env's round of windows at another agent -- so that workgroups that started together do not all write window k of
        // their env at the same moment, addresses a fixed stride apart -- was measured in round 5: no change)
        for (int a = r_lo + wv; a < r_hi; a += kBW) {
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
            bool touches = movedb && (near_src || near_dst);
            if constexpr (TAG) {
                // a tag of agent b changes two cells: the one b stands on after its move, and its victim's
                const int vy = (int)(tgb & 0xFFu), vx = (int)((tgb >> 8) & 0xFFu);
                const bool near_vic = (unsigned)(vy - y + r) <= (unsigned)(2 * r) && (unsigned)(vx - x + r) <= (unsigned)(2 * r);
                touches = touches || (taggedb && (near_vic || (movedb ? near_dst : near_src)));
            }
            unsigned long long undo = __ballot(touches && lane >= a);
            while (undo) {
                const int b = 63 - __builtin_clzll(undo);            // latest agent first
                undo &= ~(1ull << b);
                const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)srcb, b);
                const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)dstb, b);
                const uint32_t oldt = (uint32_t)__builtin_amdgcn_readlane((int)jb, b) & 31u;    // what the target held
                const uint32_t agt = (uint32_t)__builtin_amdgcn_readlane((int)atb, b) & 31u;    // the mover itself (TAG: as it was when it acted)
                const bool mvb = ((uint32_t)__builtin_amdgcn_readlane((int)jb, b) & 0x200u) != 0;
                if constexpr (TAG) {
                    // agent b's turn was: move, then tag.  Undone in reverse: the tag first (b is "it" again where it stands
                    // after its move, the victim is NotIt again), then the move
                    const uint32_t tg = (uint32_t)__builtin_amdgcn_readlane((int)tgb, b);
                    if ((tg >> 24) & 1u) {
                        const uint32_t own = mvb ? dst : src, vic = tg & 0xFFFFu;
#pragma unroll
                        for (int k = 0; k < NP; ++k) {
                            const uint32_t key = (uint32_t)((y + wdi[k]) & 0xFF) | ((uint32_t)((x + wdj[k]) & 0xFF) << 8);
                            const bool at_own = inbk[k] && key == own, at_vic = inbk[k] && key == vic;
                            if (at_own || at_vic) {
                                const uint32_t nv = (at_own ? p.tag_it : p.tag_notit) & 31u;
                                if (zw == 0) tb[k][0] = (tb[k][0] & ~(0xFFu << zsh)) | (nv << zsh);
                                else tb[k][1] = (tb[k][1] & ~(0xFFu << zsh)) | (nv << zsh);
                            }
                        }
                    }
                    if (!mvb) continue;
                }
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
            float* obase;
            if constexpr (ROWS) {
                obase = static_cast<float*>(rp.p[a]) + env * rp.stride;                                             // sgw_sweep_observe_rows: the agent's own row
                if (p.tail_kind != SGW_TAIL_NONE) {              // what pov() appends behind the flattened window (phase.h, observe_rows: the same two kinds)
                    float* t = obase + C * VV;
                    if (p.tail_kind == SGW_TAIL_AGENT_IS_IT) {   // TagAgent.pov: [self.it] (nobody acts in this launch: the flag is the bound tensor's)
                        if (lane == 0) t[0] = (p.agent_state && p.agent_state[env * p.A + a] == p.tag_it) ? 1.f : 0.f;
                    } else {                                     // CleanupObservation.observe: the positional code of the agent's cell
                        const float* src = p.tail_table + ((int64_t)y * W + x) * p.tail_len;
                        for (int k = lane; k < p.tail_len; k += 64) t[k] = src[k];
                    }
                }
            } else {
                obase = p.obs_ag ? p.obs + tix * p.ts_obs + (int64_t)a * p.obs_ag + env * (int64_t)(C * VV)           // [A][E][C][V][V]
                                 : p.obs + tix * p.ts_obs + ((env * p.obs_A + (a - p.obs_a0)) * (int64_t)C) * VV;
            }
            if constexpr (ONEHOT) {
                // the packed byte counts of window cell lane + 64 k: one table word per layer and group of four channels
                auto counts = [&](const int k, uint32_t (&cq)[NW]) {
#pragma unroll
                    for (int q = 0; q < NW; ++q) cq[q] = 0;
                    for (int z = 0; z < L; ++z) {
                        const uint32_t t = z < 4 ? (tb[k][0] >> (8 * z)) & 31u : (tb[k][1] >> (8 * (z - 4))) & 31u;
#pragma unroll
                        for (int q = 0; q < NW; ++q) cq[q] += wdelta[q * 32 + t];
                    }
#pragma unroll
                    for (int q = 0; q < NW; ++q) cq[q] = inbk[k] ? cq[q] : p.fill_delta[q];
                };
                if (TC != 0 && p.big_stage > 0) {   // (compiled into the instances with compile-time tables only: the others spill with it)
                    // (round 3) the window's byte counts are staged in this wave's LDS area, plane by plane as they lie in the
                    // tensor, and leave as 16-byte streaming stores whose lane 0 sits on a 128-byte line of global memory
                    // (step_fast.h, emit_chunk: same scheme) instead of a dword store per lane and channel whose 484-byte runs
                    // start anywhere.  Config 5 at 8 192 envs: 434 -> 372 us per turn; 48x48 / 8 agents at 16 384: 176 -> 156.
                    // The walking variant keeps the direct stores (2 048 envs: 93 us against 104 staged).
                    typedef float vfloat4 __attribute__((ext_vector_type(4)));
                    uint8_t* ob = smem + p.big_stage_off + wv * p.big_stage;
                    const uint32_t* ob4 = reinterpret_cast<const uint32_t*>(ob);
                    // the window's first element in the tensor and its distance from a line boundary: staged byte s is element (e0 - sh) + s
                    // (a row of its own -- ROWS -- by its address: the row pointers are 4-byte aligned, no more)
                    const int64_t e0 = ROWS ? (int64_t)(reinterpret_cast<uintptr_t>(obase) >> 2) : (int64_t)(obase - p.obs);
                    const int sh = (int)(e0 & 31);
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int w = lane + 64 * k;
                        if (w < VV) {
                            uint32_t cnt[NW];
                            counts(k, cnt);
#pragma unroll
                            for (int c = 0; c < 4 * NW; ++c)
                                if (c < C) ob[sh + c * VV + w] = (uint8_t)((cnt[c >> 2] >> (8 * (c & 3))) & 0xFFu);
                        }
                    }
                    gsync<1>();
                    const int he = sh + C * VV;
                    const int i0 = (sh + 3) >> 2, i1 = he >> 2;   // dwords [i0, i1) lie wholly inside the window: 16-byte streaming stores
                    // the window's first / last float4 may be partly a neighbouring window's: element by element, by lanes 0 / 1
                    const int ie = lane == 0 ? i0 - 1 : i1;
                    const bool edge = lane == 0 ? (sh & 3) != 0 : (lane == 1 && (he & 3) != 0);
                    if (!p.obs_u8) {
                        float* gb = ROWS ? obase - sh : p.obs + (e0 - sh);
                        for (int i = lane; i < i1; i += 64) {
                            if (i < i0) continue;
                            const uint32_t b = ob4[i];
                            vfloat4 v;
                            v.x = (float)(b & 0xFFu);
                            v.y = (float)((b >> 8) & 0xFFu);
                            v.z = (float)((b >> 16) & 0xFFu);
                            v.w = (float)(b >> 24);
                            __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(gb + 4 * i));      // (ordinary stores: no difference on the kernel, profiles/r06_c5_remap_ab.txt)
                        }
                        if (edge) {
                            const uint32_t b = ob4[ie];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (4 * ie + j >= sh && 4 * ie + j < he) gb[4 * ie + j] = (float)((b >> (8 * j)) & 0xFFu);
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
                                if (4 * ie + j >= sh && 4 * ie + j < he) gb[4 * ie + j] = (uint8_t)(b >> (8 * j));
                        }
                    }
                    gsync<1>();   // the next window of this wave overwrites the staging area
                } else {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int w = lane + 64 * k;
                        if (w < VV) {
                            float* o = obase + w;
                            uint32_t cnt[NW];
                            counts(k, cnt);
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
                        }
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int w = lane + 64 * k;
                    if (w < VV) {
                        float* o = obase + w;
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
    if (dirty && !do_sweep && nturns == 1 && !TAG) {   // (a rollout's earlier turns moved other cells too; a tag flips cells of agents that did not move)
        // a policy-driven phase (no sweep): only the movers' two cells changed -- write those bytes, not the whole grid
        if (wv == 0 && mine && (jr & 0x200u)) {
            uint8_t* g = p.grid + env * p.env_stride + p.zA * H * W;
            g[(yx & 0xFFu) * W + (yx >> 8)] = lg[oaddr_v];
            g[(npos_v & 0xFFu) * W + (npos_v >> 8)] = lg[ta_v];
        }
    } else if (dirty) {
        uint4* dst = reinterpret_cast<uint4*>(p.grid + env * p.env_stride);
        for (int idx = tid; idx < nunits; idx += kBT) dst[idx] = lg16[lunit(idx)];
    }
    STAMPB(4);                   // write-back issued
#ifdef SGW_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMPB(5);                   // wave 0's stores acknowledged
    STAMPB(6);
    if (p.do_move) {
        if (wv == 0 && mine) reinterpret_cast<uint16_t*>(p.pos)[env * p.A + tid] = (uint16_t)((jr & 0x200u) ? npos_v : yx);
        if (tid == 0) p.total[env] = tot;
    }
    if constexpr (!WALK) break;
    if (next_u == 0xFFFFFFFFu) break;
    env = (int64_t)next_u;
    if (env >= p.E) break;
    __syncthreads();             // the write-back has read this env's LDS image: the next env may overwrite it
    } while (true);
    if (tid0 < 64 && st_lane) atomicOr(p.status, st_lane);
}


// resolve.h -- part of the single translation unit sgw.hip (included inside its anonymous namespace).
// turn_resolve<ONEHOT>: one pass of a SPECULATIVE policy turn (sgw_turn_resolve).
#pragma once

// ---------------------------------------------------------------- speculative policy turn
// The reference steps its agents strictly one after another (sorrel/agents/agent.py:155-173: pov -> get_action -> act): agent j's
// window shows the moves of the agents before it, so a turn of A policy-driven agents is A dependent (forward, act) pairs -- 64 x 2
// launches for BASELINE config 5.  But a move changes two cells, and a window is small: for most (env, agent) pairs no earlier agent
// moves inside the window, and the action computed from the PRE-move window is already the sequential one.  So:
//   pass 1   every agent's window from the grid before anyone moves (sgw_observe_rows), ONE batched policy evaluation;
//   resolve  (this kernel, a wave per env) resolves all moves of the env in agent order from the current actions, without writing
//            the grid; for every agent whose window an earlier mover touches it renders the window the agent REALLY has when its
//            turn comes (pre-move cells + the earlier movers' changes) and compares it with the row its action was computed on:
//            different -> the row is rewritten and the agent marked dirty;
//   pass k   the host re-evaluates the policy on the dirty rows only, and resolves again.
// An env in which nobody is dirty has reached the fixed point -- every action was computed on the window the sequential loop
// would have shown its agent -- and is committed right away (movers' cells, positions, rewards, the float64 total in agent order);
// it is skipped by later passes.  The first dirty agent of an env moves to a higher index every pass: at most A passes, typically
// two or three (profiles/r05_speculation_study.txt: config 5, 99th percentile 3; a fifth of the pairs re-evaluated in pass 2, 0.3 % in
// pass 3).  Plain movers (MovingAgent.act) with impassable agent types; results are those of the sequential turn, bit for bit.
struct ResolveArgs {
    float* rows;           // [A][E][row_elems] float32: the window each agent's action was computed on (agent-major: a group of agents
                           // that share a model is one contiguous batch)
    int64_t row_elems;     // elements per env of a row (>= C * V * V; what lies behind the window is the caller's)
    uint8_t* env_done;     // [E]: 1 = the env has reached its fixed point this turn and is committed
    uint8_t* pristine;     // [E][A]: 1 = the agent's row holds the pre-move window (no earlier mover touched it when it was written)
    uint8_t* dirty;        // [E][A] out: 1 = the row was rewritten by this pass: evaluate the policy on it again
    uint8_t* prev;         // [E][A]: action | moved << 7 of the previous pass -- a later pass looks only at the windows that an agent whose
                           // move CHANGED since then touches (before or after the change); every other row is still what it was verified to be
    int64_t* list;         // out: the dirty rows of this pass as indices a * E + env, in no particular order (NULL: only the `dirty` bytes)
    uint32_t* count;       // ... and how many (zero when the launch starts; one wave-aggregated atomic per env that has any)
    uint8_t* dcount;       // [E], instead of list / count for large batches: how many rows of the env this pass left dirty -- the list is then laid
    uint32_t* bsum;        // out by a scan over these and over their sums per block of 256 envs (resolve_scan_blocks / resolve_fill_list): tens of
                           // thousands of atomics on ONE counter serialise (config 3's 65 536 envs: 575 us for the first pass)
    uint32_t* count_next;  // the counter the NEXT pass will use: zeroed by this launch
    float* reward_rows;    // optional [A][E] float32 / int64: at the commit of an env, reward and action of every agent once more in
    int64_t* action_rows;  // agent-major rows (the rows of a replay ring: add_memory then copies nothing)
    int diag;              // timing aid (option resolve_diag): bit 0 = verify no window, bit 1 = commit nothing and append nothing (the pass can be
                           // repeated on the same state), bit 2 = skip the move resolution and the touch tests
    int first;             // 1: first pass of the turn -- env_done / pristine are taken as 0 / 1 whatever the arrays hold;
                           // 2: no pass at all -- render every agent's PRE-move window into its row (what sgw_observe_rows does for
                           //    one-hot worlds, here for any appearance table and window size) and initialise the arrays
};

// The policy's output for the rows a pass marked dirty (or, `list` NULL, for every row in agent-major order) -> actions[env][agent].
__global__ __launch_bounds__(kBlock) void resolve_apply_actions(uint8_t* actions, const int64_t* list, const int64_t* fresh, const int64_t n,
                                                                const int64_t E, const int A) {
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
        const int64_t row = list ? list[k] : k;
        const int64_t a = row / E, e = row - a * E;
        const int64_t v = fresh[k];
        actions[e * A + a] = (v >= 0 && v < 255) ? (uint8_t)v : (uint8_t)255;      // (an index no ActionSpec has: SGW_STATUS_BAD_ACTION at the resolve)
    }
}

// sgw_verify_rows (round 6): the resolve of a speculative turn for ANY agent rule.  `obs` [E][A][N] holds the windows a sequential turn with the
// current actions showed its agents at their pov (sgw_step with given actions on a scratch copy of the state), `rows` [A][E][R] what the actions were
// computed on (window + tail).  A wave per (env, agent): where the window -- or Tag's "it" flag behind it -- differs, the row is rewritten, its index
// a * E + env appended to `list` (one atomic per dirty row: this path is for the small batches where a speculative turn pays at all).
__global__ __launch_bounds__(kBlock) void verify_rows_kernel(const float* __restrict__ obs, const uint8_t* __restrict__ state_at_pov, float* __restrict__ rows,
                                                             const int64_t R, const int N, const int64_t E, const int A, const int tail_it, const uint32_t it_type,
                                                             int64_t* __restrict__ list, uint32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (w >= E * A) return;
    const int64_t env = w / A;
    const int a = (int)(w - env * A);
    const float* src = obs + w * N;
    float* dst = rows + ((int64_t)a * E + env) * R;
    bool diff = false;
    for (int i = lane; i < N; i += 64) diff = diff || src[i] != dst[i];
    float flag = 0.f;
    if (tail_it) {
        flag = state_at_pov[w] == it_type ? 1.f : 0.f;
        diff = diff || (lane == 0 && dst[N] != flag);
    }
    if (__ballot(diff) == 0ull) return;
    for (int i = lane; i < N; i += 64) dst[i] = src[i];
    if (lane == 0) {
        if (tail_it) dst[N] = flag;
        list[atomicAdd(count, 1u)] = (int64_t)a * E + env;
    }
}

// Large batches: the dirty list without tens of thousands of atomics on one counter.  The resolve kernel leaves the number of dirty rows of each
// env in dcount[env] and adds it to the sum of its block of 256 envs (256 envs share a counter: the atomics spread over E / 256 addresses);
// one workgroup turns the block sums into block offsets (and the total); a workgroup per block then scans its 256 counts and writes the
// dirty rows' indices (agent * E + env) behind the block's offset.
__global__ __launch_bounds__(256) void resolve_scan_blocks(uint32_t* __restrict__ bsum, const int nb, uint32_t* __restrict__ boff, uint32_t* __restrict__ count) {
    __shared__ uint32_t part[256];
    const int t = threadIdx.x;
    const int per = (nb + 255) / 256, lo = t * per, hi = lo + per < nb ? lo + per : nb;
    uint32_t sum = 0;
    for (int b = lo; b < hi; ++b) sum += bsum[b];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                       // inclusive scan of the 256 partial sums
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int b = lo; b < hi; ++b) {
        const uint32_t v = bsum[b];
        boff[b] = run;
        bsum[b] = 0u;                                         // (for the next pass)
        run += v;
    }
    if (t == 255) *count = part[255];
}

__global__ __launch_bounds__(256) void resolve_fill_list(const uint8_t* __restrict__ dirty, const uint8_t* __restrict__ dcount, const uint8_t* __restrict__ env_done,
                                                         const uint32_t* __restrict__ boff, const int64_t E, const int A, int64_t* __restrict__ list) {
    __shared__ uint32_t part[256];
    const int t = threadIdx.x;
    const int64_t env = (int64_t)blockIdx.x * 256 + t;
    const uint32_t mine = (env < E && !env_done[env]) ? (uint32_t)dcount[env] : 0u;
    part[t] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (mine == 0) return;
    uint32_t at = boff[blockIdx.x] + part[t] - mine;
    for (int a = 0; a < A; ++a)
        if (dirty[env * A + a]) list[at++] = (int64_t)a * E + env;
}

// dst[k][:] = src[idx[k]][:]: a wave per row, 8-byte pieces where rows and pointers allow (sgw_gather_rows).
template <int VEC>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ src, const int64_t row_elems, const int64_t* __restrict__ idx,
                                                             const int64_t n, float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t k = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); k < n; k += waves) {
        const int64_t r = idx[k];
        if constexpr (VEC == 2) {
            typedef float vfloat2 __attribute__((ext_vector_type(2)));
            const vfloat2* s2 = reinterpret_cast<const vfloat2*>(src + r * row_elems);
            vfloat2* d2 = reinterpret_cast<vfloat2*>(dst + k * row_elems);
            for (int64_t i = lane; i < (row_elems >> 1); i += 64) __builtin_nontemporal_store(s2[i], d2 + i);
        } else {
            const float* s1 = src + r * row_elems;
            float* d1 = dst + k * row_elems;
            for (int64_t i = lane; i < row_elems; i += 64) d1[i] = s1[i];
        }
    }
}

// WPE: waves per env.  1: four envs per 256-thread workgroup (few agents: the turn of an env is little work).  4: a workgroup per env --
// wave 0 resolves the env's moves (a loop over the agents: sequential by nature) and publishes who moved and which windows must be
// verified through LDS; the windows (each a dependent chain of loads) are dealt out among the four waves.
#ifndef SGW_RESOLVE_SLOTS
#define SGW_RESOLVE_SLOTS 1
#endif
template <bool ONEHOT, int WPE>
__global__ __launch_bounds__(kBlock, 8) void turn_resolve(const Params p, const ResolveArgs ra) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t env = WPE == 1 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
    if (blockIdx.x == 0 && tid == 0 && ra.count_next) *ra.count_next = 0u;
    if (env >= p.E) return;
    if (!ra.first && ra.env_done[env]) return;               // (uniform over the env's waves)
    const bool render_only = ra.first == 2;
    const bool lead = WPE == 1 || wave == 0;                  // the wave that resolves, keeps the books and commits
    const DevTables* gtab = p.tab;
    const int H = p.H, W = p.W, HW = H * W, L = p.L, C = p.C, V = p.V, VV = p.VV, r = p.r, A = p.A;
    uint8_t* g = p.grid + env * p.env_stride;
    uint8_t* ga = g + p.zA * HW;                              // the agent layer
    // wave-private tables: [one-hot counter words | appearance f64]
    constexpr int kTab = ONEHOT ? 4 * SGW_MAX_TYPES * 4 : SGW_MAX_TYPES * SGW_MAX_CHANNELS * 8;
    uint8_t* wl = smem + wave * kTab;                         // (every wave keeps its own copy: no barrier needed for them)
    unsigned long long* sh = reinterpret_cast<unsigned long long*>(smem + 4 * kTab);   // WPE > 1: [0] who moved, [1] windows to verify, [2 ..] what each wave found
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

    // ---- lane a = agent a (in every wave): where it stands, where its current action takes it
    const bool live = lane < A;
    int st = 0;
    uint32_t yx = 0, act = 0, atype = 0;
    uint32_t prev = 0;
    if (live) {
        yx = reinterpret_cast<const uint16_t*>(p.pos)[env * A + lane];
        act = p.actions[env * A + lane];
        atype = gtab->agent_type[lane];
        if (!ra.first) prev = ra.prev[env * A + lane];           // (every wave: each takes its share of the agents whose move changed)
        if ((yx & 0xFFu) >= (uint32_t)H || (yx >> 8) >= (uint32_t)W) { yx = 0; st |= SGW_STATUS_BAD_POS; }
    }
    const int py = (int)(yx & 0xFFu), px = (int)(yx >> 8);
    const uint32_t oaddr = (uint32_t)(py * W + px);
    uint32_t ta = 0xFFFFFFFFu, npos = yx;
    if (live) {
        const bool act_ok = act < (uint32_t)p.nact;
        const int dy = (int)((p.dy_pack >> (2 * (act & 15u))) & 3u) - 1, dx = (int)((p.dx_pack >> (2 * (act & 15u))) & 3u) - 1;
        const int ty = py + dy, tx = px + dx;
        const bool inb = (unsigned)ty < (unsigned)H && (unsigned)tx < (unsigned)W;
        if (act_ok && inb) {
            ta = (uint32_t)(ty * W + tx);
            npos = (uint32_t)ty | ((uint32_t)tx << 8);
        }
        st |= !act_ok ? SGW_STATUS_BAD_ACTION : (!inb ? SGW_STATUS_OOB_MOVE : 0);
    }
    const int ny = (int)(npos & 0xFFu), nx = (int)(npos >> 8);
    gsync<1>();                                               // the table words are visible to every lane of this wave

    uint32_t passed = 0, found = 0xFFu;                       // lane a: did agent a move; the type it found on its target
    bool tok_v = false;
    if (lead && !render_only && !(ra.diag & 4)) {
        // ---- the moves, strictly in agent order (sorrel/agents/agent.py:213-225, worlds/gridworld.py:95-122), in registers.  What agent a
        // finds on its target differs from the pre-move grid only if an earlier mover entered or left that very cell: an agent INTERFERES
        // if another agent has the same target or stands on it.  Everyone else resolves at once from the pre-move grid; only the
        // interfering agents (typically none, or a pair) are walked in agent order -- the latest earlier mover that entered or left the
        // cell decides.  (Until the walk was restricted to them the loop over all 64 agents of config 5 was a third of the pass.)
        const uint32_t t0 = ta != 0xFFFFFFFFu ? (uint32_t)ga[ta] : 0xFFu;
        const bool valid = ta != 0xFFFFFFFFu;
        bool cf = false;
        for (int b = 0; b < A; ++b) {
            const uint32_t Xb = (uint32_t)__builtin_amdgcn_readlane((int)ta, b), Ob = (uint32_t)__builtin_amdgcn_readlane((int)oaddr, b);
            cf = cf || (lane != b && (ta == Xb || ta == Ob));
        }
        cf = cf && valid && live;
        {
            const bool tok = valid && t0 < (uint32_t)p.T;
            found = t0;
            tok_v = tok;
            passed = (tok && ((p.pass_mask >> (t0 & 31u)) & 1u)) ? 1u : 0u;
            if (valid && !tok) st |= SGW_STATUS_BAD_TYPE;
        }
        unsigned long long walk = __ballot(cf);
        while (walk) {
            const int a = __builtin_ctzll(walk);
            walk &= walk - 1ull;
            const uint32_t X = (uint32_t)__builtin_amdgcn_readlane((int)ta, a);
            uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)t0, a);
            const unsigned long long m_dst = __ballot(passed && lane < a && ta == X);
            const unsigned long long m_src = __ballot(passed && lane < a && oaddr == X);
            const unsigned long long m_any = m_dst | m_src;
            if (m_any) {
                const int last = 63 - __builtin_clzll(m_any);
                t = ((m_dst >> last) & 1ull) ? (uint32_t)__builtin_amdgcn_readlane((int)atype, last) : p.default_type;
            }
            const bool tok = t < (uint32_t)p.T;
            const bool pass = tok && ((p.pass_mask >> (t & 31u)) & 1u);
            if (lane == a) {
                found = t;
                tok_v = tok;
                passed = pass ? 1u : 0u;
                if (!tok) st |= SGW_STATUS_BAD_TYPE;
            }
        }
    }
    if constexpr (WPE > 1) {
        const unsigned long long moved = __ballot(passed != 0);   // (taken with every lane active; lane 0 then writes it)
        if (wave == 0 && lane == 0) sh[0] = moved;
        __syncthreads();
        if (wave != 0) passed = (uint32_t)((sh[0] >> lane) & 1ull);
    }
    const uint32_t cur = (act & 0x7Fu) | (passed << 7);

    // ---- which windows must be verified.  First pass: those an earlier mover touches (agent j still stands where the turn began when
    // its turn comes).  Later passes: only an agent whose move CHANGED since the previous pass (another action, or the same action with
    // another outcome) can have changed a window -- one it touched before (its source cell, its old destination) or touches now.
    // The loop over the movers is dealt out among the env's waves (every WPE-th mover each); the partial findings meet in LDS.
    bool touched = false, recheck = false;
    auto touch_of = [&](unsigned long long movers) {          // does a mover of `movers` touch lane j's window?
        bool t = false;
        int nth = 0;
        while (movers) {
            const int i = __builtin_ctzll(movers);
            movers &= movers - 1ull;
            if (WPE > 1 && (nth++ % WPE) != wave) continue;
            const int sy = __builtin_amdgcn_readlane(py, i), sx = __builtin_amdgcn_readlane(px, i);
            const int ey = __builtin_amdgcn_readlane(ny, i), ex = __builtin_amdgcn_readlane(nx, i);
            const bool near_s = (unsigned)(sy - py + r) <= (unsigned)(2 * r) && (unsigned)(sx - px + r) <= (unsigned)(2 * r);
            const bool near_e = (unsigned)(ey - py + r) <= (unsigned)(2 * r) && (unsigned)(ex - px + r) <= (unsigned)(2 * r);
            t = t || (lane > i && (near_s || near_e));
        }
        return t;
    };
    if (!render_only && !(ra.diag & 4)) {
        const unsigned long long movers = __ballot(passed != 0);
        if (ra.first == 1) {
            touched = touch_of(movers);
        } else {
            unsigned long long changed = __ballot(live && prev != cur);
            int nth = 0;
            while (changed) {
                const int i = __builtin_ctzll(changed);
                changed &= changed - 1ull;
                if (WPE > 1 && (nth++ % WPE) != wave) continue;
                const int sy = __builtin_amdgcn_readlane(py, i), sx = __builtin_amdgcn_readlane(px, i);
                const uint32_t pv = (uint32_t)__builtin_amdgcn_readlane((int)prev, i), cv = (uint32_t)__builtin_amdgcn_readlane((int)cur, i);
                const int ody = (int)((p.dy_pack >> (2 * (pv & 15u))) & 3u) - 1, odx = (int)((p.dx_pack >> (2 * (pv & 15u))) & 3u) - 1;
                const int oy = sy + ody, ox = sx + odx;                        // where it went before (if it moved then)
                const int ey = __builtin_amdgcn_readlane(ny, i), ex = __builtin_amdgcn_readlane(nx, i);
                const bool near_s = (unsigned)(sy - py + r) <= (unsigned)(2 * r) && (unsigned)(sx - px + r) <= (unsigned)(2 * r);
                const bool near_o = (unsigned)(oy - py + r) <= (unsigned)(2 * r) && (unsigned)(ox - px + r) <= (unsigned)(2 * r);
                const bool near_e = (unsigned)(ey - py + r) <= (unsigned)(2 * r) && (unsigned)(ex - px + r) <= (unsigned)(2 * r);
                const bool was_moving = (pv >> 7) != 0, is_moving = (cv >> 7) != 0;
                recheck = recheck || (lane > i && (((was_moving || is_moving) && near_s) || (was_moving && near_o) || (is_moving && near_e)));
            }
        }
    }
    unsigned long long check;
    if constexpr (WPE > 1) {
        const unsigned long long part = __ballot(ra.first == 1 ? touched : recheck);
        if (lane == 0) sh[2 + wave] = part;
        __syncthreads();
        unsigned long long all = 0ull;
#pragma unroll
        for (int k = 0; k < WPE; ++k) all |= sh[2 + k];
        if (ra.first == 1) touched = (all >> lane) & 1ull;
        else recheck = (all >> lane) & 1ull;
        __syncthreads();                                      // (sh[2 ..] is used again below)
    }
    bool checked = live && (render_only || (ra.first == 1 ? touched : recheck));
    if (ra.first == 0 && __ballot(checked)) {                 // (rare in later passes) is a checked window touched NOW?  -> its `pristine` flag
        const bool part = touch_of(__ballot(passed != 0));
        if constexpr (WPE > 1) {
            const unsigned long long pm = __ballot(part);
            if (lane == 0) sh[2 + wave] = pm;
            __syncthreads();
            unsigned long long all = 0ull;
#pragma unroll
            for (int k = 0; k < WPE; ++k) all |= sh[2 + k];
            touched = (all >> lane) & 1ull;
            __syncthreads();
        } else {
            touched = part;
        }
    }
    check = (ra.diag & 1) ? 0ull : __ballot(checked);

    // ---- the windows that may differ from their rows: rendered as the agent really has them, compared, rewritten where they differ.
    // kSlots rounds of 64 cells at a time; every load of a round -- the row's old values first, then the grid bytes -- is issued before
    // the first is used: one round trip to memory per round where the straightforward loop had half a dozen.
    constexpr int kSlots = SGW_RESOLVE_SLOTS;
    unsigned long long dmask = 0ull;
    int nth = 0;
    while (check) {
        const int j = __builtin_ctzll(check);
        check &= check - 1ull;
        if (WPE > 1 && (nth++ % WPE) != wave) continue;       // this window is another wave's
        const int y = __builtin_amdgcn_readlane(py, j), x = __builtin_amdgcn_readlane(px, j);
        // the earlier movers whose source or destination lies in window j, applied in agent order (a cell can be left and then entered)
        const bool near_s = (unsigned)(py - y + r) <= (unsigned)(2 * r) && (unsigned)(px - x + r) <= (unsigned)(2 * r);
        const bool near_e = (unsigned)(ny - y + r) <= (unsigned)(2 * r) && (unsigned)(nx - x + r) <= (unsigned)(2 * r);
        const unsigned long long tmask = __ballot(passed && lane < j && (near_s || near_e));
        float* rowp = ra.rows + ((int64_t)j * p.E + env) * ra.row_elems;
        bool diff = false;
        for (int w0 = 0; w0 < VV; w0 += 64 * kSlots) {
            bool inw[kSlots], inb[kSlots];
            uint32_t cell[kSlots];
            float oldv[kSlots][SGW_MAX_CHANNELS];
            uint32_t raw[kSlots][SGW_MAX_LAYERS];
#pragma unroll
            for (int k = 0; k < kSlots; ++k) {
                const int w = w0 + 64 * k + lane;
                inw[k] = w < VV;
                const int wi = w / V, wj = w - wi * V;
                const int gy = y - r + wi, gx = x - r + wj;
                inb[k] = inw[k] && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                cell[k] = inb[k] ? (uint32_t)(gy * W + gx) : 0u;
#pragma unroll
                for (int c = 0; c < SGW_MAX_CHANNELS; ++c)
                    if (c < C && inw[k]) oldv[k][c] = __builtin_nontemporal_load(rowp + c * VV + w);
            }
#pragma unroll
            for (int k = 0; k < kSlots; ++k) {
#pragma unroll
                for (int z = 0; z < SGW_MAX_LAYERS; ++z)
                    if (z < L) raw[k][z] = inb[k] ? (uint32_t)g[z * HW + cell[k]] : 0u;
            }
#pragma unroll
            for (int k = 0; k < kSlots; ++k) {
                if (!inw[k]) continue;
                const int w = w0 + 64 * k + lane;
                if (inb[k]) {                                 // the agent layer as the agents before j have left it
                    unsigned long long m = tmask;
                    while (m) {
                        const int b = __builtin_ctzll(m);
                        m &= m - 1ull;
                        const uint32_t src = (uint32_t)__builtin_amdgcn_readlane((int)oaddr, b);
                        const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)ta, b);
                        const uint32_t bt = (uint32_t)__builtin_amdgcn_readlane((int)atype, b);
#pragma unroll
                        for (int z = 0; z < SGW_MAX_LAYERS; ++z)
                            if (z == p.zA) {
                                if (cell[k] == src) raw[k][z] = p.default_type;
                                if (cell[k] == dst) raw[k][z] = bt;
                            }
                    }
                }
                if constexpr (ONEHOT) {
                    uint32_t cnt[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int z = 0; z < SGW_MAX_LAYERS; ++z)
                        if (z < L) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) cnt[q] += wdelta[q * 32 + (raw[k][z] & 31u)];
                        }
#pragma unroll
                    for (int q = 0; q < 4; ++q) cnt[q] = inb[k] ? cnt[q] : p.fill_delta[q];
#pragma unroll
                    for (int c = 0; c < SGW_MAX_CHANNELS; ++c)
                        if (c < C) {
                            const float v = (float)((cnt[c >> 2] >> (8 * (c & 3))) & 0xFFu);
                            if (oldv[k][c] != v) { rowp[c * VV + w] = v; diff = true; }
                        }
                } else {
#pragma unroll
                    for (int c = 0; c < SGW_MAX_CHANNELS; ++c)
                        if (c < C) {
                            double acc = wapp[raw[k][0] & 31u][c];   // left-to-right float64 layer sum (np.sum over <= 7 layers)
#pragma unroll
                            for (int z = 1; z < SGW_MAX_LAYERS; ++z)
                                if (z < L) acc += wapp[raw[k][z] & 31u][c];
                            const float v = obs_finish(inb[k] ? acc : wapp[p.fill_type][c], p.obs_post);
                            const float old = oldv[k][c];
                            if (!(old == v) && !(old != old && v != v)) { rowp[c * VV + w] = v; diff = true; }   // (a NaN appearance equals itself here)
                        }
                }
            }
        }
        if (__ballot(diff)) dmask |= 1ull << j;
    }

    // ---- bookkeeping of the pass (WPE > 1: the waves' findings meet in LDS; wave 0 does the rest)
    if constexpr (WPE > 1) {
        if (lane == 0) sh[2 + wave] = dmask;
        __syncthreads();
        if (wave != 0) return;
        dmask = 0ull;
#pragma unroll
        for (int k = 0; k < WPE; ++k) dmask |= sh[2 + k];
    }
    if (render_only) {                                        // the rows hold the pre-move windows: nothing resolved, nothing committed
        if (live) {
            ra.pristine[env * A + lane] = 1;
            ra.dirty[env * A + lane] = 0;
            ra.prev[env * A + lane] = 0;
        }
        if (lane == 0) ra.env_done[env] = 0;
        return;
    }
    if (live) {
        if (checked) ra.pristine[env * A + lane] = touched ? (uint8_t)0 : (uint8_t)1;   // (a checked row holds its true window now; the others are what they were)
        else if (ra.first) ra.pristine[env * A + lane] = 1;
        ra.dirty[env * A + lane] = (uint8_t)((dmask >> lane) & 1ull);
        ra.prev[env * A + lane] = (uint8_t)cur;
    }
    if (ra.diag & 2) return;
    if (ra.dcount && lane == 0) {
        const uint32_t n = (uint32_t)__builtin_popcountll(dmask);
        ra.dcount[env] = (uint8_t)n;
        if (n) atomicAdd(ra.bsum + (env >> 8), n);
    }
    if (dmask) {                                              // somebody must think again: nothing of this env is committed
        if (lane == 0 && ra.first) ra.env_done[env] = 0;
        if (ra.list) {                                        // the dirty rows of the env behind the others': ONE atomic per env
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(ra.count, (uint32_t)__builtin_popcountll(dmask));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if ((dmask >> lane) & 1ull)
                ra.list[base + (uint32_t)__builtin_popcountll(dmask & ((1ull << lane) - 1ull))] = (int64_t)lane * p.E + env;
        }
        return;
    }

    // ---- the fixed point: commit the turn of this env.  Every mover's old cell <- default, then every mover's new cell <- its type
    // (a cell can be left and then entered in one turn, never the other way round: an agent moves once)
    if (passed) ga[oaddr] = (uint8_t)p.default_type;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gsync<1>();
    if (passed) ga[ta] = (uint8_t)atype;
    double val = (tok_v && found < (uint32_t)SGW_MAX_TYPES) ? gtab->value[found & 31u] : 0.0;   // reward = value of the target BEFORE the move
    if (!live) val = 0.0;
    if (live) {
        p.rewards[env * A + lane] = (float)val;
        if (passed) reinterpret_cast<uint16_t*>(p.pos)[env * A + lane] = (uint16_t)npos;
        if (ra.reward_rows) ra.reward_rows[(int64_t)lane * p.E + env] = (float)val;
        if (ra.action_rows) ra.action_rows[(int64_t)lane * p.E + env] = (int64_t)act;
    }
    {   // float64, agent order (agent.py:172)
        double tot = lane == 0 ? p.total[env] : 0.0;
        const uint32_t v_lo = (uint32_t)__double_as_longlong(val), v_hi = (uint32_t)(__double_as_longlong(val) >> 32);
        for (int a = 0; a < A; ++a) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v_lo, a), hi = (uint32_t)__builtin_amdgcn_readlane((int)v_hi, a);
            tot += __longlong_as_double(((long long)hi << 32) | lo);
        }
        if (lane == 0) {
            p.total[env] = tot;
            ra.env_done[env] = 1;
        }
    }
    if (st) atomicOr(p.status, st);
}

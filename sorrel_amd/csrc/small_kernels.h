// small_kernels.h -- part of the single translation unit sgw.hip (included inside its anonymous namespace).
// reset_kernel (create_world + populate_environment), random actions, agent-state initialisation, the metric reduction.
#pragma once

// ---------------------------------------------------------------- reset kernel
// create_world + populate_environment (gridworld.py:47-65, treasurehunt/env.py:114-147).  The fill + border image of an
// env is the same for every env and epoch: sgw_create builds it once on the host (`tmpl`), the kernel copies it into LDS
// with 16-byte loads and adds what differs per env -- the optional dense pre-seeding (one Philox block per dword of the
// agent layer) and the agents, placed by sequential sampling without replacement.
//
// Placement, wave-parallel.  The reference order is: agent i draws d0 uniformly from the n - i cells still free, then
// maps it to the d0-th FREE cell by walking the ascending list of taken cells (`for j < i: if (d >= taken[j]) ++d`).
// With lane j holding the j-th smallest taken cell, the cells that bump d are exactly a prefix of that list, and
// `taken[j] - j <= d0` is monotone in j, so their number is one ballot + popcount; d = d0 + k lands at sorted position k
// (one lane shift).  O(A) wave instructions instead of thread 0's O(A^2) loop (config 5, 64 agents: the reset kernel took
// 512 us; the old per-byte fill with two divisions per cell took most of config 3's 249 us).
template <int WPE>
__global__ __launch_bounds__(kBlock) void reset_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int G = WPE * kWave;
    constexpr int EPB = kBlock / G;
    const int tid = threadIdx.x;
    const int sub = tid / G;
    const int gtid = tid - sub * G;
    const int lane = tid & 63;
    uint8_t* lg = smem + sub * p.env_lds;
    const int HW = p.H * p.W;
    const int zoff = p.zA * HW;
    const int64_t env = (int64_t)blockIdx.x * EPB + sub;      // one env per group and launch
    if (env >= p.E) return;                                     // whole groups leave together (a group is >= one wave)
    const uint32_t env_id = p.first_env + (uint32_t)env;
    {   // the fill + border image (pad bytes 0xFF)
        const uint4* s = reinterpret_cast<const uint4*>(p.tmpl);
        uint4* d = reinterpret_cast<uint4*>(lg);
        for (int i = gtid; i < (p.cells_pad >> 4); i += G) d[i] = s[i];
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
                    lg[4 * d + b] = p.tab->dense_choice[(uint32_t)(((uint64_t)word_of(k, b) * (uint32_t)p.dense_count) >> 32)];
        }
        gsync<WPE>();
    }
    // agents: the first wave of the group, lane a = agent a (and, beyond 64 agents -- round 6 --, agent 64 + a in a second set of registers:
    // the sorted list of taken cells is then 128 long, positions 0..63 in `taken`, 64..127 in `taken2`)
    if (gtid < kWave) {
        uint32_t u = 0, u2 = 0;
        if (lane < p.A) {
            const U4 w = philox4x32_10((uint32_t)lane >> 2, 0u, env_id, (p.epoch << 4) | SGW_STREAM_PLACE, p.seed_lo, p.seed_hi);
            u = word_of(w, lane & 3);
        }
        if (64 + lane < p.A) {
            const U4 w = philox4x32_10((uint32_t)(64 + lane) >> 2, 0u, env_id, (p.epoch << 4) | SGW_STREAM_PLACE, p.seed_lo, p.seed_hi);
            u2 = word_of(w, lane & 3);
        }
        const int n = (p.H - 2) * (p.W - 2);
        const int iw = p.W - 2;
        int taken = 0x7FFFFFFF, taken2 = 0x7FFFFFFF;   // lane j: the j-th (64 + j-th) smallest taken interior index (valid below i)
        int mine = 0, mine2 = 0;                       // lane i: agent i's (agent 64 + i's) interior index
        for (int i = 0; i < p.A; ++i) {
            const uint32_t ui = i < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)u, i) : (uint32_t)__builtin_amdgcn_readlane((int)u2, i - 64);
            const int d0 = (int)__umulhi(ui, (uint32_t)(n - i));
            int k = __popcll(__ballot(lane < i && taken - lane <= d0));
            if (i > 64) k += __popcll(__ballot(64 + lane < i && taken2 - (64 + lane) <= d0));
            const int d = d0 + k;
            const int below = __shfl_up(taken, 1);                               // lane j: taken[j - 1]
            if (i >= 64) {                                                       // the upper half shifts too; taken[63] crosses over
                const int below2 = __shfl_up(taken2, 1);
                const int carry = __builtin_amdgcn_readlane(taken, 63);
                const int pos2 = 64 + lane;
                taken2 = pos2 > k ? (lane == 0 ? carry : below2) : (pos2 == k ? d : taken2);
                mine2 = lane == i - 64 ? d : mine2;
            } else {
                mine = lane == i ? d : mine;
            }
            taken = lane > k ? below : (lane == k ? d : taken);                  // insert d at sorted position k
        }
        if (lane < p.A) {
            const int y = 1 + mine / iw, x = 1 + mine - (mine / iw) * iw;
            lg[zoff + y * p.W + x] = p.agent_state ? p.agent_state[env * p.A + lane] : p.tab->agent_type[lane];
            reinterpret_cast<uint16_t*>(p.pos)[env * p.A + lane] = (uint16_t)((uint32_t)y | ((uint32_t)x << 8));
        }
        if (64 + lane < p.A) {
            const int a2 = 64 + lane;
            const int y = 1 + mine2 / iw, x = 1 + mine2 - (mine2 / iw) * iw;
            lg[zoff + y * p.W + x] = p.agent_state ? p.agent_state[env * p.A + a2] : p.tab->agent_type[a2];
            reinterpret_cast<uint16_t*>(p.pos)[env * p.A + a2] = (uint16_t)((uint32_t)y | ((uint32_t)x << 8));
        }
        if (lane == 0) p.total[env] = 0.0;
    }
    gsync<WPE>();
    store_grid<G>(p, p.grid + env * p.env_stride, lg, gtid);
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


// ---------------------------------------------------------------- whole-map observation (full_view)
// ObservationSpec(full_view=True).observe -> visual_field(location=None) (observation_spec.py:140-142, 197-203;
// visual_field.py:41-55): every cell's appearance summed over the layers, [C][H][W] per env, no shift / crop / fill.
// Thread = cell; consecutive threads read consecutive bytes of each layer and write consecutive floats of each channel
// plane.  One-hot tables use the packed byte counters, anything else the float64 layer sum (left to right, as np.sum
// over <= 7 layers) with the spec's post-processing.
__global__ __launch_bounds__(kBlock) void observe_full_kernel(const Params p, void* out) {
    const int HW = p.H * p.W;
    const int64_t n = p.E * (int64_t)HW;
    const DevTables* tab = p.tab;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t env = i / HW;
        const int cell = (int)(i - env * HW);
        const uint8_t* g = p.grid + env * p.env_stride + cell;
        const int64_t o = env * p.C * (int64_t)HW + cell;
        if (p.onehot) {
            uint32_t cnt[4] = {0u, 0u, 0u, 0u};
            for (int z = 0; z < p.L; ++z) {
                const uint32_t t = g[z * HW] & 31u;
#pragma unroll
                for (int q = 0; q < 4; ++q) cnt[q] += tab->delta[q][t];
            }
            for (int c = 0; c < p.C; ++c) {
                const uint32_t v = (cnt[c >> 2] >> (8 * (c & 3))) & 0xFFu;
                if (p.obs_u8) reinterpret_cast<uint8_t*>(out)[o + (int64_t)c * HW] = (uint8_t)v;
                else reinterpret_cast<float*>(out)[o + (int64_t)c * HW] = (float)v;
            }
        } else {
            for (int c = 0; c < p.C; ++c) {
                double acc = tab->appearance[g[0] & 31u][c];
                for (int z = 1; z < p.L; ++z) acc += tab->appearance[g[z * HW] & 31u][c];
                reinterpret_cast<float*>(out)[o + (int64_t)c * HW] = obs_finish(acc, p.obs_post);
            }
        }
    }
}

// ---------------------------------------------------------------- device-side turn state (sgw_turn_*)
__global__ void turn_epsilon_kernel(TurnState* ts, const int agent, const int A, const uint64_t thr) {
    const int a = threadIdx.x;
    if (a < A && (agent < 0 || a == agent)) ts->eps_thr[a] = thr;
}
// sgw_choose_actions: out[k] = the action the sequential turn's sgw_act(SGW_ACT_QF32) takes for row idx[k] = agent * E + env, from that row's
// action values q[k][:] -- first index of the maximum, or with probability epsilon[agent] the engine's own draw for (env, turn, agent).
__global__ __launch_bounds__(kBlock) void choose_actions_kernel(const TurnState* __restrict__ ts, const float* __restrict__ q, const int nact,
                                                                const int64_t* __restrict__ idx, const int64_t n, const int64_t E,
                                                                const uint32_t first_env, const uint32_t epoch, const uint32_t turn,
                                                                const uint32_t seed_lo, const uint32_t seed_hi, int64_t* __restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
        const int64_t row = idx ? idx[k] : k;
        const int a = (int)(row / E);
        const int64_t env = row - (int64_t)a * E;
        out[k] = (int64_t)argmax_explore(q + k * nact, nact, ts->eps_thr[a], first_env + (uint32_t)env, turn, epoch << 4, a, seed_lo, seed_hi);
    }
}
__global__ void turn_set_kernel(TurnState* ts, const uint32_t epoch, const uint32_t turn) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { ts->epoch = epoch; ts->turn = turn; }
}
// Buffer.add: idx = (idx + 1) % capacity, once per agent sharing the ring; and the turn is over
__device__ __forceinline__ void turn_advance(TurnState* ts, const int A, const int lane) {
    if (lane < A && ts->cap[lane] > 0) ts->row[lane] = (ts->row[lane] + ts->step[lane]) % ts->cap[lane];
    if (lane == 0) ts->turn += 1u;
}
__global__ void turn_advance_kernel(TurnState* ts, const int A) {
    if (blockIdx.x == 0) turn_advance(ts, A, (int)threadIdx.x);
}
// The windows of the turn (the [E][A][N] tensor the policies read) into each agent's replay row of the turn in flight
// (Agent.add_memory -> Buffer.add, sorrel/agents/agent.py:127-130, sorrel/buffers.py:46-63), VEC elements per thread and step.
// Buffer.current_state (sorrel/buffers.py:143-154) by the device's row count: the `count` rows before the one the turn in flight
// fills, oldest first, wrapping around the ring -- out[j] = states[(row - count + j) mod capacity], each row_bytes long.
template <int VEC>
__global__ __launch_bounds__(kBlock) void turn_prev_rows_kernel(const TurnState* __restrict__ ts, const int a, const int count, uint8_t* __restrict__ out,
                                                                const int64_t row_bytes) {
    struct alignas(VEC) Pack { uint8_t v[VEC]; };
    const int64_t cap = ts->cap[a], row = ts->row[a];
    if (cap <= 0 || !ts->states[a]) return;
    const int64_t per = row_bytes / VEC, n = per * count;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t j = i / per, k = i - j * per;
        const int64_t src_row = (((row - count + j) % cap) + cap) % cap;
        reinterpret_cast<Pack*>(out + j * row_bytes)[k] = reinterpret_cast<const Pack*>(static_cast<const uint8_t*>(ts->states[a]) + src_row * row_bytes)[k];
    }
}

// (Letting the workgroup that finishes last advance the state -- one launch instead of two -- was measured: 4 096 same-address
// atomics cost more than the ~4 us of a dependent launch, 149 -> 328 us per recorded turn at 1 024 envs.)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void turn_commit_kernel(const TurnState* __restrict__ ts, const T* __restrict__ obs, const int64_t E, const int A, const int N) {
    // a wave per window: where it goes is wave-uniform (scalar loads of the state, no per-element division), its lanes copy
    // consecutive VEC-element pieces (a first version with a thread per piece and the state read per piece: 417 us for config 3's
    // 617 MB at 65 536 envs = 3 TB/s of traffic)
    struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };
    const int lane = threadIdx.x & 63;
    const int NV = N / VEC;
    const int64_t nwin = E * A, nwaves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t win = (int64_t)blockIdx.x * (kBlock / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); win < nwin; win += nwaves) {
        const int64_t e = win / A;
        const int a = (int)(win - e * A);
        if (ts->cap[a] <= 0 || !ts->states[a]) continue;
        const int64_t row = ts->row[a];
        if (lane == 0 && ts->dones[a]) ts->dones[a][row * E + e] = 0.f;
        const Pack* src = reinterpret_cast<const Pack*>(obs + win * N);
        Pack* dst = reinterpret_cast<Pack*>(static_cast<T*>(ts->states[a]) + (row * E + e) * ts->row_elems[a]);
        for (int k = lane; k < NV; k += 64) dst[k] = src[k];
    }
}

// Micro-benchmark (diagnostic only, not part of the product): write-only streams shaped like config 5's observations -- f32 [E][64][6][11][11],
// 185 856 B per env, a 512-thread workgroup per env -- to see what the emit pattern alone is worth:
//   mode 0  step_big's direct stores: wave w renders windows w, w + 8, ...; per window 6 planes x (64 + 57) lanes, one dword each
//   mode 1  the whole env as ONE burst by all 8 waves (16-byte streaming stores from staged bytes), after all the "work"
//   mode 2  two half-env bursts (32 agents each)
//   mode 3  per window a contiguous run of 8-byte streaming stores by its wave (the staged variant's shape)
//   mode 4  eight windows at a time (one per wave) staged, then the 23 232-byte run of the eight written by all 8 waves together
// `spin` = dependent VALU ops per window before its stores (stands in for the gather); lds = dynamic LDS bytes (sets workgroups per CU).
// Round 6 -- what varies is only WHICH ADDRESSES ARE OPEN AT ONCE:
//   `remap` (argv 6) = 1 for modes 0-4, 6: workgroup b takes env (b % 8) * (E / 8) + b / 8, so the workgroups that share an XCD (b, b + 8, ...:
//            round-robin dispatch) hold a CONTIGUOUS range of envs and the eight XCDs eight such ranges
//   mode 8   block per piece: `spin` = k pieces per env, block b writes the b-th run of ENV / k floats front to back with 512 threads (k = 1 is
//            mode 6; k = 4: "four cooperating workgroups per env"); remap 2: the k pieces of an env sit on ONE XCD (blocks 8 j + x, j < k)
//   mode 9   a persistent grid-stride fill: gridDim = lds bytes argument (e.g. 1024 / 2048 / 4096 blocks) x 256 threads, every thread walks the
//            whole buffer with a stride of the grid, `spin` float4 stores per iteration (1, 2 or 4; independent, 4 KB x grid apart)
//   mode 10  fill_ as torch launches it (profiles/r06_c5_emit_micro.txt has its grid from a kernel trace): 256-thread blocks, each thread `spin`
//            consecutive float4 (spin = 1, 2, 4: 16 / 32 / 64 bytes per lane), blocks consecutive
//   mode 11  step_big's direct pattern with every wave-wide store ALIGNED: wave w writes windows w, w + 8, ... as a flat run of 726 floats in
//            256-byte chunks that start on 256-byte boundaries (first and last chunk partial) -- same instruction count as mode 0, no 64-byte
//            sector written twice by two instructions
//   mode 12  ... as 16-byte stores whose lane 0 sits on a 128-byte line (the shape of step_big's staged windows), edges by single floats
// usage: c5_emit <mode> <envs> <spin> <lds bytes> [nontemporal 0/1] [remap 0/1/2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef float vfloat2 __attribute__((ext_vector_type(2)));
constexpr int A = 64, C = 6, VV = 121, N = C * VV, ENV = A * N;   // floats

__device__ __forceinline__ long env_of_block(const int remap, const int E) {
    const int b = blockIdx.x;
    return remap ? (long)(b & 7) * (E >> 3) + (b >> 3) : (long)b;       // (E a multiple of 8)
}
template <int MODE, bool NT>
__global__ __launch_bounds__(512) void emit(float* __restrict__ out, int E, int spin, int remap) {
    extern __shared__ uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long env = env_of_block(remap, E);
    if (env >= E) return;
    float* o = out + env * (long)ENV;
    int x = tid;
    auto work = [&](int n) { for (int i = 0; i < n; ++i) x = x * 1664525 + 1013904223; };
    auto st1 = [&](float* p, float v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
    if (MODE == 0) {
        for (int a = wv; a < A; a += 8) {
            work(spin);
            float* ob = o + a * N;
            const float v = (float)(x & 1);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                *(ob + c * VV + lane) = v;
                if (lane + 64 < VV) *(ob + c * VV + 64 + lane) = v;
            }
        }
    } else if (MODE == 11) {
        for (int a = wv; a < A; a += 8) {
            work(spin);
            const long e0 = (env * A + a) * (long)N;                  // first element of the window in the tensor
            const long first = e0 & ~63L;                              // ... and the 256-byte chunk it lies in
            const float v = (float)(x & 1);
            for (long c = first; c < e0 + N; c += 64) {
                const long e = c + lane;
                if (e >= e0 && e < e0 + N) st1(out + e, v);
            }
        }
    } else if (MODE == 12) {
        for (int a = wv; a < A; a += 8) {
            work(spin);
            const long e0 = (env * A + a) * (long)N, e1 = e0 + N;
            const long l0 = (e0 + 31) & ~31L;                          // first element on a 128-byte line
            const float v = (float)(x & 1);
            const vfloat4 v4 = {v, v, v, v};
            if (e0 + lane < l0) st1(out + e0 + lane, v);               // head: < 32 floats
            for (long c = l0 + 4 * lane; c + 4 <= e1; c += 256) {
                if (NT) __builtin_nontemporal_store(v4, reinterpret_cast<vfloat4*>(out + c)); else *reinterpret_cast<vfloat4*>(out + c) = v4;
            }
            const long t0 = l0 + ((e1 - l0) & ~3L);                    // tail: < 4 floats
            if (t0 + lane < e1) st1(out + t0 + lane, v);
        }
    } else if (MODE == 3) {
        for (int a = wv; a < A; a += 8) {
            work(spin);
            lds[wv * 768 + lane] = (uint8_t)x;
            __builtin_amdgcn_wave_barrier();
            float* ob = o + a * N;   // 8-byte aligned (N even)
            const uint16_t* s2 = reinterpret_cast<const uint16_t*>(lds + wv * 768);
            for (int i = lane; i < N / 2; i += 64) {
                const uint32_t b = s2[i & 255];
                vfloat2 v = {(float)(b & 0xFF), (float)(b >> 8)};
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vfloat2*>(ob) + i); else reinterpret_cast<vfloat2*>(ob)[i] = v;
            }
        }
    } else {
        const int groups = MODE == 1 ? 1 : (MODE == 2 ? 2 : 8);
        const int per = A / groups;                      // agents per burst
        for (int g = 0; g < groups; ++g) {
            for (int k = 0; k < per / 8; ++k) {          // every wave "renders" its windows of this group into LDS
                work(spin);
                lds[(wv * 2048 + lane + 64 * k) & 0x3FFF] = (uint8_t)x;
            }
            __syncthreads();
            const int n4 = per * N / 4;                  // float4 of the burst (per * 726 / 4: per even -> integer)
            vfloat4* o4 = reinterpret_cast<vfloat4*>(o + g * per * N);
            const uint32_t* l4 = reinterpret_cast<const uint32_t*>(lds);
            for (int i = tid; i < n4; i += 512) {
                const uint32_t b = l4[i & 0xFFF];
                vfloat4 v = {(float)(b & 0xFF), (float)((b >> 8) & 0xFF), (float)((b >> 16) & 0xFF), (float)(b >> 24)};
                if (NT) __builtin_nontemporal_store(v, o4 + i); else o4[i] = v;
            }
            __syncthreads();
        }
    }
    if (x == 0x7ffffff1) out[0] = 1.f;
}

// mode 5..8: no LDS, no work -- only WHO writes WHAT, to find what separates the patterns above from torch's fill_ (6.9 TB/s):
//   5  fill_'s shape: 256-thread blocks, each 16 KB contiguous (4 float4 per thread, 4 KB apart); `spin` = 1: 512-thread blocks, 32 KB each
//   6  a 512-thread block per env writes the env's 185 856 B front to back (mode 1 without staging and barriers)
//   7  like 6, but block b writes the b-th 185 856-byte slice in 8 pieces of 23 232 B with all OTHER blocks' pieces in between
//      (piece p of block b at ((p * gridDim + b) * 23 232 B): the chip's blocks advance through memory together, as with agent-major rows)
template <int MODE, bool NT>
__global__ __launch_bounds__(512) void plain(float* __restrict__ out, long total4, int spin, int remap, int E) {
    vfloat4* o4 = reinterpret_cast<vfloat4*>(out);
    const vfloat4 v = {1.f, 2.f, 3.f, (float)spin};
    auto st = [&](long i) { if (i < total4) { if (NT) __builtin_nontemporal_store(v, o4 + i); else o4[i] = v; } };
    if (MODE == 5) {
        const long base = (long)blockIdx.x * blockDim.x * 4;
        for (int k = 0; k < 4; ++k) st(base + k * blockDim.x + threadIdx.x);
    } else if (MODE == 6) {
        const long base = env_of_block(remap, E) * (ENV / 4);
        for (int i = threadIdx.x; i < ENV / 4; i += 512) st(base + i);
    } else if (MODE == 8) {
        const int k = spin, P4 = ENV / 4 / k;                 // float4 per piece
        long piece = blockIdx.x;
        if (remap == 1) { const long n = (long)E * k; piece = (long)(blockIdx.x & 7) * (n >> 3) + (blockIdx.x >> 3); }
        else if (remap == 2) {                                // the k pieces of env e on one XCD: block = 8 * (k * (e / 8) + j) + e % 8
            const int x = blockIdx.x & 7, q = blockIdx.x >> 3, j = q % k, e = (q / k) * 8 + x;
            piece = (long)e * k + j;
        }
        const long base = piece * P4;
        for (int i = threadIdx.x; i < P4; i += 512) st(base + i);
    } else if (MODE == 9) {
        const long stride = (long)gridDim.x * 256;
        long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (spin == 4) for (; i < total4; i += 4 * stride) { st(i); st(i + stride); st(i + 2 * stride); st(i + 3 * stride); }
        else if (spin == 2) for (; i < total4; i += 2 * stride) { st(i); st(i + stride); }
        else for (; i < total4; i += stride) st(i);
    } else if (MODE == 10) {
        const long base = ((long)blockIdx.x * 256 + threadIdx.x) * spin;
        for (int k = 0; k < spin; ++k) st(base + k);
    } else {
        constexpr int P4 = ENV / 4 / 8;      // float4 per piece (5 808)
        for (int p = 0; p < 8; ++p) {
            const long base = ((long)p * gridDim.x + blockIdx.x) * P4;
            for (int i = threadIdx.x; i < P4; i += 512) st(base + i);
        }
    }
}

template <int MODE, bool NT>
float run_plain(float* out, int E, int spin, int iters, int remap, int gridarg) {
    const long total4 = (long)E * ENV / 4;
    const int threads = MODE == 5 ? (spin ? 512 : 256) : ((MODE == 9 || MODE == 10) ? 256 : 512);
    long blocks = MODE == 5 ? (total4 + threads * 4 - 1) / (threads * 4) : E;
    if (MODE == 8) blocks = (long)E * spin;
    if (MODE == 9) blocks = gridarg > 0 ? gridarg : 2048;
    if (MODE == 10) blocks = (total4 + 256L * spin - 1) / (256L * spin);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((plain<MODE, NT>), dim3(blocks), dim3(threads), 0, 0, out, total4, spin, remap, E);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((plain<MODE, NT>), dim3(blocks), dim3(threads), 0, 0, out, total4, spin, remap, E);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1000.f;
}

template <int MODE, bool NT>
float run(float* out, int E, int spin, int ldsb, int iters, int remap) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&emit<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((emit<MODE, NT>), dim3(E), dim3(512), ldsb, 0, out, E, spin, remap);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((emit<MODE, NT>), dim3(E), dim3(512), ldsb, 0, out, E, spin, remap);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1000.f;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, E = argc > 2 ? atoi(argv[2]) : 2048, spin = argc > 3 ? atoi(argv[3]) : 0;
    const int ldsb = argc > 4 ? atoi(argv[4]) : 39936, nt = argc > 5 ? atoi(argv[5]) : 1, remap = argc > 6 ? atoi(argv[6]) : 0;
    const int iters = getenv("C5_ITERS") ? atoi(getenv("C5_ITERS")) : 200;
    float* out;
    CK(hipMalloc(&out, (size_t)E * ENV * 4 + 4096));
    float us = 0;
#define RUN(M) us = nt ? run<M, true>(out, E, spin, ldsb, iters, remap) : run<M, false>(out, E, spin, ldsb, iters, remap)
#define RUNP(M) us = nt ? run_plain<M, true>(out, E, spin, iters, remap, ldsb) : run_plain<M, false>(out, E, spin, iters, remap, ldsb)
    if (mode == 0) RUN(0); else if (mode == 1) RUN(1); else if (mode == 2) RUN(2); else if (mode == 3) RUN(3); else if (mode == 4) RUN(4);
    else if (mode == 11) RUN(11); else if (mode == 12) RUN(12);
    else if (mode == 5) RUNP(5); else if (mode == 6) RUNP(6); else if (mode == 7) RUNP(7); else if (mode == 8) RUNP(8); else if (mode == 9) RUNP(9); else RUNP(10);
    printf("mode %2d  envs %5d  spin %4d  lds %6d  nt %d  remap %d : %8.1f us  %.2f TB/s\n", mode, E, spin, ldsb, nt, remap, us, (double)E * ENV * 4 / us / 1e6);
    return 0;
}

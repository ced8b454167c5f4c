// Micro-benchmark: how fast can the observation tensor of BASELINE config 3 be written with different
// store shapes?  obs f32 [E][8][6][7][7] = 9408 B per env, one wave per env, 4 waves per workgroup
// (the step kernel's launch shape).  Diagnostic only; not part of the product.
//   v0: 48 dword stores of 49 lanes per env (what step_fast does: one store per (agent, channel))
//   v1: 588 float4 per env = 10 wave-wide 16-byte stores
//   v2: float2, 147 lanes per agent = 3 stores per agent
//   v3: v0 plus reading the 2 KiB grid first and writing it back (the whole traffic mix)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256, 8) void writer(float* __restrict__ obs, uint4* __restrict__ grid, int E, float val) {
    const int lane = threadIdx.x & 63;
    const long env = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env >= E) return;
    float* o = obs + env * 2352;
    uint4 g0, g1;
    if (V == 3) { g0 = grid[env * 128 + lane]; g1 = grid[env * 128 + 64 + lane]; val += (float)(g0.x & 1u); }
    if (V == 0 || V == 3) {
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (lane < 49) o[(a * 6 + c) * 49 + lane] = val + c;
    } else if (V == 1) {
        float4* o4 = reinterpret_cast<float4*>(o);
#pragma unroll
        for (int k = 0; k < 10; ++k)
            if (lane + 64 * k < 588) o4[lane + 64 * k] = make_float4(val, val + 1, val + 2, val + k);
    } else if (V == 2) {
        for (int a = 0; a < 8; ++a) {
            float2* o2 = reinterpret_cast<float2*>(o + a * 294);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (lane + 64 * k < 147) o2[lane + 64 * k] = make_float2(val, val + k);
        }
    }
    if (V == 3) { g0.y ^= 1u; grid[env * 128 + lane] = g0; grid[env * 128 + 64 + lane] = g1; }
}

template <int V>
float run(float* obs, uint4* grid, int E, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) writer<V><<<(E + 3) / 4, 256>>>(obs, grid, E, 1.0f);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) writer<V><<<(E + 3) / 4, 256>>>(obs, grid, E, (float)i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

int main() {
    const int E = 65536;
    float* obs; uint4* grid;
    CK(hipMalloc(&obs, (size_t)E * 9408));
    CK(hipMalloc(&grid, (size_t)E * 2048));
    CK(hipMemset(grid, 0, (size_t)E * 2048));
    const double mb = E * 9408.0 / 1e6;
    float t;
    t = run<0>(obs, grid, E, 100); printf("v0 dword x49 per (agent, channel): %.1f us  %.2f TB/s\n", t, mb / t / 1e0 * 1e-6 * 1e6 / 1e6);
    t = run<1>(obs, grid, E, 100); printf("v1 float4 x588 per env           : %.1f us  %.2f TB/s\n", t, mb / t);
    t = run<2>(obs, grid, E, 100); printf("v2 float2 x147 per agent         : %.1f us  %.2f TB/s\n", t, mb / t);
    t = run<3>(obs, grid, E, 100); printf("v3 v0 + grid read/write          : %.1f us  %.2f TB/s (obs+grid r/w)\n", t, (mb + E * 4096.0 / 1e6) / t);
    return 0;
}

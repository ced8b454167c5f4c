// Micro-benchmark: write-only streams shaped like Cleanup's observations (f32 [E][10][9][11][11] = 43 560 B per env; here
// 43 520 = 2 720 float4).  How does the achieved HBM write rate depend on WHO writes a region and WHEN?
//   mode 0: a wave per region (4 regions per workgroup), the region in `nb` bursts with `spin` dependent VALU ops between
//           bursts (the step kernel's shape: gather a few agents, emit them, gather the next)
//   mode 1: the 4 waves of a workgroup write the workgroup's 4 regions together, one region after the other
//   mode 2: linear (workgroup w writes bytes [w * 4R, (w + 1) * 4R) with its 256 threads striding 4 KiB rows): torch's fill shape
// lds = dynamic LDS bytes per workgroup (sets workgroups per CU).  Diagnostic only; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float vfloat4 __attribute__((ext_vector_type(4)));
constexpr int R4 = 2720;   // float4 per region

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void writer(vfloat4* __restrict__ out, long E, int nb, int spin, float val) {
    extern __shared__ int lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    vfloat4 v = {val, val + 1.f, val + 2.f, val + 3.f};
    auto st = [&](vfloat4* p) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
    if (MODE == 0) {
        const long env = (long)blockIdx.x * 4 + wave;
        if (env >= E) return;
        vfloat4* o = out + env * R4;
        const int per = (R4 + nb - 1) / nb;
        int x = lane;
        for (int b = 0; b < nb; ++b) {
            for (int i = 0; i < spin; ++i) x = x * 1664525 + 1013904223;
            if (x == 0x7fffffff) v.x += 1.f;
            const int hi = min(R4, (b + 1) * per);
            for (int i = b * per + lane; i < hi; i += 64) st(o + i);
        }
    } else if (MODE == 3) {   // mode 0 with the data coming from LDS bytes (ds_read_b32 + 4 converts per float4) and a region that starts
                              // 4 * (spin & 3) bytes off a 16-byte boundary (edge elements as single dword stores), like the step kernel's chunks
        const long env = (long)blockIdx.x * 4 + wave;
        if (env >= E) return;
        float* o = reinterpret_cast<float*>(out + env * R4) + (spin & 3);
        const uint32_t* l4 = reinterpret_cast<const uint32_t*>(lds) + wave * 1024;
        const int per = (R4 + nb - 1) / nb;
        for (int b = 0; b < nb; ++b) {
            const int sh = (4 - (spin & 3)) & 3;     // elements until the first aligned float4
            const int n = (min(R4, (b + 1) * per) - b * per) * 4 - 4;   // elements of this burst
            float* ob = o + b * per * 4;
            const int nq = (n - sh) >> 2;
            vfloat4* q = reinterpret_cast<vfloat4*>(ob + sh);
            const int mis = (spin & 4) ? (int)((reinterpret_cast<uintptr_t>(q) >> 4) & 7) : 0;   // spin & 4: lane 0 of every store sits on a 128-byte line
            for (int i = lane - mis; i < nq; i += 64) {
                if (i < 0) continue;
                const uint32_t w = l4[i & 1023];
                v.x = (float)(w & 0xFFu); v.y = (float)((w >> 8) & 0xFFu); v.z = (float)((w >> 16) & 0xFFu); v.w = (float)(w >> 24);
                st(q + i);
            }
            if (lane < sh) ob[lane] = val;
            if (lane < n - sh - 4 * nq) ob[sh + 4 * nq + lane] = val;
            __builtin_amdgcn_wave_barrier();
        }
    } else if (MODE == 1) {
        const long env0 = (long)blockIdx.x * 4;
        int x = lane;
        for (int b = 0; b < nb; ++b) {
            for (int i = 0; i < spin; ++i) x = x * 1664525 + 1013904223;
            if (x == 0x7fffffff) v.x += 1.f;
            __syncthreads();
            const int per = (R4 + nb - 1) / nb;
            const int hi = min(R4, (b + 1) * per);
            for (int e = 0; e < 4 && env0 + e < E; ++e) {
                vfloat4* o = out + (env0 + e) * R4;
                for (int i = b * per + threadIdx.x; i < hi; i += 256) st(o + i);
            }
        }
    } else {
        const long base = (long)blockIdx.x * 4 * R4;
        for (int i = threadIdx.x; i < 4 * R4; i += 256)
            if (base + i < E * R4) st(out + base + i);
    }
    if (lds[threadIdx.x & 1] == 0x12345) out[0] = v;
}

template <int MODE, bool NT>
float run(vfloat4* out, long E, int nb, int spin, int ldsb, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)writer<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int i = 0; i < 3; ++i) writer<MODE, NT><<<(E + 3) / 4, 256, ldsb>>>(out, E, nb, spin, 1.0f);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) writer<MODE, NT><<<(E + 3) / 4, 256, ldsb>>>(out, E, nb, spin, (float)i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    const long E = argc > 1 ? atol(argv[1]) : 65536;
    vfloat4* out;
    CK(hipMalloc(&out, (size_t)E * R4 * 16));
    const double mb = E * R4 * 16.0 / 1e6;
    printf("E = %ld, %.0f MB written per launch\n", E, mb);
    for (int nb : {1, 4, 10})        // 1. who writes a region, in how many bursts, with how much dependent work between the bursts (five workgroups per CU)
        for (int spin : {0, 400}) {
            float t0 = run<0, true>(out, E, nb, spin, 32768, 20), t0p = run<0, false>(out, E, nb, spin, 32768, 20);
            float t1 = run<1, true>(out, E, nb, spin, 32768, 20), t1p = run<1, false>(out, E, nb, spin, 32768, 20);
            printf("bursts %2d  spin %3d | wave-per-region nt %.1f us %.2f TB/s  plain %.1f us %.2f TB/s | workgroup-per-region nt %.1f us %.2f TB/s  plain %.1f us %.2f TB/s\n",
                   nb, spin, t0, mb / t0, t0p, mb / t0p, t1, mb / t1, t1p, mb / t1p);
        }
    for (int nb : {1, 4})            // 2. data from LDS bytes; a region 4 / 8 bytes off a 16-byte boundary, without (1, 2) and with (5, 6) lane 0 on a 128-byte line
        for (int sp : {0, 1, 5, 2, 6}) {
            float t3 = run<3, true>(out, E, nb, sp, 32768, 20), t3p = run<3, false>(out, E, nb, sp, 32768, 20);
            printf("bursts %2d  misalign %d | wave-per-region from LDS bytes: nt %.1f us %.2f TB/s  plain %.1f us %.2f TB/s\n", nb, sp, t3, mb / t3, t3p, mb / t3p);
        }
    for (int off4 : {0, 1, 2, 4})   // 3. the whole pattern shifted by 16 / 32 / 64 bytes: every wave-wide store then starts inside a 128-byte line
    {
        float t0 = run<0, true>(out + off4, E - 1, 1, 0, 32768, 20), t0p = run<0, false>(out + off4, E - 1, 1, 0, 32768, 20);
        printf("wave-per-region shifted by %2d bytes: nt %.1f us %.2f TB/s  plain %.1f us %.2f TB/s\n", off4 * 16, t0, mb / t0, t0p, mb / t0p);
    }
    // 4. torch's fill shape
    float t2 = run<2, true>(out, E, 1, 0, 1024, 20), t2p = run<2, false>(out, E, 1, 0, 1024, 20);
    printf("linear: nt %.1f us %.2f TB/s  plain %.1f us %.2f TB/s\n", t2, mb / t2, t2p, mb / t2p);
    return 0;
}

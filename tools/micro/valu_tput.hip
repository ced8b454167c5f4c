// Micro-benchmark: VALU instruction throughput of one SIMD as a function of the waves resident on it.
// Diagnostic only.  grid = 256 CUs x (waves per SIMD); each wave runs 8192 x 8 independent instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WHICH>
__global__ __launch_bounds__(256) void k(unsigned* sink, int iters) {
    unsigned a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    const unsigned m = 0xD2511F53u;
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) {
            asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                         "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 1) {
            asm volatile("v_mad_u64_u32 %0, s[20:21], %4, %5, 0\n v_mad_u64_u32 %1, s[20:21], %4, %5, 0\n v_mad_u64_u32 %2, s[20:21], %4, %5, 0\n v_mad_u64_u32 %3, s[20:21], %4, %5, 0\n"
                         "v_mad_u64_u32 %0, s[20:21], %4, %5, 0\n v_mad_u64_u32 %1, s[20:21], %4, %5, 0\n v_mad_u64_u32 %2, s[20:21], %4, %5, 0\n v_mad_u64_u32 %3, s[20:21], %4, %5, 0\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a0), "s"(m) : "s20", "s21");
        } else if (WHICH == 2) {
            asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_hi_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 3) {
            asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n"
                         "v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s23, %7, 3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m) : "s20", "s21", "s22", "s23");
        } else if (WHICH == 4) {   // dependent chain of xors (latency)
            asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %8\n"
                         "v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %8\n v_xor_b32 %0, %0, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 6) {   // 8 VALU + 8 SALU interleaved: do the two pipes issue in parallel (across waves)?
            asm volatile("v_xor_b32 %0, %0, %8\n s_xor_b32 s20, s20, %8\n v_xor_b32 %1, %1, %8\n s_xor_b32 s21, s21, %8\n v_xor_b32 %2, %2, %8\n s_xor_b32 s22, s22, %8\n v_xor_b32 %3, %3, %8\n s_xor_b32 s23, s23, %8\n"
                         "v_xor_b32 %4, %4, %8\n s_xor_b32 s20, s20, %8\n v_xor_b32 %5, %5, %8\n s_xor_b32 s21, s21, %8\n v_xor_b32 %6, %6, %8\n s_xor_b32 s22, s22, %8\n v_xor_b32 %7, %7, %8\n s_xor_b32 s23, s23, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m) : "s20", "s21", "s22", "s23", "scc");
        } else if (WHICH == 7) {   // 8 VALU + 8 LDS reads interleaved
            asm volatile("v_xor_b32 %0, %0, %8\n ds_read_b32 %4, %7\n v_xor_b32 %1, %1, %8\n ds_read_b32 %5, %7\n v_xor_b32 %2, %2, %8\n ds_read_b32 %6, %7\n v_xor_b32 %3, %3, %8\n ds_read_b32 %4, %7\n"
                         "v_xor_b32 %0, %0, %8\n ds_read_b32 %5, %7\n v_xor_b32 %1, %1, %8\n ds_read_b32 %6, %7\n v_xor_b32 %2, %2, %8\n ds_read_b32 %4, %7\n v_xor_b32 %3, %3, %8\n ds_read_b32 %5, %7\n s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : "v"((threadIdx.x & 63) * 4), "s"(m));
        } else if (WHICH == 5) {   // SALU
            asm volatile("s_xor_b32 s20, s20, %8\n s_xor_b32 s21, s21, %8\n s_xor_b32 s22, s22, %8\n s_xor_b32 s23, s23, %8\n"
                         "s_xor_b32 s20, s20, %8\n s_xor_b32 s21, s21, %8\n s_xor_b32 s22, s22, %8\n s_xor_b32 s23, s23, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m) : "s20", "s21", "s22", "s23", "scc");
        }
    }
    sink[threadIdx.x & 255] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(b0 ^ b1 ^ b2 ^ b3);
}

template <int WHICH>
void run(const char* name, unsigned* sink) {
    const int iters = 4096;
    printf("%-16s", name);
    for (int wps : {1, 2, 4, 8}) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        k<WHICH><<<256 * wps, 256>>>(sink, 64);
        CK(hipEventRecord(a));
        k<WHICH><<<256 * wps, 256>>>(sink, iters);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double cyc = ms * 1e-3 * 2.4e9;                 // at the nominal 2.4 GHz
        printf("  %dw/SIMD: %.2f cyc/instr/SIMD", wps, cyc / (iters * 8.0 * wps));
    }
    printf("\n");
}

int main() {
    unsigned* sink;
    CK(hipMalloc(&sink, 256 * 4));
    run<0>("v_xor_b32", sink);
    run<1>("v_mad_u64_u32", sink);
    run<2>("v_mul_hi/lo_u32", sink);
    run<3>("v_readlane_b32", sink);
    run<4>("v_xor dependent", sink);
    run<5>("s_xor_b32", sink);
    run<6>("8 VALU + 8 SALU", sink);
    run<7>("8 VALU + 8 LDS", sink);
    return 0;
}

// Micro-benchmark: issue cost (cycles per instruction, one wave alone on a SIMD) of the integer
// instructions Philox-4x32 can be built from on gfx950.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(x) x x x x x x x x
template <int WHICH>
__global__ void k(unsigned long long* out, unsigned* sink) {
    unsigned a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
    const unsigned m = 0xD2511F53u;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < 1024; ++i) {
        if (WHICH == 0) {   // 8 independent v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 1) {   // v_mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                         "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 2) {   // v_mad_u64_u32 (64-bit result)
            asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, 0\n v_mad_u64_u32 %1, s[20:21], %8, %9, 0\n v_mad_u64_u32 %2, s[20:21], %8, %9, 0\n v_mad_u64_u32 %3, s[20:21], %8, %9, 0\n"
                         "v_mad_u64_u32 %4, s[20:21], %8, %9, 0\n v_mad_u64_u32 %5, s[20:21], %8, %9, 0\n v_mad_u64_u32 %6, s[20:21], %8, %9, 0\n v_mad_u64_u32 %7, s[20:21], %8, %9, 0\n"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(a0), "s"(m) : "s20", "s21");
        } else if (WHICH == 3) {   // v_xor (reference: plain 32-bit VALU)
            asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                         "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 4) {   // v_mul_u32_u24
            asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n"
                         "v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 5) {   // v_bitop3_b32
            asm volatile("v_bitop3_b32 %0, %0, %1, %8 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %8 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %8 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %8 bitop3:0x96\n"
                         "v_bitop3_b32 %4, %4, %5, %8 bitop3:0x96\n v_bitop3_b32 %5, %5, %6, %8 bitop3:0x96\n v_bitop3_b32 %6, %6, %7, %8 bitop3:0x96\n v_bitop3_b32 %7, %7, %0, %8 bitop3:0x96\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 6) {   // v_mad_u32_u24
            asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n"
                         "v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        } else if (WHICH == 7) {   // v_readlane_b32 (to an SGPR)
            asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 3\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 3\n"
                         "v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %6, 3\n v_readlane_b32 s23, %7, 3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m) : "s20", "s21", "s22", "s23");
        } else if (WHICH == 8) {   // v_cvt_f32_ubyte0
            asm volatile("v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %1, %2\n v_cvt_f32_ubyte2 %2, %3\n v_cvt_f32_ubyte3 %3, %4\n"
                         "v_cvt_f32_ubyte0 %4, %5\n v_cvt_f32_ubyte1 %5, %6\n v_cvt_f32_ubyte2 %6, %7\n v_cvt_f32_ubyte3 %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[WHICH] = t1 - t0;
    sink[threadIdx.x & 255] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7);
}

int main() {
    unsigned long long* out; unsigned* sink;
    CK(hipMalloc(&out, 16 * 8)); CK(hipMalloc(&sink, 256 * 4));
    const int T = 1024;   // 16 waves = 4 per SIMD: the SIMD's VALU is saturated, elapsed / (4 waves x instructions) = issue cost
    k<0><<<1, T>>>(out, sink); k<1><<<1, T>>>(out, sink); k<2><<<1, T>>>(out, sink); k<3><<<1, T>>>(out, sink);
    k<4><<<1, T>>>(out, sink); k<5><<<1, T>>>(out, sink); k<6><<<1, T>>>(out, sink); k<7><<<1, T>>>(out, sink); k<8><<<1, T>>>(out, sink);
    CK(hipDeviceSynchronize());
    unsigned long long h[16];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    const char* names[] = {"v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_xor_b32", "v_mul_u32_u24", "v_bitop3_b32", "v_mad_u32_u24", "v_readlane_b32", "v_cvt_f32_ubyteN"};
    for (int i = 0; i < 9; ++i) printf("%-18s %.2f cycles per instruction per SIMD (4 waves per SIMD)\n", names[i], (double)h[i] / (1024.0 * 8 * 4));
    return 0;
}

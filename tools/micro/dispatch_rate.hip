// Micro-benchmark: how long does the dispatcher take to push E one-wave-per-env workgroups through the chip
// when each wave does (almost) nothing?  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SPIN>
__global__ void tiny(int* out, int E) {
    extern __shared__ int lds[];
    const long env = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (env >= E) return;
    int v = threadIdx.x;
    for (int i = 0; i < SPIN; ++i) v = v * 1664525 + 1013904223;   // SPIN dependent VALU ops
    lds[threadIdx.x] = v;
    if (v == 0x12345678) out[0] = lds[(threadIdx.x + 1) & 63];
}

template <int SPIN>
float run(int* out, int E, int block, int lds_per_wave, int iters) {
    const int wpb = block / 64;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) tiny<SPIN><<<(E + wpb - 1) / wpb, block, lds_per_wave * wpb>>>(out, E);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) tiny<SPIN><<<(E + wpb - 1) / wpb, block, lds_per_wave * wpb>>>(out, E);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    int* out;
    CK(hipMalloc(&out, 64));
    const int E = argc > 1 ? atoi(argv[1]) : 65536;      // (4 096: the launch floor next to BASELINE config 2)
    printf("empty waves, E = %d\n", E);
    for (int block : {64, 128, 256, 512, 1024})
        printf("  block %4d  lds/wave 2304: %.1f us   lds/wave 0: %.1f us\n", block, run<0>(out, E, block, 2304, 50), run<0>(out, E, block, 256, 50));
    printf("waves with 1000 dependent VALU ops (4000+ cycles alone)\n");
    for (int block : {64, 256})
        printf("  block %4d  lds/wave 2304: %.1f us\n", block, run<1000>(out, E, block, 2304, 50));
    return 0;
}

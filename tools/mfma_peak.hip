// Dev microbenchmark: sustained fp32 MFMA rate (v_mfma_f32_32x32x2_f32) with no memory traffic, to calibrate
// what "100 %" means for the convolution kernels on this chip under its power/clock management.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// RANDOM = 1: operands are full-entropy pseudo-random floats that change every step (realistic switching activity,
// hence realistic clocks under the power limit) -- but generated IN the loop by ~10 VALU instructions per 4 MFMAs, and fp32 VALU
// time adds to fp32 MFMA time (tools/mfma_valu_overlap.hip): this mode under-reports the MFMA ceiling;
// RANDOM = 2: eight full-entropy operand pairs per lane prepared BEFORE the loop and rotated through it -- random data, no VALU
// in the loop: the sustained fp32 MFMA ceiling on realistic operands;  RANDOM = 0: a handful of constant values (best case).
template <int NACC, int RANDOM = 0>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed * (threadIdx.x % 7 + 1) * 0.01f, b = seed * (threadIdx.x % 5 + 1) * 0.013f;
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float ra[8], rb[8];
    if (RANDOM == 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            st = st * 1664525u + 1013904223u;
            ra[q] = __uint_as_float(0x3f000000u | (st >> 9)) - 0.75f;
            rb[q] = __uint_as_float(0x3f000000u | ((st * 2246822519u) >> 9)) - 0.75f;
        }
    }
    for (int it = 0; it < iters; ++it) {
        if (RANDOM == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[q], rb[(q + i) & 7], acc[i], 0, 0, 0);
            continue;
        }
        if (RANDOM == 1) {
            st = st * 1664525u + 1013904223u;
            a = __uint_as_float(0x3f000000u | (st >> 9)) - 0.75f;          // uniform in [-0.25, 0.25)
            b = __uint_as_float(0x3f000000u | ((st * 2246822519u) >> 9)) - 0.75f;
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        if (!RANDOM) a = -a;   // keep values bounded and data-dependent
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int RANDOM = 0>
void run(int blocks, int iters) {
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NACC, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NACC, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
    if (RANDOM == 2) flops *= 8.0;
    printf("%s NACC=%d blocks=%d (%.1f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", RANDOM == 2 ? "random-data, operands prepared outside the loop" : RANDOM ? "random-data (RNG in the loop)" : "constant-data", NACC, blocks, blocks * 4 / 1024.0, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(256, 20000); run<2>(256, 10000); run<4>(256, 5000); run<4>(512, 5000); run<4>(1024, 2500); run<1>(2048, 5000);
    run<4, 1>(512, 5000); run<4, 1>(1024, 2500); run<4, 1>(512, 20000); run<4>(512, 20000);
    run<4, 2>(256, 2500); run<4, 2>(512, 2500); run<4, 2>(1024, 600); run<4, 2>(512, 2500);
    return 0;
}

// Dev microbenchmark: sustained rate of v_mfma_f32_32x32x16_bf16 with no memory traffic, in the shape the bf16x3 mesh kernel issues it
// (three accumulators taking turns, six piece products per 16 k), on constant and on full-entropy operands prepared outside the loop --
// what "100 %" of the bf16 matrix pipe means on this chip under its power / clock management (csrc/mesh_split.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    bf16x8 ra[3], rb[3][3];
    auto mk = [&]() {
        uint4 u;
        unsigned* p = reinterpret_cast<unsigned*>(&u);
        for (int q = 0; q < 4; ++q) {
            st = st * 1664525u + 1013904223u;
            // two bf16 in [-0.5, 0.5): random mantissas, small exponents (RANDOM) or one constant pair
            p[q] = RANDOM ? (((st >> 8) & 0x807F807Fu) | 0x3E003E00u) : 0x3E803E80u;
        }
        return __builtin_bit_cast(bf16x8, u);
    };
    for (int s = 0; s < 3; ++s) { ra[s] = mk(); for (int t = 0; t < 3; ++t) rb[t][s] = mk(); }
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[PA[p]], rb[t % 3][PB[p]], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int RANDOM>
void run(int blocks, int iters) {
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NACC, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NACC, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double n_mfma = (double)blocks * 4 * iters * 6 * NACC;
    const double flops = n_mfma * 2.0 * 32 * 32 * 16;
    const double waves_per_simd = blocks * 4 / 1024.0;
    printf("%s NACC=%d blocks=%d (%.1f waves/SIMD): %.3f ms  %.0f TFLOP/s  %.1f ns per MFMA and SIMD (32 cycles at 2.4 GHz = 13.3 ns)\n",
           RANDOM ? "random-data  " : "constant-data", NACC, blocks, waves_per_simd, ms, flops / ms / 1e9,
           ms * 1e6 / (n_mfma / 1024.0));
    hipFree(out);
}
int main() {
    run<3, 0>(256, 4000); run<3, 0>(1024, 1000); run<3, 1>(256, 4000); run<3, 1>(512, 2000); run<3, 1>(1024, 1000); run<3, 1>(1024, 200);
    run<3, 1>(1024, 50); run<1, 1>(1024, 1000); run<6, 1>(512, 1000);
    return 0;
}

// Dev microbenchmark: does fp32 VALU work overlap with v_mfma_f32_32x32x2_f32 on gfx950?
//   same-wave : each wave issues 4 MFMAs (independent accumulators) and NV independent v_fma_f32 per iteration
//   cross-wave: 512-thread workgroups, waves 0-3 issue only the MFMAs, waves 4-7 only the VALU work (one of each per SIMD)
// If the two overlap, time(MFMA + VALU) ~ max(time(MFMA), time(VALU)); if they share the datapath, it is the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV, int SPLIT>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1);
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const int wave = threadIdx.x >> 6;
    const bool do_m = !SPLIT || wave < 4, do_v = !SPLIT || wave >= 4;
    for (int it = 0; it < iters; ++it) {
        st = st * 1664525u + 1013904223u;
        const float a = __uint_as_float(0x3f000000u | (st >> 9)) - 0.75f;
        const float b = __uint_as_float(0x3f000000u | ((st * 2246822519u) >> 9)) - 0.75f;
        if (do_m) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 15] = __builtin_fmaf(v[i & 15], a, b);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NM, int NV, int SPLIT>
void run(int threads, const char* what) {
    const int blocks = 256, iters = 4000;
    float* out; hipMalloc(&out, blocks * 512 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NM, NV, SPLIT>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NM, NV, SPLIT>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-34s threads=%d NM=%d NV=%3d : %.3f ms  = %.0f ns/iter\n", what, threads, NM, NV, ms, ms * 1e6 / iters);
    hipFree(out);
}
int main() {
    run<4, 0, 0>(256, "MFMA only, 1 wave/SIMD");
    run<0, 32, 0>(256, "VALU only, 1 wave/SIMD");
    run<0, 64, 0>(256, "VALU only, 1 wave/SIMD");
    run<0, 128, 0>(256, "VALU only, 1 wave/SIMD");
    run<4, 32, 0>(256, "same wave, 1 wave/SIMD");
    run<4, 64, 0>(256, "same wave, 1 wave/SIMD");
    run<4, 128, 0>(256, "same wave, 1 wave/SIMD");
    run<4, 0, 0>(512, "MFMA only, 2 waves/SIMD");
    run<4, 64, 0>(512, "same wave, 2 waves/SIMD");
    run<4, 128, 0>(512, "same wave, 2 waves/SIMD");
    run<4, 0, 1>(512, "cross-wave: MFMA waves only");
    run<0, 64, 1>(512, "cross-wave: VALU waves only");
    run<0, 128, 1>(512, "cross-wave: VALU waves only");
    run<4, 64, 1>(512, "cross-wave: MFMA || VALU");
    run<4, 128, 1>(512, "cross-wave: MFMA || VALU");
    run<8, 128, 1>(512, "cross-wave: MFMA || VALU");
    return 0;
}

// Dev microbenchmark: sustained fp32 MFMA rate (v_mfma_f32_32x32x2_f32) with no memory traffic, to calibrate
// what "100 %" means for the convolution kernels on this chip under its power/clock management.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed * (threadIdx.x % 7 + 1) * 0.01f, b = seed * (threadIdx.x % 5 + 1) * 0.013f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a = -a;   // keep values bounded and data-dependent
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters) {
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
    printf("NACC=%d blocks=%d (%.1f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", NACC, blocks, blocks * 4 / 1024.0, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(256, 20000); run<2>(256, 10000); run<4>(256, 5000); run<4>(512, 5000); run<4>(1024, 2500); run<1>(2048, 5000);
    return 0;
}

// One wave per SIMD (1 024 waves): cycles per v_fma_f32 when every instruction depends on the previous one (1 chain) and when 2, 4 or
// 8 independent chains are interleaved.  What DESIGN.md's "a wave alone on its SIMD issues dependent VALU instructions every ~5.7
// cycles, independent ones every 4" rests on.      hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_dep_probe tools/valu_dep_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CHAINS>
__global__ __launch_bounds__(64) void k(float* out, float a, float b, int iters, long long* cyc) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / CHAINS; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
static void run(float* out, long long* cyc, long long* h, int blocks = 1024) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)h[i]; m /= blocks;
    printf("%d chain(s), %d waves per SIMD: %.2f ns per v_fma_f32 and SIMD (kernel %.3f ms, %d instructions per wave); clock ticks per instruction of a wave %.3f\n",
           CHAINS, blocks / 1024, ms * 1e6 / ((double)iters * 64 * (blocks / 1024)), ms, iters * 64, m / ((double)iters * 64));
}

int main() {
    float* out; long long* cyc; hipMalloc(&out, 4096 * 64 * 4); hipMalloc(&cyc, 4096 * 8);
    static long long h[4096];
    run<1>(out, cyc, h); run<2>(out, cyc, h); run<4>(out, cyc, h); run<8>(out, cyc, h);
    run<1>(out, cyc, h, 2048); run<8>(out, cyc, h, 2048); run<1>(out, cyc, h, 4096); run<8>(out, cyc, h, 4096);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device clock rate attribute: %d kHz\n", clk);
    return 0;
}

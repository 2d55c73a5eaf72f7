// Dev microbenchmark 2: placement of the LDS reads relative to the MFMAs (no register copies: two fragment sets, loop unrolled by two).
//   MODE 0: all reads of the NEXT group in a burst, then the group's MFMAs (what the product loops do, sched_barrier between)
//   MODE 1: one read after every (NM / NL)-th MFMA (interleaved by hand with sched_group_barrier)
//   MODE 2: burst, but the wave raises its priority for the MFMA part (s_setprio 1) and drops it for the reads
// 16 MFMAs (4 accumulators) + NL ds_read_b128 per group, as in the direct convolution kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NL>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.001f * (i % 97) - 0.05f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float4* base = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
    float4 fa[NL], fb[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) { fa[l] = base[l * 64]; fb[l] = base[(l * 64 + 256) & 2047]; }
    int off = 0;
    auto mfmas = [&](const float4 (&f)[NL]) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float4 v = f[m % NL];
            const float a = (m & 1) ? v.y : v.x, b = (m & 2) ? v.w : v.z;
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
        }
    };
    for (int it = 0; it < iters; ++it) {
        // group A: reads into fb while MFMAs consume fa; group B: the other way round
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 (&dst)[NL] = half ? fa : fb;
            const float4 (&src)[NL] = half ? fb : fa;
            off = (off + 64) & 1023;
            if (MODE == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int l = 0; l < NL; ++l) dst[l] = base[(l * 64 + off) & 2047];
            if (MODE != 1) __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2) __builtin_amdgcn_s_setprio(1);
            mfmas(src);
            if (MODE == 1) {
                // ask the scheduler for: 16 / NL MFMAs, then one DS read, repeated
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 16 / NL, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // DS read
                }
            } else __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + fa[0].x + fb[0].y;
}

template <int MODE, int NL>
void run(int wg_per_cu, const char* what) {
    const int blocks = 256 * wg_per_cu, iters = 2000;
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, NL>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<MODE, NL>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
    printf("%-40s %d WG/CU  16 MFMA + %d ds_read_b128 : %.3f ms  %.1f TFLOP/s\n", what, wg_per_cu, NL, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int wg : {1, 2, 4}) {
        run<0, 4>(wg, "burst then MFMAs");
        run<1, 4>(wg, "reads interleaved with the MFMAs");
        run<2, 4>(wg, "burst, priority raised for the MFMAs");
        run<0, 2>(wg, "burst then MFMAs");
        run<0, 8>(wg, "burst then MFMAs");
        run<1, 8>(wg, "reads interleaved with the MFMAs");
    }
    return 0;
}

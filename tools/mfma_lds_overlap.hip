// Dev microbenchmark: what does the rest of a GEMM K loop cost beside v_mfma_f32_32x32x2_f32?  (mfma_peak: 153.7 TF/s with
// nothing else in the loop; the product K loops reach 135-139.)  Per iteration a wave issues NM MFMAs (4 accumulators, operands
// taken from the LDS reads of the PREVIOUS iteration so that the reads are independent of this iteration's MFMAs) and
//   NL  LDS reads of width W (1 = b32, 2 = b64, 4 = b128), conflict free;
//   BAR s_barrier per iteration (workgroup = 4 waves, one per SIMD);
//   SAL scalar instructions (address bookkeeping);
// for 1, 2 and 4 workgroups per CU.  Reported: time per iteration and the MFMA rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NL, int W, int BAR, int SAL>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.001f * (i % 97) - 0.05f;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float f[16];
    for (int i = 0; i < 16; ++i) f[i] = 0.01f * (i + 1);
    const float* base = lds + (threadIdx.x & 63) * W;
    int soff = 0;
    for (int it = 0; it < iters; ++it) {
        float g[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) g[i] = f[i];
        // reads for the next iteration (independent of this iteration's MFMAs)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float* p = base + ((l * 64 * W + soff) & 4095);
            if (W == 1) f[l & 15] = p[0];
            else if (W == 2) { const float2 v = *reinterpret_cast<const float2*>(p); f[(2 * l) & 15] = v.x; f[(2 * l + 1) & 15] = v.y; }
            else { const float4 v = *reinterpret_cast<const float4*>(p); f[(4 * l) & 15] = v.x; f[(4 * l + 1) & 15] = v.y; f[(4 * l + 2) & 15] = v.z; f[(4 * l + 3) & 15] = v.w; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[m & 15], g[(m + 5) & 15], acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < SAL; ++s) asm volatile("s_add_u32 %0, %0, 64" : "+s"(soff));
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + f[0];
}

template <int NM, int NL, int W, int BAR, int SAL>
void run(int wg_per_cu, const char* what) {
    const int blocks = 256 * wg_per_cu, iters = 3000;
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NM, NL, W, BAR, SAL>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NM, NL, W, BAR, SAL>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)blocks * 4 * iters * NM * 2.0 * 32 * 32 * 2;
    printf("%-44s %d WG/CU  NM=%2d NL=%2d W=%d BAR=%d SAL=%2d : %.3f ms  %.1f TFLOP/s\n", what, wg_per_cu, NM, NL, W, BAR, SAL, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int wg : {1, 2, 4}) {
        run<24, 0, 1, 0, 0>(wg, "MFMA only (registers)");
        run<24, 32, 1, 0, 0>(wg, "+ 32 ds_read_b32 (the mesh kernel's chunk)");
        run<24, 8, 4, 0, 0>(wg, "+ 8 ds_read_b128 (same bytes)");
        run<24, 32, 1, 1, 0>(wg, "+ 32 ds_read_b32 + barrier");
        run<24, 32, 1, 1, 12>(wg, "+ 32 ds_read_b32 + barrier + 12 SALU");
        run<16, 4, 4, 0, 0>(wg, "16 MFMA + 4 ds_read_b128 (conv group)");
        run<16, 4, 4, 1, 0>(wg, "16 MFMA + 4 ds_read_b128 + barrier");
    }
    return 0;
}

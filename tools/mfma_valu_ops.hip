// Dev microbenchmark: WHICH VALU instructions cost MFMA time beside v_mfma_f32_32x32x2_f32 on gfx950?
// Per iteration a wave issues 4 MFMAs (one accumulator each) and/or NV independent VALU instructions of one kind:
//   same-wave : both in one wave (1 wave per SIMD);   cross-wave: waves 0-3 only MFMAs, waves 4-7 only the VALU work.
// Reported: ns per iteration alone and together -- "sum" means the instruction kind runs on the MFMA datapath, "max" that it overlaps.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__device__ __forceinline__ void valu(v2f& x, v2f a, v2f b) {
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x.x) : "v"(a.x), "v"(b.x));
    if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.x) : "v"(a.x));
    if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x.x) : "v"(a.x));
    if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x.x) : "v"(a.x));
    if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(x.x) : "v"(a.x));
    if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x.x) : "v"(a.x));
}
// the same with three DIFFERENT register operands per instruction (x = y op z [+ x]): operand fetch from the register file
template <int OP>
__device__ __forceinline__ void valu3(v2f& x, v2f y, v2f z) {
    if (OP == 10) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x.x) : "v"(y.x), "v"(z.x));
    if (OP == 11) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x.x) : "v"(y.x), "v"(z.x));
    if (OP == 13) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
    if (OP == 14) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
    if (OP == 15) asm volatile("v_pk_fma_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(x) : "v"(y), "v"(z));       // y + z as a packed FMA
    if (OP == 17) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
}

template <int NM, int NV, int SPLIT, int OP>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    v2f v[16];
    for (int i = 0; i < 16; ++i) v[i] = (v2f){seed * (i + 1), seed * (i + 2)};
    const int wave = threadIdx.x >> 6;
    const bool do_m = !SPLIT || wave < 4, do_v = !SPLIT || wave >= 4;
    const float a = seed * 0.999f + 1e-3f * (threadIdx.x & 3), b = 1.0f - seed * 1e-3f;
    const v2f a2 = {a, a * 0.5f}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (OP < 10) valu<OP>(v[i & 15], a2, b2);
                else valu3<OP>(v[i & 15], v[(i + 5) & 15], v[(i + 11) & 15]);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NM, int NV, int SPLIT, int OP>
float run(int threads) {
    const int blocks = 256, iters = 4000;
    float* out; hipMalloc(&out, blocks * 512 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NM, NV, SPLIT, OP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NM, NV, SPLIT, OP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    hipFree(out);
    return ms * 1e6f / iters;
}
template <int OP>
void kind(const char* name) {
    const float m1 = run<4, 0, 0, OP>(256), v1 = run<0, 32, 0, OP>(256), b1 = run<4, 32, 0, OP>(256);
    const float mx = run<4, 0, 1, OP>(512), vx = run<0, 32, 1, OP>(512), bx = run<4, 32, 1, OP>(512);
    printf("%-14s same wave: MFMA %4.0f + 32 VALU %4.0f -> together %4.0f ns   | cross wave: %4.0f + %4.0f -> %4.0f ns\n", name, m1, v1, b1, mx, vx,
           bx);
}
template <int OP>
void grain(const char* name) {
    // the same work per 16 MFMAs + 128 VALU instructions, interleaved at different grain (one wave per SIMD)
    const float a = run<1, 8, 0, OP>(256) * 16, b = run<2, 16, 0, OP>(256) * 8, c = run<4, 32, 0, OP>(256) * 4, d = run<8, 64, 0, OP>(256) * 2,
                e = run<16, 128, 0, OP>(256);
    const float a2 = run<1, 8, 0, OP>(512) * 16, c2 = run<4, 32, 0, OP>(512) * 4, e2 = run<16, 128, 0, OP>(512);
    printf("%-14s 16 MFMA + 128 VALU as (1+8)x16: %4.0f  (2+16)x8: %4.0f  (4+32)x4: %4.0f  (8+64)x2: %4.0f  (16+128): %4.0f ns;  two waves per SIMD, per wave: %4.0f %4.0f %4.0f\n",
           name, a, b, c, d, e, a2 / 2, c2 / 2, e2 / 2);
}
int main() {
    grain<0>("v_fma_f32");
    grain<1>("v_add_f32");
    grain<3>("v_pk_fma_f32");
    kind<0>("v_fma_f32");
    kind<1>("v_add_f32");
    kind<2>("v_mul_f32");
    kind<3>("v_pk_fma_f32");
    kind<4>("v_pk_add_f32");
    kind<7>("v_pk_mul_f32");
    kind<5>("v_xor_b32");
    kind<6>("v_mov_b32");
    kind<8>("v_cndmask_b32");
    printf("three different register operands per instruction:\n");
    kind<10>("v_fma_f32");
    kind<11>("v_add_f32");
    kind<13>("v_pk_fma_f32");
    kind<14>("v_pk_add_f32");
    kind<15>("v_pk_fma(y,1,z)");
    kind<17>("v_pk_mul_f32");
    return 0;
}

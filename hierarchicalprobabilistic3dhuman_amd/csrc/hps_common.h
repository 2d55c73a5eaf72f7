// Shared helpers for the libhps.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "hps.h"
#ifdef HPS_DEV_BUILD
#include "hps_dev.h"
#endif

namespace hps {

// thread-local last-error text returned by hps_last_error()
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return HPS_OK;
}

inline int bad_arg(const char* what) {
    set_error("bad argument: %s", what);
    return HPS_E_BADARG;
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Dynamic LDS above 64 KiB has to be granted per kernel function AND per device (the attribute lives with the device's copy
// of the code object): once per (kernel instantiation, current device), safe from several host threads, result checked.
// No behaviour depends on it.  Returns HPS_OK or the hipError_t.
template <auto Kernel>
inline int grant_lds(int bytes, const char* what) {
    static std::atomic<uint64_t> granted{0};               // bit d: done on device d
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && dev < 64 && ((granted.load(std::memory_order_acquire) >> dev) & 1ull)) return HPS_OK;
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("%s: granting %d bytes of dynamic LDS failed: %s", what, bytes, hipGetErrorString(e));
        return (int)e;
    }
    if (dev < 64) granted.fetch_or(1ull << dev, std::memory_order_release);
    return HPS_OK;
}

// 12-byte vertex record: one global_load_dwordx3 / global_store_dwordx3 per lane, lane-contiguous.
struct __attribute__((packed, aligned(4))) f3 {
    float x, y, z;
};

__device__ __forceinline__ float det3(const float* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// C = A * B, 3x3 row-major
__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

// C = A * B^T, 3x3 row-major
__device__ __forceinline__ void mat3_mul_bt(const float* a, const float* b, float* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[i * 3 + j] = a[i * 3 + 0] * b[j * 3 + 0] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
}

// utils/rigid_transform_utils.py:113-133 -- q = (w,x,y,z), re-normalised, row-major 3x3 out
__device__ __forceinline__ void quat_to_rotmat_dev(float qw, float qx, float qy, float qz, float* r) {
    float n = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    float w = qw / n, x = qx / n, y = qy / n, z = qz / n;
    float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    r[0] = w2 + x2 - y2 - z2; r[1] = 2 * xy - 2 * wz;     r[2] = 2 * wy + 2 * xz;
    r[3] = 2 * wz + 2 * xy;   r[4] = w2 - x2 + y2 - z2;   r[5] = 2 * yz - 2 * wx;
    r[6] = 2 * xz - 2 * wy;   r[7] = 2 * wx + 2 * yz;     r[8] = w2 - x2 - y2 + z2;
}

// smplx.lbs.batch_rodrigues: angle = ||r + 1e-8||, R = I + sin K + (1 - cos) K^2, K = [r/angle]_x
__device__ __forceinline__ void rodrigues_dev(float rx, float ry, float rz, float* r) {
    float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
    float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    float dx = rx / angle, dy = ry / angle, dz = rz / angle;
    float s = sinf(angle), c = 1.0f - cosf(angle);
    // K = [[0,-dz,dy],[dz,0,-dx],[-dy,dx,0]];  K^2 computed as the matrix product
    float k[9] = {0.f, -dz, dy, dz, 0.f, -dx, -dy, dx, 0.f};
    float k2[9];
    mat3_mul(k, k, k2);
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + s * k[i] + c * k2[i];
}


// Linear blend skinning of one vertex (smplx lbs step (5)): T = sum_k w_k A[idx_k] (3x4, idx pre-multiplied by 12),
// out = T [p; 1] + t.  One definition with explicit fused multiply-adds, shared by lbs_kernel and the fused mesh kernel,
// so that both paths produce the same bits.
template <int K>
__device__ __forceinline__ f3 skin_vertex(const float* Am, const int (&idx)[K], const float (&w)[K], const f3 pv, float tx,
                                          float ty, float tz) {
    // The blend T = sum_k w_k A_k as packed fp32 FMAs (v_pk_fma_f32: two IEEE FMAs per lane and instruction, the same bits as
    // twelve scalar ones at half the issue slots -- VALU time is paid in MFMA time in the fused kernel).
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f P[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) P[e] = (v2f){0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4* t4 = reinterpret_cast<const float4*>(Am + idx[k]);
        const float4 r0 = t4[0], r1 = t4[1], r2 = t4[2];
        const v2f wk = (v2f){w[k], w[k]};
        P[0] = __builtin_elementwise_fma(wk, (v2f){r0.x, r0.y}, P[0]);
        P[1] = __builtin_elementwise_fma(wk, (v2f){r0.z, r0.w}, P[1]);
        P[2] = __builtin_elementwise_fma(wk, (v2f){r1.x, r1.y}, P[2]);
        P[3] = __builtin_elementwise_fma(wk, (v2f){r1.z, r1.w}, P[3]);
        P[4] = __builtin_elementwise_fma(wk, (v2f){r2.x, r2.y}, P[4]);
        P[5] = __builtin_elementwise_fma(wk, (v2f){r2.z, r2.w}, P[5]);
    }
    const float T[12] = {P[0].x, P[0].y, P[1].x, P[1].y, P[2].x, P[2].y, P[3].x, P[3].y, P[4].x, P[4].y, P[5].x, P[5].y};
    f3 o;
    o.x = __builtin_fmaf(T[2], pv.z, __builtin_fmaf(T[1], pv.y, T[0] * pv.x)) + T[3] + tx;
    o.y = __builtin_fmaf(T[6], pv.z, __builtin_fmaf(T[5], pv.y, T[4] * pv.x)) + T[7] + ty;
    o.z = __builtin_fmaf(T[10], pv.z, __builtin_fmaf(T[9], pv.y, T[8] * pv.x)) + T[11] + tz;
    return o;
}

// v / 3, correctly rounded, without the ~12-instruction division expansion: q = v * RN(1/3), then one Newton correction with two
// fused multiply-adds -- r = v - 3 q is exact, q + r * RN(1/3) rounds to RN(v / 3); r == 0 means q is already the quotient (this
// also keeps the sign of -0).  Bit-identical to the IEEE quotient for EVERY one of the 4 278 190 080 finite fp32 values
// (tools/div3_check.hip runs them all on the device; numpy agreed on the host).
__device__ __forceinline__ float div3_rn(float v) {
    const float inv = 0.3333333432674407958984375f;              // RN(1/3) = 0x3EAAAAAB
    const float q = v * inv;
    const float r = __builtin_fmaf(-3.0f, q, v);
    return r == 0.0f ? q : __builtin_fmaf(r, inv, q);
}

// One 1 KiB LDS-DMA piece: every lane moves 16 bytes from sbase + voff (bytes) to LDS[lds_addr + 16 * lane].
__device__ __forceinline__ void lds_dma16(unsigned voff, const float* sbase, unsigned lds_addr) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);       // wave-uniform by construction; M0 needs an SGPR
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

}  // namespace hps

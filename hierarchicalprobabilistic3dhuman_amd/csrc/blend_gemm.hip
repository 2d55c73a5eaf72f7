// Shape + pose blend shapes as one fp32 GEMM on the gfx950 matrix cores:
//   v_posed[m, n] = v_template[n] + sum_k xt[k, m] * bmat[k, n]
// (smplx lbs steps (1)+(3): blend_shapes einsum 'bl,mkl->bmk' and pose_feature @ posedirs;
//  SURVEY.md section 8 row A11).  M = meshes, N = 3 * 6890, K = num_betas + 207 (padded to kp).
//
// Both operands are k-major in HBM, so a K-chunk of either is BK contiguous 512-byte rows that go
// to LDS unchanged, and the v_mfma_f32_32x32x2_f32 fragments (A[i][k]: lane = i + 32 k,
// B[k][j]: lane = j + 32 k) are conflict-free 32-lane row reads.
// Workgroup tile 128 (meshes) x 128 (coords), 4 waves as 2 x 2, each wave 2 x 2 MFMA tiles of 32 x 32.
// (Tried: the convolution kernels' LDS-DMA + 128-bit-fragment staging on this GEMM -- 0.634 ms vs 0.615 ms here: with
//  only K = 224 the kernel is bound by its 540 MB epilogue write and tile prologue, not by the K loop.)
#include "hps_common.h"

namespace hps {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16;

__global__ __launch_bounds__(256) void blend_gemm_kernel(const float* __restrict__ xt, const float* __restrict__ bmat,
                                                         const float* __restrict__ v_template,
                                                         float* __restrict__ out, int M, int N, int kp, int mp,
                                                         int np, int tiles_n, int ld_out) {
    __shared__ __attribute__((aligned(16))) float sA[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float sB[2][BK][BN];

    // mesh-tile major: the workgroups in flight share one 128-mesh slice of xt (115 KB, L2-hot) and walk the coordinate
    // panels of bmat (18.6 MB, resident in the memory-side cache); measured 0.6 % faster end to end than panel-major
    const int tile_m = blockIdx.x / tiles_n;
    const int tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // staging: BK rows x 32 float4 per operand = 512 float4, 2 per thread per operand
    const int srow = tid >> 5;          // 0..7  (+8 for the second)
    const int scol = (tid & 31) * 4;    // float offset in the row
    const float* ga = xt + (size_t)srow * mp + m0 + scol;
    const float* gb = bmat + (size_t)srow * np + n0 + scol;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nchunks = kp / BK;
    float4 ra0, ra1, rb0, rb1;
    ra0 = *reinterpret_cast<const float4*>(ga);
    ra1 = *reinterpret_cast<const float4*>(ga + (size_t)8 * mp);
    rb0 = *reinterpret_cast<const float4*>(gb);
    rb1 = *reinterpret_cast<const float4*>(gb + (size_t)8 * np);

    const int kl = lane >> 5;       // k within an MFMA step
    const int il = lane & 31;       // row / column within a 32-tile
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        *reinterpret_cast<float4*>(&sA[buf][srow][scol]) = ra0;
        *reinterpret_cast<float4*>(&sA[buf][srow + 8][scol]) = ra1;
        *reinterpret_cast<float4*>(&sB[buf][srow][scol]) = rb0;
        *reinterpret_cast<float4*>(&sB[buf][srow + 8][scol]) = rb1;
        __syncthreads();
        if (c + 1 < nchunks) {   // next chunk's global loads fly under this chunk's MFMAs
            const size_t ko = (size_t)(c + 1) * BK;
            ra0 = *reinterpret_cast<const float4*>(ga + ko * mp);
            ra1 = *reinterpret_cast<const float4*>(ga + (ko + 8) * mp);
            rb0 = *reinterpret_cast<const float4*>(gb + ko * np);
            rb1 = *reinterpret_cast<const float4*>(gb + (ko + 8) * np);
        }
#pragma unroll
        for (int k = 0; k < BK; k += 2) {
            const float a0 = sA[buf][k + kl][wm * 64 + il];
            const float a1 = sA[buf][k + kl][wm * 64 + 32 + il];
            const float b0 = sB[buf][k + kl][wn * 64 + il];
            const float b1 = sB[buf][k + kl][wn * 64 + 32 + il];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        // double-buffered LDS: the next iteration writes the other buffer; the barrier at its top
        // orders those writes against this iteration's reads of that buffer two chunks ago.
    }

    // epilogue: C layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + il;
        if (n >= N) continue;
        const float vt = v_template[n];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                if (m < M) out[(size_t)m * ld_out + n] = vt + acc[tm][tn][r];
            }
        }
    }
}

}  // namespace hps

using namespace hps;

extern "C" int hps_smpl_blend(const float* xt, const float* bmat, const float* v_template, float* v_posed, int M,
                              int N, int kp, int mp, int np, int ld_out, hps_stream_t stream) {
    if (!xt || !bmat || !v_template || !v_posed) return bad_arg("hps_smpl_blend: null pointer");
    if (kp <= 0 || kp % BK != 0) return bad_arg("hps_smpl_blend: kp must be a positive multiple of 16");
    if (mp % BM != 0 || mp < M || np % BN != 0 || np < N) return bad_arg("hps_smpl_blend: mp/np must be multiples of 128 covering M/N");
    if (ld_out < N) return bad_arg("hps_smpl_blend: ld_out < N");
    if (M <= 0 || N <= 0) return HPS_OK;
    const int tiles_m = ceil_div(M, BM), tiles_n = ceil_div(N, BN);
    hipLaunchKernelGGL(blend_gemm_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, (hipStream_t)stream, xt, bmat,
                       v_template, v_posed, M, N, kp, mp, np, tiles_n, ld_out);
    return check_launch("hps_smpl_blend");
}

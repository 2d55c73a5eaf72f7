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

#ifdef HPS_DEV_BUILD      // alternate kernel kept for A/B runs in the dev library only
// ---------------------------------------------------------------------------------------------
// Stationary-A form.  With K = 224 a 128 x 128 tile is only 7-14 K-chunks long, so the tile prologue (first operand
// fetch) and epilogue are ~20 % of blend_gemm_kernel.  Here a workgroup owns one 128-mesh slice of xt for its whole life:
// every lane loads its MFMA A fragments for ALL of K once (KP/2 registers: wave w holds rows 32 w .. 32 w + 31) and then
// walks a range of coordinate panels, streaming only bmat through LDS (LDS-DMA, one chunk ahead, continuing across
// panels -- no per-tile prologue).  Same MFMA order per output element as blend_gemm_kernel: identical bits.
// ---------------------------------------------------------------------------------------------

template <int KP>
__global__ __launch_bounds__(256, 2) void blend_gemm_sa_kernel(const float* __restrict__ xt, const float* __restrict__ bmat,
                                                               const float* __restrict__ v_template,
                                                               float* __restrict__ out, int M, int N, int mp, int np,
                                                               int tiles_n, int n_splits, int ld_out) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int SBK = 32;                            // K rows per streamed chunk (64 MFMAs per wave between barriers)
    constexpr int NCH = KP / SBK;                      // chunks per panel
    constexpr int PPW = SBK / 8;                       // one-KiB DMA pieces per wave per chunk (a piece = 2 rows of 512 B)
    static_assert(KP % SBK == 0, "K must be a multiple of the chunk");
    __shared__ __attribute__((aligned(16))) float sB[2][SBK][BN];
    const int tile_m = blockIdx.x / n_splits, split = blockIdx.x % n_splits;
    const int per = (tiles_n + n_splits - 1) / n_splits;
    const int nt_begin = split * per, nt_end = min(tiles_n, nt_begin + per);
    if (nt_begin >= nt_end) return;
    const int m0 = tile_m * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kl = lane >> 5, il = lane & 31;
    const bool vec_ok = (ld_out & 3) == 0 && ((size_t)out & 15) == 0 && ((size_t)v_template & 15) == 0;

    // A fragments of rows m0 + 32 wave + il for every k pair: element p is xt[2 p + kl][row]
    float areg[KP / 2];
    {
        const float* src = xt + (size_t)kl * mp + m0 + wave * 32 + il;
#pragma unroll
        for (int p = 0; p < KP / 2; ++p) areg[p] = src[(size_t)(2 * p) * mp];
    }

    // bmat stream: piece q of a chunk covers rows 2 q, 2 q + 1; wave w issues pieces PPW w .. PPW w + PPW - 1
    const int q0 = wave * PPW;
    unsigned b_off[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) b_off[j] = (unsigned)(((2 * (q0 + j) + (lane >> 5)) * np + (lane & 31) * 4) * 4);
    const unsigned lds_b = (unsigned)(size_t)(lptr_t)(&sB[0][0][0]) + (unsigned)q0 * 2 * BN * 4;
    const float* b_src = bmat + (size_t)nt_begin * BN;        // chunk 0 of the first panel
    int c_next = 0, nt_next = nt_begin;                         // the chunk the next DMA fetches
    auto dma_next = [&](int buf) {
        const unsigned lb = __builtin_amdgcn_readfirstlane(lds_b + buf * SBK * BN * 4);
#pragma unroll
        for (int j = 0; j < PPW; ++j) lds_dma16(b_off[j], b_src, lb + j * 2 * BN * 4);
        if (++c_next < NCH) b_src += (size_t)SBK * np;
        else { c_next = 0; ++nt_next; b_src += (size_t)BN - (size_t)(NCH - 1) * SBK * np; }
    };

    int buf = 0;
    dma_next(0);
    for (int nt = nt_begin; nt < nt_end; ++nt) {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (nt_next < nt_end) dma_next(buf ^ 1);
#pragma unroll
            for (int k = 0; k < SBK; k += 2) {
                const float a = areg[(c * SBK + k) / 2];
                const float* brow = &sB[buf][k + kl][il];
                const float b0 = brow[0], b1 = brow[32], b2 = brow[64], b3 = brow[96];
                // the bmat fragment is the MFMA's row operand, the mesh fragment its column operand (products commute,
                // same k order): a lane ends with ONE mesh (column = lane & 31) and quads of consecutive coordinates
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2, a, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b3, a, acc[3], 0, 0, 0);
            }
            buf ^= 1;
        }
        // epilogue: row (coordinate) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column (mesh) = lane & 31:
        // 16-byte stores of four consecutive coordinates when the rows of `out` are 16-byte aligned
        const int n0 = nt * BN;
        const int m = m0 + wave * 32 + il;
        if (m < M) {
            float* orow = out + (size_t)m * ld_out;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + j * 32 + 8 * q + 4 * kl;
                    if (vec_ok && n + 3 < N) {
                        const float4 vt = *reinterpret_cast<const float4*>(v_template + n);
                        *reinterpret_cast<float4*>(orow + n) = make_float4(vt.x + acc[j][4 * q], vt.y + acc[j][4 * q + 1],
                                                                           vt.z + acc[j][4 * q + 2], vt.w + acc[j][4 * q + 3]);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (n + t < N) orow[n + t] = v_template[n + t] + acc[j][4 * q + t];
                    }
                }
        }
    }
}

#endif  // HPS_DEV_BUILD

}  // namespace hps

using namespace hps;

#ifdef HPS_DEV_BUILD
static int g_blend_mode = 0;      // hps_dev_blend_mode: 0 / 1 = tiled kernel (default), 2 = stationary-A kernel
extern "C" int hps_dev_blend_mode(int mode) {
    g_blend_mode = mode;
    return HPS_OK;
}
#endif

extern "C" int hps_smpl_blend(const float* xt, const float* bmat, const float* v_template, float* v_posed, int M,
                              int N, int kp, int mp, int np, int ld_out, hps_stream_t stream) {
    if (!xt || !bmat || !v_template || !v_posed) return bad_arg("hps_smpl_blend: null pointer");
    if (kp <= 0 || kp % BK != 0) return bad_arg("hps_smpl_blend: kp must be a positive multiple of 16");
    if (mp % BM != 0 || mp < M || np % BN != 0 || np < N) return bad_arg("hps_smpl_blend: mp/np must be multiples of 128 covering M/N");
    if (ld_out < N) return bad_arg("hps_smpl_blend: ld_out < N");
    if (M <= 0 || N <= 0) return HPS_OK;
    const int tiles_m = ceil_div(M, BM), tiles_n = ceil_div(N, BN);
#ifdef HPS_DEV_BUILD
    if (kp == 224 && g_blend_mode == 2 && (size_t)kp * np * 4 < 0xffffffffull) {
        // stationary-A kernel (opt-in, hps_dev_blend_mode(2)): alone it is 15 % faster than the tiled kernel (0.54 vs
        // 0.65 ms at 6 528 meshes, 120 vs 102 TF/s at 16 032), but inside the pipelined step the difference shrinks to
        // 5 % (529 vs 558 us) and the LBS kernel that reads its output right after runs 3 % slower (16-byte store
        // segments instead of whole 128-byte lines): +0.4 % images/s, -2 points of LBS roofline -- not the default.
        // one resident round: at most 512 workgroups (2 per CU), never a sparsely filled second round
        int n_splits = 512 / tiles_m;
        if (n_splits < 1) n_splits = 1;
        if (n_splits > tiles_n) n_splits = tiles_n;
        hipLaunchKernelGGL(blend_gemm_sa_kernel<224>, dim3(tiles_m * n_splits), dim3(256), 0, (hipStream_t)stream, xt, bmat,
                           v_template, v_posed, M, N, mp, np, tiles_n, n_splits, ld_out);
        return check_launch("hps_smpl_blend");
    }
#endif
    hipLaunchKernelGGL(blend_gemm_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, (hipStream_t)stream, xt, bmat,
                       v_template, v_posed, M, N, kp, mp, np, tiles_n, ld_out);
    return check_launch("hps_smpl_blend");
}

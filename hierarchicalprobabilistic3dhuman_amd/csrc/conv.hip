// ResNet-18 encoder kernels (models/resnet.py:202-217): implicit-GEMM convolution
// on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32) with eval-mode BatchNorm, residual add
// and ReLU fused into the epilogue; NHWC max-pool / global average pool; NCHW -> NHWC input relayout.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.  Activations are NHWC so a K-run of
// one filter tap is contiguous channels of one input pixel; the filter is stored k-major (K, Cout).
#include "hps_common.h"

namespace hps {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Epilogue shared by the MFMA convolution kernels: eval-mode BatchNorm (scale, shift), residual add, ReLU.
// C layout of a 32x32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
// Full tiles take the branch-free path: all residual loads of a 32x32 tile are issued before the first use.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[TM][TN], const float* __restrict__ scale,
                                              const float* __restrict__ shift, const float* __restrict__ residual,
                                              float* __restrict__ y, int mbase, int nbase, int il, int kl, int Cout,
                                              int Mtot, int relu, bool full_tile) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = nbase + j * 32 + il;
        const float sc = scale[co], sh = shift[co];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mrow = mbase + i * 32 + 4 * kl;
            if (full_tile) {
                float res[16];
                if (residual) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[r] = residual[(size_t)(mrow + (r & 3) + 8 * (r >> 2)) * Cout + co];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) res[r] = 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * sc + sh + res[r];
                    if (relu) v = fmaxf(v, 0.0f);
                    y[(size_t)(mrow + (r & 3) + 8 * (r >> 2)) * Cout + co] = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mrow + (r & 3) + 8 * (r >> 2);
                    if (m < Mtot) {
                        float v = acc[i][j][r] * sc + sh;
                        if (residual) v += residual[(size_t)m * Cout + co];
                        if (relu) v = fmaxf(v, 0.0f);
                        y[(size_t)m * Cout + co] = v;
                    }
                }
            }
        }
    }
}

constexpr int CBK = 16;        // K-chunk
constexpr int LDA = CBK + 1;   // odd pitch: the 32 pixel rows a half-wave reads hit 32 distinct banks

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const float* __restrict__ x, const float* __restrict__ wk, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ y, int H, int W,
    int Cin, int Cout, int KW, int stride, int pad, int Ho, int Wo, int Mtot, int Kreal, int Kp, int relu,
    int tiles_m) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM * 4 / 256;       // float4 loads per thread for the A chunk
    constexpr int B_LD = (CBK * BN / 4 + 255) / 256;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    static_assert(BM * 4 % 256 == 0, "A tile divides over 256 threads");

    __shared__ float sA[2][BM][LDA];
    __shared__ __attribute__((aligned(16))) float sB[2][CBK][BN];

    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x % tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int kl = lane >> 5, il = lane & 31;

    // ---- per-thread A-load coordinates (fixed over the K loop) ----
    int a_pix[A_LD];          // pixel row inside the tile
    int a_hi0[A_LD], a_wi0[A_LD];
    long a_base[A_LD];        // element offset of image b, or -1 if the pixel is outside M
    const int a_q = (tid & 3) * 4;   // k offset inside the chunk
#pragma unroll
    for (int r = 0; r < A_LD; ++r) {
        const int f = tid + r * 256;
        a_pix[r] = f >> 2;
        const int m = m0 + a_pix[r];
        if (m < Mtot) {
            const int b = m / (Ho * Wo), rem = m % (Ho * Wo);
            a_hi0[r] = (rem / Wo) * stride - pad;
            a_wi0[r] = (rem % Wo) * stride - pad;
            a_base[r] = (long)b * H * W * Cin;
        } else {
            a_hi0[r] = 0; a_wi0[r] = 0; a_base[r] = -1;
        }
    }
    constexpr int BN4 = BN / 4;

    auto load_a = [&](int k0, float4* ra) {
        const int k = k0 + a_q;
        const int tap = k / Cin, ci = k - tap * Cin;
        const int kh = tap / KW, kw = tap - kh * KW;
        const bool kvalid = k < Kreal;
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            const int hi = a_hi0[r] + kh, wi = a_wi0[r] + kw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kvalid && a_base[r] >= 0 && hi >= 0 && hi < H && wi >= 0 && wi < W)
                v = *reinterpret_cast<const float4*>(x + a_base[r] + ((long)hi * W + wi) * Cin + ci);
            ra[r] = v;
        }
    };
    auto load_b = [&](int k0, float4* rb) {
#pragma unroll
        for (int r = 0; r < B_LD; ++r) {
            const int f = tid + r * 256;
            const int row = f / BN4, c4 = f - row * BN4;
            if (row < CBK) rb[r] = *reinterpret_cast<const float4*>(wk + (size_t)(k0 + row) * Cout + n0 + c4 * 4);
        }
    };
    auto store_ab = [&](int buf, const float4* ra, const float4* rb) {
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            float* d = &sA[buf][a_pix[r]][a_q];
            d[0] = ra[r].x; d[1] = ra[r].y; d[2] = ra[r].z; d[3] = ra[r].w;
        }
#pragma unroll
        for (int r = 0; r < B_LD; ++r) {
            const int f = tid + r * 256;
            const int row = f / BN4, c4 = f - row * BN4;
            if (row < CBK) *reinterpret_cast<float4*>(&sB[buf][row][c4 * 4]) = rb[r];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[A_LD], rb[B_LD];
    load_a(0, ra);
    load_b(0, rb);
    const int nchunks = Kp / CBK;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        store_ab(buf, ra, rb);
        __syncthreads();
        if (c + 1 < nchunks) {           // next chunk's global loads overlap this chunk's MFMAs
            load_a((c + 1) * CBK, ra);
            load_b((c + 1) * CBK, rb);
        }
#pragma unroll
        for (int k = 0; k < CBK; k += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = sA[buf][wm0 + i * 32 + il][k + kl];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = sB[buf][k + kl][wn0 + j * 32 + il];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    conv_epilogue<TM, TN>(acc, scale, shift, residual, y, m0 + wm0, n0 + wn0, il, kl, Cout, Mtot, relu, m0 + BM <= Mtot);
}

// ---------------------------------------------------------------------------------------------
// v2 implicit GEMM for Cin % 32 == 0 (every 3x3 / 1x1 convolution of ResNet-18 after the stem).
//   * a K-chunk of 32 lies inside ONE filter tap, so the tap decode and the channel offset are scalar
//     (wave-uniform) and the per-lane address work is one multiply-add and a bounds test per row;
//   * both LDS tiles are [row][36]: rows = output pixels (A) / output channels (B), k contiguous, 144-byte
//     pitch.  Stores are ds_write_b128 (8 lanes fill one 128-byte row), fragment loads are ds_read_b128 and
//     both are bank-conflict free at this pitch;
//   * the MFMA k index is a free permutation as long as A and B agree: lane-half kl of 32x32x2 step t of
//     8-wide group g takes actual k = 8g + 4kl + t, which is exactly what one 16-byte read delivers -- one
//     LDS instruction feeds four MFMA steps instead of one;
//   * filters are stored n-major (Cout, Kp) so a B row is 128 contiguous bytes like an A row.
// Double-buffered LDS, next chunk prefetched into registers under the MFMAs, one barrier per chunk.
// ---------------------------------------------------------------------------------------------
constexpr int VBK = 32;
constexpr int VPITCH = VBK + 4;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_v2_kernel(
    const float* __restrict__ x, const float* __restrict__ wn, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ y, int H, int W,
    int Cin, int Cout, int KW, int stride, int pad, int Ho, int Wo, int Mtot, int Kp, int relu, int tiles_m) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM / 32, B_LD = BN / 32;        // float4 loads per thread per chunk
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                   // [2][BM][VPITCH]
    float* sB = smem + 2 * BM * VPITCH;                 // [2][BN][VPITCH]

    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x % tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int kl = lane >> 5, il = lane & 31;
    const int lrow = tid >> 3, lq = (tid & 7) * 4;      // staging: row within a 32-row slab, float offset in the chunk

    int a_hi0[A_LD], a_wi0[A_LD];
    long a_base[A_LD];
#pragma unroll
    for (int r = 0; r < A_LD; ++r) {
        const int m = m0 + lrow + 32 * r;
        if (m < Mtot) {
            const int b = m / (Ho * Wo), rem = m - b * (Ho * Wo);
            const int ho = rem / Wo, wo = rem - ho * Wo;
            a_hi0[r] = ho * stride - pad;
            a_wi0[r] = wo * stride - pad;
            a_base[r] = (long)b * H * W * Cin + lq;
        } else {
            a_hi0[r] = -(1 << 20); a_wi0[r] = 0; a_base[r] = 0;      // never in bounds
        }
    }
    const float* b_src = wn + (size_t)(n0 + lrow) * Kp + lq;
    const int chunks_per_tap = Cin / VBK;

    // staging registers as scalars (float4 arrays written under a branch are not promoted out of scratch)
    float ra[A_LD][4], rb[B_LD][4];
    auto load_chunk = [&](int c) {
        const int tap = c / chunks_per_tap;                      // wave-uniform
        const int ci0 = (c - tap * chunks_per_tap) * VBK;
        const int kh = tap / KW, kw = tap - kh * KW;
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            const int hi = a_hi0[r] + kh, wi = a_wi0[r] + kw;
            const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
            const long off = ok ? a_base[r] + ((long)hi * W + wi) * Cin + ci0 : 0;     // clamped: always a valid address
            const float4 v = *reinterpret_cast<const float4*>(x + off);
            ra[r][0] = ok ? v.x : 0.f; ra[r][1] = ok ? v.y : 0.f; ra[r][2] = ok ? v.z : 0.f; ra[r][3] = ok ? v.w : 0.f;
        }
#pragma unroll
        for (int r = 0; r < B_LD; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(b_src + (size_t)(32 * r) * Kp + (size_t)c * VBK);
            rb[r][0] = v.x; rb[r][1] = v.y; rb[r][2] = v.z; rb[r][3] = v.w;
        }
    };
    auto store_chunk = [&](int buf) {
        float* da = sA + (size_t)buf * BM * VPITCH + lrow * VPITCH + lq;
        float* db = sB + (size_t)buf * BN * VPITCH + lrow * VPITCH + lq;
#pragma unroll
        for (int r = 0; r < A_LD; ++r)
            *reinterpret_cast<float4*>(da + 32 * r * VPITCH) = make_float4(ra[r][0], ra[r][1], ra[r][2], ra[r][3]);
#pragma unroll
        for (int r = 0; r < B_LD; ++r)
            *reinterpret_cast<float4*>(db + 32 * r * VPITCH) = make_float4(rb[r][0], rb[r][1], rb[r][2], rb[r][3]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    load_chunk(0);
    const int nchunks = Kp / VBK;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        store_chunk(buf);
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
        const float* pa = sA + (size_t)buf * BM * VPITCH + (wm0 + il) * VPITCH + 4 * kl;
        const float* pb = sB + (size_t)buf * BN * VPITCH + (wn0 + il) * VPITCH + 4 * kl;
#pragma unroll
        for (int g = 0; g < VBK / 8; ++g) {
            float4 a4[TM], b4[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const float4*>(pa + i * 32 * VPITCH + 8 * g);
#pragma unroll
            for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4*>(pb + j * 32 * VPITCH + 8 * g);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }

    conv_epilogue<TM, TN>(acc, scale, shift, residual, y, m0 + wm0, n0 + wn0, il, kl, Cout, Mtot, relu, m0 + BM <= Mtot);
}

// ---------------------------------------------------------------------------------------------
// v3: v2's tiling with LDS filled directly from global memory (global_load_lds_dwordx4, 1 KiB per wave
// instruction = 8 tile rows of 128 bytes) -- no staging registers, no ds_write pass, no per-element zero
// select.  An LDS-DMA destination is lane-linear, so the tile rows are unpadded (128-byte pitch) and the bank
// conflicts of the 128-bit fragment reads are removed by an XOR swizzle applied on the SOURCE side: lane
// (row, q) fetches 16-byte quad q ^ f(row) of its row, the reader of quad G of row r looks in slot G ^ f(r),
// f(r) = (r >> 1) & 7 (conflict-free for ds_read_b128's 16-lane groups: checked in DESIGN.md).
// Out-of-image taps read from a small zero buffer.  Two LDS buffers: the DMA of chunk c+1 flies under the
// MFMAs of chunk c; one barrier per chunk.
// ---------------------------------------------------------------------------------------------
// One LDS-DMA instruction issued from inline asm: hipcc does not count it, so it does not drain it with a
// vmcnt(0) in front of the next ds_read (which is what the builtin form does and what would serialise the DMA of
// chunk c+1 with the MFMAs of chunk c).  The kernel waits for it explicitly (s_waitcnt vmcnt(0) + barrier) before
// the buffer is read.  lds_addr: wave-uniform LDS byte address; lane l lands at lds_addr + 16 l.  M0 is saved and
// restored because the compiler owns it (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

template <int BM, int BN, int WM, int WN, int ABLATE = 0>   // ABLATE (tuning only): 1 = no DMA after chunk 0, 2 = no MFMA
__global__ __launch_bounds__(256) void conv_igemm_v3_kernel(
    const float* __restrict__ x, const float* __restrict__ wn, const float* __restrict__ zeros,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ residual,
    float* __restrict__ y, int H, int W, int Cin, int Cout, int KW, int stride, int pad, int Ho, int Wo, int Mtot,
    int Kp, int relu, int tiles_m, int ksplit, float* __restrict__ partial) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM / 32, B_LD = BN / 32;        // DMA instructions per wave per chunk
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                   // [2][BM][32]
    float* sB = smem + 2 * BM * VBK;                    // [2][BN][32]

    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x % tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int kl = lane >> 5, il = lane & 31;
    // DMA role of this lane: tile row (r * 32 + wave * 8 + lane / 8), slot lane % 8, source quad slot ^ f(row)
    const int drow = wave * 8 + (lane >> 3);
    const int dquad = ((lane & 7) ^ ((drow >> 1) & 7)) * 4;      // float offset of the source quad (f(row + 32 r) = f(row))

    int a_hi0[A_LD], a_wi0[A_LD];
    long a_base[A_LD];
#pragma unroll
    for (int r = 0; r < A_LD; ++r) {
        const int m = m0 + drow + 32 * r;
        if (m < Mtot) {
            const int b = m / (Ho * Wo), rem = m - b * (Ho * Wo);
            const int ho = rem / Wo, wo = rem - ho * Wo;
            a_hi0[r] = ho * stride - pad;
            a_wi0[r] = wo * stride - pad;
            a_base[r] = (long)b * H * W * Cin + dquad;
        } else {
            a_hi0[r] = -(1 << 20); a_wi0[r] = 0; a_base[r] = 0;
        }
    }
    const float* b_src = wn + (size_t)(n0 + drow) * Kp + dquad;
    const int chunks_per_tap = Cin / VBK;

    // pieces [p0, p1) of chunk c's A_LD + B_LD one-KiB DMA pieces (A rows first, then filter rows)
    auto dma_pieces = [&](int c, int buf, int p0, int p1) {
        const int tap = c / chunks_per_tap;                      // wave-uniform
        const int ci0 = (c - tap * chunks_per_tap) * VBK;
        const int kh = tap / KW, kw = tap - kh * KW;
        // wave-uniform LDS byte addresses; lanes land at +16 B each
        const unsigned la = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(lptr_t)(sA + (size_t)buf * BM * VBK + wave * 8 * VBK));
        const unsigned lb = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(lptr_t)(sB + (size_t)buf * BN * VBK + wave * 8 * VBK));
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            if (r < p0 || r >= p1) continue;
            if (ABLATE == 4) {   // experiment: no tap decode / bounds test (wrong results, same traffic volume)
                lds_dma16(x + a_base[r] + (long)(c % 8) * VBK, la + r * 32 * VBK * 4);
                continue;
            }
            const int hi = a_hi0[r] + kh, wi = a_wi0[r] + kw;
            const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
            const float* src = ok ? x + a_base[r] + ((long)hi * W + wi) * Cin + ci0 : zeros;
            lds_dma16(src, la + r * 32 * VBK * 4);
        }
#pragma unroll
        for (int r = 0; r < B_LD; ++r) {
            if (A_LD + r < p0 || A_LD + r >= p1) continue;
            lds_dma16(b_src + (size_t)(32 * r) * Kp + (size_t)c * VBK, lb + r * 32 * VBK * 4);
        }
    };
    auto dma_chunk = [&](int c, int buf) { dma_pieces(c, buf, 0, A_LD + B_LD); };
    constexpr int NPIECE = A_LD + B_LD, PPG = (NPIECE + VBK / 8 - 1) / (VBK / 8);   // pieces per MFMA group when spread

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int fsw = (il >> 1) & 7;                       // f(row) of the fragment rows this lane reads
    // split-K: blockIdx.y owns the chunk range [c_begin, c_end) and writes raw partial sums (reduced, in slice
    // order, by splitk_epilogue_kernel -- deterministic, no atomics)
    const int chunks_total = Kp / VBK;
    const int per_split = chunks_total / ksplit;
    const int c_begin = blockIdx.y * per_split, c_end = c_begin + per_split;
    dma_chunk(c_begin, 0);
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA of chunk c has landed
        __syncthreads();                                     // ... everyone's has, and buf^1 is no longer being read
        if (c + 1 < c_end && ABLATE != 1 && ABLATE != 3) dma_chunk(c + 1, buf ^ 1);
        const float* pa = sA + (size_t)buf * BM * VBK + (wm0 + il) * VBK;
        const float* pb = sB + (size_t)buf * BN * VBK + (wn0 + il) * VBK;
#pragma unroll
        for (int g = 0; g < VBK / 8; ++g) {
            const int slot = ((2 * g + kl) ^ fsw) * 4;
            float4 a4[TM], b4[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const float4*>(pa + i * 32 * VBK + slot);
#pragma unroll
            for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4*>(pb + j * 32 * VBK + slot);
            // ABLATE 3 (experiment): the next chunk's DMA pieces spread over the MFMA groups instead of all up front
            if (ABLATE == 3 && c + 1 < c_end) dma_pieces(c + 1, buf ^ 1, g * PPG, (g + 1) * PPG < NPIECE ? (g + 1) * PPG : NPIECE);
            if (ABLATE == 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(a4[i].x), "v"(a4[i].y), "v"(a4[i].z), "v"(a4[i].w));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(b4[j].x), "v"(b4[j].y), "v"(b4[j].z), "v"(b4[j].w));
                continue;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }

    if (ksplit > 1) {
        float* dst = partial + (size_t)blockIdx.y * Mtot * Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn0 + j * 32 + il;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                    if (m < Mtot) dst[(size_t)m * Cout + co] = acc[i][j][r];
                }
        }
        return;
    }
    conv_epilogue<TM, TN>(acc, scale, shift, residual, y, m0 + wm0, n0 + wn0, il, kl, Cout, Mtot, relu, m0 + BM <= Mtot);
}

// second pass of a split-K convolution: y = act(scale * (sum of the slices, in slice order) + shift + residual)
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ partial, int ksplit,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ residual, float* __restrict__ y,
                                                              long total4, int Cout, int relu) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 acc = reinterpret_cast<const float4*>(partial)[i];
    for (int k = 1; k < ksplit; ++k) {
        const float4 v = reinterpret_cast<const float4*>(partial)[(size_t)k * total4 + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int co = (int)((i * 4) % Cout);
    const float4 sc = *reinterpret_cast<const float4*>(scale + co), sh = *reinterpret_cast<const float4*>(shift + co);
    float4 o = make_float4(acc.x * sc.x + sh.x, acc.y * sc.y + sh.y, acc.z * sc.z + sh.z, acc.w * sc.w + sh.w);
    if (residual) {
        const float4 r = reinterpret_cast<const float4*>(residual)[i];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    reinterpret_cast<float4*>(y)[i] = o;
}

// (B,C,H,W) -> (B,H,W,Cp): lanes along w read each channel plane coalesced; every lane assembles its
// pixel's Cp channels and stores them as float4s.
template <int CP>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                           int HW, long total) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;   // pixel index over B*H*W
    if (p >= total) return;
    const long b = p / HW, hw = p % HW;
    float v[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) v[c] = (c < C) ? x[(b * C + c) * HW + hw] : 0.0f;
    float4* d = reinterpret_cast<float4*>(y + p * CP);
#pragma unroll
    for (int q = 0; q < CP / 4; ++q) d[q] = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
}

// MaxPool2d(3, 2, 1), NHWC, thread per (output pixel, 4 channels)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                      int C, int Ho, int Wo, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = C / 4;
    const int cq = (int)(i % c4);
    long p = i / c4;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const long b = p / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - 1 + kh;
        if (hi < 0 || hi >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - 1 + kw;
            if (wi < 0 || wi >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + ((b * H + hi) * W + wi) * C + cq * 4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<float4*>(y + i * 4) = m;
}

// global average pool: thread per (b, c), coalesced over c
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C,
                                                      int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = i / C, c = i % C;
    const float* s = x + (size_t)b * HW * C + c;
    float acc = 0.0f;
    for (int p = 0; p < HW; ++p) acc += s[(size_t)p * C];
    y[i] = acc / (float)HW;
}

template <int BM, int BN, int WM, int WN>
static int launch_conv(const float* x, const float* wk, const float* scale, const float* shift, const float* residual,
                       float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                       hipStream_t s) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const int Mtot = B * Ho * Wo, Kreal = KH * KW * Cin, Kp = ceil_div(Kreal, CBK) * CBK;
    const int tiles_m = ceil_div(Mtot, BM), tiles_n = Cout / BN;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), dim3(tiles_m * tiles_n), dim3(256), 0, s, x, wk, scale,
                       shift, residual, y, H, W, Cin, Cout, KW, stride, pad, Ho, Wo, Mtot, Kreal, Kp, relu, tiles_m);
    return check_launch("hps_conv2d_bn_act");
}

template <int BM, int BN, int WM, int WN>
static int launch_conv_v2(const float* x, const float* wn, const float* scale, const float* shift, const float* residual,
                          float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                          hipStream_t s) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const int Mtot = B * Ho * Wo, Kp = KH * KW * Cin;
    const int tiles_m = ceil_div(Mtot, BM), tiles_n = Cout / BN;
    const size_t lds = (size_t)2 * (BM + BN) * VPITCH * sizeof(float);
    if (lds > 64 * 1024)
        if (int rc = grant_lds<&conv_igemm_v2_kernel<BM, BN, WM, WN>>((int)lds, "hps_conv2d_bn_act_v2")) return rc;
    hipLaunchKernelGGL((conv_igemm_v2_kernel<BM, BN, WM, WN>), dim3(tiles_m * tiles_n), dim3(256), lds, s, x, wn, scale,
                       shift, residual, y, H, W, Cin, Cout, KW, stride, pad, Ho, Wo, Mtot, Kp, relu, tiles_m);
    return check_launch("hps_conv2d_bn_act_v2");
}

template <int BM, int BN, int WM, int WN, int ABLATE = 0>
static int launch_conv_v3(const float* x, const float* wn, const float* zeros, const float* scale, const float* shift,
                          const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                          int stride, int pad, int relu, hipStream_t s, int ksplit = 1, float* partial = nullptr) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const int Mtot = B * Ho * Wo, Kp = KH * KW * Cin;
    const int tiles_m = ceil_div(Mtot, BM), tiles_n = Cout / BN;
    if (ksplit < 1 || (Kp / VBK) % ksplit != 0 || (ksplit > 1 && !partial)) return bad_arg("hps_conv2d_bn_act_v3: ksplit");
    const size_t lds = (size_t)2 * (BM + BN) * VBK * sizeof(float);
    if (lds > 64 * 1024)
        if (int rc = grant_lds<&conv_igemm_v3_kernel<BM, BN, WM, WN, ABLATE>>((int)lds, "hps_conv2d_bn_act_v3")) return rc;
    hipLaunchKernelGGL((conv_igemm_v3_kernel<BM, BN, WM, WN, ABLATE>), dim3(tiles_m * tiles_n, ksplit), dim3(256), lds, s, x,
                       wn, zeros, scale, shift, residual, y, H, W, Cin, Cout, KW, stride, pad, Ho, Wo, Mtot, Kp, relu, tiles_m,
                       ksplit, partial);
    if (ksplit > 1) {
        const long total4 = (long)Mtot * Cout / 4;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, partial, ksplit,
                           scale, shift, residual, y, total4, Cout, relu);
    }
    return check_launch("hps_conv2d_bn_act_v3");
}

}  // namespace hps

using namespace hps;

// variant: 1 = 128x128, 2 = 128x64, 3 = 64x64 tiles; zeros: >= 64 bytes of zeros on the device
extern "C" int hps_conv2d_bn_act_v3(const float* x, const float* wn, const float* zeros, const float* scale,
                                    const float* shift, const float* residual, float* y, int B, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad, int relu, int variant, int ksplit,
                                    float* splitk_ws, hps_stream_t stream) {
    if (!x || !wn || !zeros || !scale || !shift || !y) return bad_arg("hps_conv2d_bn_act_v3: null pointer");
    if (ksplit > 1)     // split-K runs on the 128x128 tile
        return launch_conv_v3<128, 128, 64, 64>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride,
                                                pad, relu, (hipStream_t)stream, ksplit, splitk_ws);
    if (Cin % 32 != 0 || Cout % 64 != 0) return bad_arg("hps_conv2d_bn_act_v3: Cin % 32 == 0 and Cout % 64 == 0 required");
    if (B <= 0) return HPS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0) {
        // measured per ResNet-18 layer at B = 64 (tests/dev/gpu_bringup.py conv_tune): the L2->LDS path per CU is the
        // limiter, so take the largest tile that still gives every CU a workgroup: 128x128 (>= 256 workgroups),
        // 256x64 for 64-channel outputs, else 64x64 (8 waves/SIMD)
        const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
        const long Mtot = (long)B * Ho * Wo;
        if (Cout % 128 == 0 && (Mtot / 128) * (Cout / 128) >= 256) variant = 1;
        else if (Cout == 64 && Mtot / 256 >= 512) variant = 4;
        else variant = 3;
    }
    if (variant % 10 == 1 && Cout % 128 != 0) variant = 2;
    switch (variant) {
        case 1: return launch_conv_v3<128, 128, 64, 64>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 2: return launch_conv_v3<128, 64, 64, 32>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 3: return launch_conv_v3<64, 64, 32, 32>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 4: return launch_conv_v3<256, 64, 64, 64>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 41: return launch_conv_v3<128, 128, 64, 64, 3>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 43: return launch_conv_v3<64, 64, 32, 32, 3>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 44: return launch_conv_v3<256, 64, 64, 64, 3>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 51: return launch_conv_v3<128, 128, 64, 64, 4>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 21: return launch_conv_v3<128, 128, 64, 64, 1>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 31: return launch_conv_v3<128, 128, 64, 64, 2>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 23: return launch_conv_v3<64, 64, 32, 32, 1>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 33: return launch_conv_v3<64, 64, 32, 32, 2>(x, wn, zeros, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        default: return bad_arg("hps_conv2d_bn_act_v3: variant");
    }
}

// variant: 0 = automatic, 1 = 128x128, 2 = 128x64, 3 = 64x64 tiles
extern "C" int hps_conv2d_bn_act_v2(const float* x, const float* wn, const float* scale, const float* shift,
                                    const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int KH,
                                    int KW, int stride, int pad, int relu, int variant, hps_stream_t stream) {
    if (!x || !wn || !scale || !shift || !y) return bad_arg("hps_conv2d_bn_act_v2: null pointer");
    if (Cin % 32 != 0 || Cout % 64 != 0) return bad_arg("hps_conv2d_bn_act_v2: Cin % 32 == 0 and Cout % 64 == 0 required");
    if (B <= 0) return HPS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const long Mtot = (long)B * Ho * Wo;
    if (variant == 0) {
        // measured per ResNet-18 layer at B = 64 (tests/dev/gpu_bringup.py conv_tune): the 64x64 tile (7 waves/SIMD)
        // wins everywhere except where 128x128 tiles still give every CU two workgroups
        variant = (Cout % 128 == 0 && (Mtot / 128) * (Cout / 128) >= 512) ? 1 : 3;
    }
    if (variant == 1 && Cout % 128 != 0) variant = 2;
    switch (variant) {
        case 1: return launch_conv_v2<128, 128, 64, 64>(x, wn, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 2: return launch_conv_v2<128, 64, 64, 32>(x, wn, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        case 3: return launch_conv_v2<64, 64, 32, 32>(x, wn, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
        default: return bad_arg("hps_conv2d_bn_act_v2: variant");
    }
}

extern "C" int hps_conv2d_bn_act(const float* x, const float* wk, const float* scale, const float* shift,
                                 const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int KH,
                                 int KW, int stride, int pad, int relu, hps_stream_t stream) {
    if (!x || !wk || !scale || !shift || !y) return bad_arg("hps_conv2d_bn_act: null pointer");
    if (Cin % 4 != 0 || Cout % 64 != 0) return bad_arg("hps_conv2d_bn_act: Cin % 4 == 0 and Cout % 64 == 0 required");
    if (B <= 0) return HPS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const long Mtot = (long)B * Ho * Wo;
    // largest tile that still gives every CU (256) at least two workgroups
    if (Cout % 128 == 0 && (Mtot / 128) * (Cout / 128) >= 512)
        return launch_conv<128, 128, 64, 64>(x, wk, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
    if ((Mtot / 128) * (Cout / 64) >= 512)
        return launch_conv<128, 64, 64, 32>(x, wk, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
    return launch_conv<64, 64, 32, 32>(x, wk, scale, shift, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, relu, s);
}

extern "C" int hps_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, int Cp, hps_stream_t stream) {
    if (!x || !y) return bad_arg("hps_nchw_to_nhwc: null pointer");
    if (C > Cp) return bad_arg("hps_nchw_to_nhwc: Cp < C");
    if (B <= 0) return HPS_OK;
    const long total = (long)B * H * W;
    const dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    switch (Cp) {
        case 4: hipLaunchKernelGGL(nchw_to_nhwc_kernel<4>, grid, dim3(256), 0, s, x, y, C, H * W, total); break;
        case 20: hipLaunchKernelGGL(nchw_to_nhwc_kernel<20>, grid, dim3(256), 0, s, x, y, C, H * W, total); break;
        case 64: hipLaunchKernelGGL(nchw_to_nhwc_kernel<64>, grid, dim3(256), 0, s, x, y, C, H * W, total); break;
        default: set_error("hps_nchw_to_nhwc: Cp=%d unsupported (4, 20, 64)", Cp); return HPS_E_UNSUPPORTED;
    }
    return check_launch("hps_nchw_to_nhwc");
}

extern "C" int hps_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, hps_stream_t stream) {
    if (!x || !y) return bad_arg("hps_maxpool3x3s2: null pointer");
    if (C % 4 != 0) return bad_arg("hps_maxpool3x3s2: C % 4 == 0 required");
    if (B <= 0) return HPS_OK;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, H,
                       W, C, Ho, Wo, total);
    return check_launch("hps_maxpool3x3s2");
}

extern "C" int hps_global_avgpool(const float* x, float* y, int B, int HW, int C, hps_stream_t stream) {
    if (!x || !y) return bad_arg("hps_global_avgpool: null pointer");
    if (B <= 0) return HPS_OK;
    hipLaunchKernelGGL(avgpool_kernel, dim3(ceil_div(B * C, 256)), dim3(256), 0, (hipStream_t)stream, x, y, HW, C, B * C);
    return check_launch("hps_global_avgpool");
}

// Winograd F(2x2, 3x3) convolution for the stride-1 3x3 layers of the ResNet-18 encoder (models/resnet.py:62-78: conv1 /
// conv2 of the BasicBlocks), fused with BatchNorm, residual and ReLU, on the fp32 MFMA pipe.
//
// Why: the encoder is bound by the fp32 MFMA rate (157 TF/s peak, 136 TF/s sustained on random data) and the direct
// implicit-GEMM kernel (conv_pad.hip) already runs at 120-131 TF/s -- the only large lever left is doing fewer
// multiplications.  F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input patch with 16 multiplications per
// (cin, cout) instead of 36: 2.25 x fewer MFMA FLOPs for 62 % of the encoder's arithmetic.
//
//   V = B^T d B   (input transform, 4x4 patch d, per channel: 32 additions)
//   M_p[tile, cout] = sum_cin V_p[tile, cin] * U_p[cin, cout]        p = 16 positions: 16 independent GEMMs on the MFMA pipe
//   Y = A^T M A   (output transform: 24 additions per (tile, cout)), then scale * Y + shift (+ residual) (ReLU)
//   U = G g G^T is precomputed on the host when the weights are prepared (resnet.py).
//
// Work item = one 8 x 8 block of tiles of one image (16 x 16 output pixels, an 18 x 18 input window) x 64 output channels.
// Workgroup = 256 threads = 4 waves as 2 (tile groups of 32) x 2 (cout groups of 32): a wave keeps 16 accumulators of 32x32
// (256 accumulator registers), one wave per SIMD; the grid is persistent (one workgroup per CU walks its items).
// K = Cin is streamed in chunks of 8 channels through LDS (160 KiB: 2 x 32 KiB of transformed input, 2 x 32 KiB of
// transformed filters, a ring of three 10 KiB raw input windows):
//   * the raw 18 x 18 x 8 window of chunk c + 3 arrives by LDS-DMA (each input pixel once per workgroup -- the 4x4 patches of
//     neighbouring tiles overlap by half -- instead of 3.2 times through per-thread loads);
//   * thread (tile, channel pair) reads its 16 patch pixels of chunk c + 1 from that window, transforms them (64 additions) and
//     writes the 16 positions to LDS;
//   * the transformed filters of chunk c + 1 (chunk-contiguous in HBM) arrive by LDS-DMA;
//   * the wave multiplies chunk c: per position 2 conflict-free ds_read_b128 and 4 MFMAs (k pairs (j, 4 + j)), the reads of
//     position p + 1 requested before the MFMAs of position p are issued.
// The DMAs of the next item's first chunks are issued before the epilogue of the current one, so the output transform and the
// stores overlap the next prologue's memory latency.  (First version: per-thread 8-byte patch loads and one workgroup per item
// -- 0.10-0.145 ms per layer, bound by the L1 / TA rate of the scattered 32-byte accesses and by un-overlapped pro/epilogues.)
// The activations are halo-padded NHWC frames (conv_pad.hip): every window is in bounds.
// Accuracy: the transforms use only +-1 and +-1/2 -- exact scalings; the result differs from the direct convolution by
// fp32 rounding of a different summation order (<= 1e-5 of the output scale, tests/test_gpu_net.py).
// The summation order depends on the layer only, never on the batch size (sharding invariance).
#include <mutex>
#include <type_traits>

#include "hps_common.h"

namespace hps {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WT = 64;            // tiles per item (8 x 8)
constexpr int WC = 64;            // output channels per item
constexpr int WK = 8;             // input channels per chunk
constexpr int W_OPER = 16 * 2 * 64 * 4;                   // floats of one operand chunk: [16 positions][2 k-quads][64 rows][4]
constexpr int W_WIN = 18 * 18;                            // pixels of the raw input window of an item
constexpr int W_RAW = W_WIN * WK;                         // floats of one raw window chunk (10 368 bytes)
constexpr int W_RAW_PIECES = (W_WIN * 2 + 63) / 64;       // 1 KiB DMA pieces of a raw window chunk (11, the last one partial)

struct WinoGeom {
    int in_row, in_img;           // input frame pitches in floats: (W + 2 ipad) * Cin, (H + 2 ipad) * that
    int out_row, out_img;         // output frame pitches
    int ipad, opad;
    int blocks_x, blocks_img;     // 8 x 8-tile blocks per row / per image
    int Cin, Cout, n_ct, relu;
    int items;                    // B * blocks_img * n_ct
    unsigned magic_ct, magic_img, magic_x;
};

__device__ __forceinline__ unsigned wino_div(unsigned n, unsigned d, unsigned magic) {
    unsigned q = __umulhi(n, magic);
    if (n - q * d >= d) ++q;
    return q;
}

// AB: profiling ablations (compile-time; the product instantiates AB = 0 only): see hps_dev_conv3x3_winograd
template <int AB>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ residual, float* __restrict__ y,
                                                           const WinoGeom g) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // sA[2][W_OPER] | sB[2][W_OPER] | raw[3][W_RAW]
    float* sA = smem;
    float* sB = smem + 2 * W_OPER;
    float* sR = smem + 4 * W_OPER;
    constexpr int ab = AB;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, il = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int nchunks = g.Cin / WK;

    // ---- raw-window DMA role: piece q = wave + 4 t covers window entries e = 64 q + lane (pixel e >> 1, 16-byte half e & 1) ----
    unsigned r_off[3];
    bool r_ok[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int e = 64 * (wave + 4 * t) + lane;
        r_ok[t] = (wave + 4 * t) < W_RAW_PIECES && e < 2 * W_WIN;
        const int px = r_ok[t] ? (e >> 1) : 0;
        r_off[t] = (unsigned)(((px / 18) * g.in_row + (px % 18) * g.Cin + (e & 1) * 4) * 4);
        if (ab == 7) r_off[t] = (unsigned)((e & 1) * 16);          // profiling: every lane fetches the same line (no memory-system cost)
    }
    const unsigned lds_r0 = (unsigned)(size_t)(lptr_t)(sR);
    const unsigned lds_b0 = (unsigned)(size_t)(lptr_t)(sB);

    // ---- transform role: thread = (tile (ty, tx) = (tid >> 5, (tid >> 2) & 7), channel pair tid & 3) ----
    const int ltile = tid >> 2, cp = tid & 3;
    const int w_slot = ((2 * (ltile >> 3)) * 18 + 2 * (ltile & 7)) * WK + cp * 2;      // float offset of patch pixel (0,0) in a raw window
    const int a_slot = ((cp >> 1) * 64 + ltile) * 4 + (cp & 1) * 2;                     // ... of the (tile, pair) slot of position 0 in sA

    const float* fa = sA + (kl * 64 + wm * 32 + il) * 4;          // this lane's fragment slot of position 0, buffer 0
    const float* fb = sB + (kl * 64 + wn * 32 + il) * 4;

    // per-item state (wave-uniform)
    const float* x_item = nullptr;     // input window origin of the item, chunk 0
    const float* u_item = nullptr;     // transformed filters of the item's cout tile, chunk 0
    auto locate = [&](int item, int& ct, size_t& out_base) {
        const unsigned blk = wino_div((unsigned)item, (unsigned)g.n_ct, g.magic_ct);
        ct = item - blk * g.n_ct;
        const unsigned b = wino_div(blk, (unsigned)g.blocks_img, g.magic_img), rem = blk - b * g.blocks_img;
        const unsigned by = wino_div(rem, (unsigned)g.blocks_x, g.magic_x), bx = rem - by * g.blocks_x;
        x_item = x + (size_t)b * g.in_img + (size_t)(16 * by + g.ipad - 1) * g.in_row + (size_t)(16 * bx + g.ipad - 1) * g.Cin;
        if (ab == 8) x_item = x + (size_t)(g.ipad - 1) * g.in_row + (size_t)(g.ipad - 1) * g.Cin;      // profiling: every item reads the first window (cache-hot, same line count)
        u_item = u + (size_t)ct * W_OPER;
        out_base = (size_t)b * g.out_img + (size_t)(16 * by + g.opad) * g.out_row + (size_t)(16 * bx + g.opad) * g.Cout;
    };
    // One DMA instruction at a time: a burst of them stalls the wave at issue while the texture path works through the
    // line requests (a window piece touches 32 lines), and with one wave per SIMD nothing else issues MFMAs meanwhile --
    // the main loop spreads the eleven pieces of an interval over its sixteen positions.
    auto dma_raw_piece = [&](int chunk, int t) {              // window of `chunk` -> ring slot chunk % 3, piece wave + 4 t
        if (ab == 1 || ab == 6) return;
        if (r_ok[t]) lds_dma16(r_off[t], x_item + chunk * WK, lds_r0 + (unsigned)((chunk % 3) * W_RAW * 4 + (wave + 4 * t) * 1024));
    };
    auto dma_filter_piece = [&](int chunk, int q) {           // filters of `chunk` -> sB[chunk & 1]; wave w moves pieces 8 w .. 8 w + 7
        if (ab == 3) return;
        lds_dma16((unsigned)((wave * 8 + q) * 1024 + lane * 16), u_item + (size_t)chunk * g.n_ct * W_OPER,
                  lds_b0 + (unsigned)((chunk & 1) * W_OPER * 4 + (wave * 8 + q) * 1024));
    };
    auto dma_raw = [&](int chunk) {
#pragma unroll
        for (int t = 0; t < 3; ++t) dma_raw_piece(chunk, t);
    };
    auto dma_filters = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < 8; ++q) dma_filter_piece(chunk, q);
    };
    // input transform of one chunk, in two parts so that the LDS round trip of the reads hides under MFMAs:
    //   t_read : the thread's 16 patch pixels (one channel pair) from the raw ring slot chunk % 3
    //   t_write: V = B^T d B (64 additions) and the 16 positions to sA[chunk & 1]
    float2 d[16];
    auto t_read2 = [&](int chunk, int q) {     // patch pixels 2 q and 2 q + 1 (q = 0..7)
        if (ab == 1 || ab == 5) return;
        const float* src = sR + (chunk % 3) * W_RAW + w_slot;
        d[2 * q] = *reinterpret_cast<const float2*>(src + (((2 * q) >> 2) * 18 + ((2 * q) & 3)) * WK);
        d[2 * q + 1] = *reinterpret_cast<const float2*>(src + (((2 * q + 1) >> 2) * 18 + ((2 * q + 1) & 3)) * WK);
    };
    auto t_read = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < 8; ++q) t_read2(chunk, q);
    };
    auto t_write = [&](int chunk) {
        if (ab == 1 || ab == 5) return;
        float* dst = sA + (chunk & 1) * W_OPER + a_slot;
        float2 t[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {          // t = B^T d
            t[0 * 4 + j] = make_float2(d[0 * 4 + j].x - d[2 * 4 + j].x, d[0 * 4 + j].y - d[2 * 4 + j].y);
            t[1 * 4 + j] = make_float2(d[1 * 4 + j].x + d[2 * 4 + j].x, d[1 * 4 + j].y + d[2 * 4 + j].y);
            t[2 * 4 + j] = make_float2(d[2 * 4 + j].x - d[1 * 4 + j].x, d[2 * 4 + j].y - d[1 * 4 + j].y);
            t[3 * 4 + j] = make_float2(d[1 * 4 + j].x - d[3 * 4 + j].x, d[1 * 4 + j].y - d[3 * 4 + j].y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {          // v = t B
            const float2 v0 = make_float2(t[i * 4 + 0].x - t[i * 4 + 2].x, t[i * 4 + 0].y - t[i * 4 + 2].y);
            const float2 v1 = make_float2(t[i * 4 + 1].x + t[i * 4 + 2].x, t[i * 4 + 1].y + t[i * 4 + 2].y);
            const float2 v2 = make_float2(t[i * 4 + 2].x - t[i * 4 + 1].x, t[i * 4 + 2].y - t[i * 4 + 1].y);
            const float2 v3 = make_float2(t[i * 4 + 1].x - t[i * 4 + 3].x, t[i * 4 + 1].y - t[i * 4 + 3].y);
            *reinterpret_cast<float2*>(dst + (i * 4 + 0) * 512) = v0;
            *reinterpret_cast<float2*>(dst + (i * 4 + 1) * 512) = v1;
            *reinterpret_cast<float2*>(dst + (i * 4 + 2) * 512) = v2;
            *reinterpret_cast<float2*>(dst + (i * 4 + 3) * 512) = v3;
        }
    };
    auto prologue_dma = [&]() {                               // first DMAs of an item: windows 0..2 and filters 0
        dma_filters(0);
        dma_raw(0);
        if (nchunks > 1) dma_raw(1);
        if (nchunks > 2) dma_raw(2);
    };

    int item = blockIdx.x;
    if (item >= g.items) return;
    int ct;
    size_t out_base;
    locate(item, ct, out_base);
    prologue_dma();

    for (;;) {
        f32x16 acc[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of windows 0..2 and filters 0 (and the last stores)
        __syncthreads();                                     // ... everyone's
        t_read(0);
        t_write(0);
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c > 0) {
                // filters of chunk c were issued in iteration c - 1 BEFORE window c + 2: with in-order returns, all but the
                // wave's own pieces of that window (needed one iteration later) must have landed
                // (waves 0-2 move three pieces of a window, wave 3 two: the count is wave-uniform)
                if (c + 2 < nchunks && ab != 1 && ab != 6) {
                    if (wave < 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                                 // operand chunk c complete; sA / sB [buf ^ 1] and ring slot c % 3 are free
            const bool more = c + 1 < nchunks;
            // one wave per SIMD: a latency is hidden only by this wave's own MFMAs.  Order of the interval: first fragments, the
            // DMAs, positions 0-8 with two patch pixels of chunk c + 1 requested per position, transform + store chunk c + 1 (VALU:
            // the only part that does not overlap), positions 9-15.
            const float* pa = fa + buf * W_OPER;
            const float* pb = fb + buf * W_OPER;
            float4 a4[2], b4[2];
            a4[0] = *reinterpret_cast<const float4*>(pa);
            b4[0] = *reinterpret_cast<const float4*>(pb);
            if (more && ab == 9) {                           // profiling: the burst this kernel used to issue
                dma_filters(c + 1);
                if (c + 3 < nchunks) dma_raw(c + 3);
            }
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (more && ab != 9) {                       // filters first, then the window: the vmcnt above counts on that order
                    if (p < 8) dma_filter_piece(c + 1, p);
                    else if (p >= 9 && p < 12 && c + 3 < nchunks) dma_raw_piece(c + 3, p - 9);
                }
                if (p < 15) {
                    a4[(p + 1) & 1] = *reinterpret_cast<const float4*>(pa + (p + 1) * 512);
                    b4[(p + 1) & 1] = *reinterpret_cast<const float4*>(pb + (p + 1) * 512);
                }
                if (p < 8 && more) t_read2(c + 1, p);      // two patch pixels per position: never a long LDS queue ahead of a fragment
                __builtin_amdgcn_sched_barrier(0);
                const float4 a = a4[p & 1], b = b4[p & 1];
                if (ab == 2) { acc[p][0] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
                else {
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[p], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (p == 8 && more) {
                    t_write(c + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // ---- the next item's first DMAs go out before this item's epilogue (which touches no LDS) ----
        const int cur_ct = ct;
        const size_t cur_out = out_base;
        const int next = item + gridDim.x;
        __syncthreads();                                     // everyone is done with this item's LDS
        if (next < g.items) {
            locate(next, ct, out_base);
            prologue_dma();
        }

        // ---- output transform Y = A^T M A, BatchNorm, residual, ReLU.  A lane owns one output channel (MFMA column) and 16 tiles
        //      (MFMA rows dr = (r & 3) + 8 (r >> 2), + 4 kl, of the wave's 32 = tile rows 4 wm + (dr >> 3), columns 4 kl + (dr & 3)):
        //      per output pixel a half-wave stores 128 contiguous bytes ----
        if (ab != 4) {
            const int co = cur_ct * WC + wn * 32 + il;
            const float sc = scale[co], sh = shift[co];
            const size_t lane_base = cur_out + (size_t)(8 * wm) * g.out_row + (size_t)(8 * kl) * g.Cout + co;
            float res[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            auto load_res = [&](int r, float (&dst)[4]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[q] = residual[lane_base + (size_t)(2 * (r >> 2) + (q >> 1)) * g.out_row + (size_t)(2 * (r & 3) + (q & 1)) * g.Cout];
            };
            if (residual) load_res(0, res[0]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (residual && r < 15) load_res(r + 1, res[(r + 1) & 1]);      // one tile ahead of its use
                float s0[4], s1[4];                          // s = A^T M  (2 x 4), A^T = [1 1 1 0; 0 1 -1 -1]
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s0[j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
                    s1[j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
                }
                float yv[4];
                yv[0] = s0[0] + s0[1] + s0[2];
                yv[1] = s0[1] - s0[2] - s0[3];
                yv[2] = s1[0] + s1[1] + s1[2];
                yv[3] = s1[1] - s1[2] - s1[3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = yv[q] * sc + sh;
                    if (residual) v += res[r & 1][q];
                    if (g.relu) v = fmaxf(v, 0.0f);
                    y[lane_base + (size_t)(2 * (r >> 2) + (q >> 1)) * g.out_row + (size_t)(2 * (r & 3) + (q & 1)) * g.Cout] = v;
                }
            }
        } else {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < 16; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[p][r];
            if (t == 12345.678f) y[0] = t;
        }
        if (next >= g.items) break;
        item = next;
    }
}

static unsigned wino_magic(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / d); }

}  // namespace hps

using namespace hps;

static int wino_launch(const float* x, const float* u, const float* scale, const float* shift, const float* residual,
                       float* y, int B, int H, int W, int ipad, int Cin, int Cout, int opad, int relu, int ablate,
                       hps_stream_t stream) {
    if (!x || !u || !scale || !shift || !y) return bad_arg("hps_conv3x3_winograd: null pointer");
    if (B <= 0) return HPS_OK;
    if (H <= 0 || W <= 0 || (H % 16) || (W % 16)) return bad_arg("hps_conv3x3_winograd: H and W must be multiples of 16 (8 x 8 blocks of 2 x 2 tiles)");
    if (Cin <= 0 || Cin % WK != 0 || Cout <= 0 || Cout % WC != 0) return bad_arg("hps_conv3x3_winograd: Cin % 8 == 0 and Cout % 64 == 0 required");
    if (ipad < 1 || opad < 0) return bad_arg("hps_conv3x3_winograd: the input frame needs a halo of at least one pixel");
    if ((size_t)B * (H + 2 * ipad) * (W + 2 * ipad) * Cin * 4 >= 0xffffffffull || (size_t)B * (H + 2 * opad) * (W + 2 * opad) * Cout * 4 >= 0xffffffffull)
        return bad_arg("hps_conv3x3_winograd: tensor exceeds the 32-bit lane offsets");
    WinoGeom g;
    g.in_row = (W + 2 * ipad) * Cin;
    g.in_img = (H + 2 * ipad) * g.in_row;
    g.out_row = (W + 2 * opad) * Cout;
    g.out_img = (H + 2 * opad) * g.out_row;
    g.ipad = ipad; g.opad = opad;
    g.blocks_x = W / 16;
    g.blocks_img = (H / 16) * (W / 16);
    g.Cin = Cin; g.Cout = Cout; g.n_ct = Cout / WC; g.relu = relu;
    g.items = B * g.blocks_img * g.n_ct;
    g.magic_ct = wino_magic((unsigned)g.n_ct);
    g.magic_img = wino_magic((unsigned)g.blocks_img);
    g.magic_x = wino_magic((unsigned)g.blocks_x);
    const size_t lds = (size_t)(4 * W_OPER + 3 * W_RAW) * sizeof(float);           // 162 176 bytes
    // persistent grid: one workgroup per CU (256 on MI355X), items strided over the workgroups
    const dim3 grid((unsigned)(g.items < 256 ? g.items : 256));
    auto launch = [&](auto AB) {
        constexpr int ab = decltype(AB)::value;
        static std::once_flag once;
        std::call_once(once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<ab>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        hipLaunchKernelGGL(conv_wino_kernel<ab>, grid, dim3(256), lds, (hipStream_t)stream, x, u, scale, shift, residual, y, g);
    };
    switch (ablate) {
        case 0: launch(std::integral_constant<int, 0>()); break;
#ifdef HPS_DEV_BUILD
        case 1: launch(std::integral_constant<int, 1>()); break;
        case 2: launch(std::integral_constant<int, 2>()); break;
        case 3: launch(std::integral_constant<int, 3>()); break;
        case 4: launch(std::integral_constant<int, 4>()); break;
        case 5: launch(std::integral_constant<int, 5>()); break;
        case 6: launch(std::integral_constant<int, 6>()); break;
        case 7: launch(std::integral_constant<int, 7>()); break;
        case 8: launch(std::integral_constant<int, 8>()); break;
        case 9: launch(std::integral_constant<int, 9>()); break;
#endif
        default: return bad_arg("hps_conv3x3_winograd: ablate");
    }
    return check_launch("hps_conv3x3_winograd");
}

extern "C" int hps_conv3x3_winograd(const float* x, const float* u, const float* scale, const float* shift, const float* residual,
                                    float* y, int B, int H, int W, int ipad, int Cin, int Cout, int opad, int relu,
                                    hps_stream_t stream) {
    return wino_launch(x, u, scale, shift, residual, y, B, H, W, ipad, Cin, Cout, opad, relu, 0, stream);
}

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_conv3x3_winograd(const float* x, const float* u, const float* scale, const float* shift, const float* residual,
                                        float* y, int B, int H, int W, int ipad, int Cin, int Cout, int opad, int relu, int ablate,
                                        hps_stream_t stream) {
    return wino_launch(x, u, scale, shift, residual, y, B, H, W, ipad, Cin, Cout, opad, relu, ablate, stream);
}
#endif

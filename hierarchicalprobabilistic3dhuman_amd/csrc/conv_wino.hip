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
// Workgroup (the first form, kept in the dev library) = 256 threads = 4 waves as 2 (tile groups of 32) x 2 (cout groups of 32): a
// wave keeps 16 accumulators of 32x32 (256 accumulator registers), one wave per SIMD; the grid is persistent (one workgroup per CU
// walks its items).  The product form (W8, described at the kernel) runs the same item on 8 waves, two per SIMD.
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
    int items;                    // B * blocks_img * n_ct  (quad geometry: ceil(B / 4) * n_ct * ksplit)
    unsigned magic_ct, magic_img, magic_x;
    // quad geometry (8 x 8 maps, four images per item, K split into slices that write raw partial sums)
    int B, ksplit, cps, raw;      // images, K slices, chunks per slice, raw = 1: store Y without BatchNorm / residual / ReLU
    unsigned magic_ks;
};

#ifdef HPS_DEV_BUILD
__device__ unsigned long long g_wino_stamps[256 * 16];      // profiling (ablate 11): shader-clock stamps of workgroup b's second item at [16 b + k]
#define WINO_STAMP(k) do { if (ab == 11 && tid == 0 && nth_item == 1) g_wino_stamps[blockIdx.x * 16 + (k)] = clock64(); } while (0)
#else
#define WINO_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ unsigned wino_div(unsigned n, unsigned d, unsigned magic) {
    unsigned q = __umulhi(n, magic);
    if (n - q * d >= d) ++q;
    return q;
}

// AB: profiling ablations (compile-time; the product instantiates AB = 0 only): see hps_dev_conv3x3_winograd
// QUAD: the item is the 8 x 8 maps of four consecutive images (layer4: an 8 x 8 map has 4 x 4 tiles, a quarter of an item) x 64
// output channels x one slice of K.  The raw ring holds the four 8 x 8 interiors (the halo is not fetched: a patch pixel outside
// its image reads a zero pixel kept in LDS), wave (wm, kl) of the epilogue owns image 2 wm + kl, and the item writes raw partial
// sums Y (the output transform is linear: the slices are added, in slice order, by wino_splitk_epilogue_kernel).
// W8 (16 x 16-block geometry only): the same item, LDS contents and chunk schedule on EIGHT waves -- two per SIMD.  The sixteen
// positions are split by their column j (p = 4 i + j): waves 0-3 multiply j = 0, 1, waves 4-7 j = 2, 3, 128 accumulator registers
// each.  Why: a wave alone on its SIMD issues a VALU instruction every 5 cycles at best (8 when it depends on the previous one), two
// waves together one per 2.5 (tools/valu_dep_probe.hip) -- and the input transform, the fragment reads and the epilogue are a
// quarter of a layer's time.  The transform role becomes thread = (tile, ONE channel); the DMA pieces are dealt over eight waves;
// the output transform needs the other column pair's s[.][2] (resp. s[.][1]): two floats per (tile, channel) cross through the
// then idle sA buffers; every sum is formed in the four-wave kernel's order (identical bits).
// TR (eight waves only; dev library, ablate = 23): the MFMA operands change places -- filters are the A operand, tiles the B operand -- so a
// lane owns ONE tile and sixteen output channels in groups of four consecutive ones: the epilogue loads the residual and stores the output
// as 16-byte vectors (8 + 8 memory instructions per lane instead of 32 + 32 of four bytes).  Every product and every sum is the same:
// identical bits -- but 2-6 % SLOWER per layer (tests/dev/wino8_check.py): a 16-byte store of 64 lanes touches 32 lines of 128 bytes where
// a 4-byte store of a lane-per-channel wave writes two whole lines.
template <int AB, bool QUAD, bool W8 = false, bool TR = false>
__global__ __launch_bounds__(W8 ? 512 : 256, 1) void conv_wino_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ residual, float* __restrict__ y,
                                                           const WinoGeom g) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // sA[2][W_OPER] | sB[2][W_OPER] | raw[3][W_RAW]
    float* sA = smem;
    float* sB = smem + 2 * W_OPER;
    float* sR = smem + 4 * W_OPER;
    constexpr int ab = AB;

    constexpr int NW = W8 ? 8 : 4;                                // waves per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, il = lane & 31;
    const int pg = W8 ? wave >> 2 : 0;                            // W8: the wave's column pair of positions (j = 2 pg, 2 pg + 1)
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    const int nchunks = QUAD ? g.cps : g.Cin / WK;
    constexpr int RP = QUAD ? (W8 ? 1 : 2) : (W8 ? 2 : 3);        // raw DMA pieces per wave and chunk (quad: 8 pieces = 256 pixels; else 11)

    // ---- raw-window DMA role: piece q = wave + 4 t covers window entries e = 64 q + lane (pixel e >> 1, 16-byte half e & 1) ----
    unsigned r_off[3];             // byte offset of the lane's 16 bytes from the item's origin (quad: without the image term)
    int r_img[3];                  // quad: image of the item (0..3) the lane's pixel belongs to
    bool r_ok[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int e = 64 * (wave + NW * t) + lane;
        if (QUAD) {
            const int px = e >> 1;                                // [image][y][x]: 4 x 8 x 8
            r_ok[t] = t < RP;
            r_img[t] = (px >> 6) & 3;
            r_off[t] = (unsigned)((((px >> 3) & 7) * g.in_row + (px & 7) * g.Cin + (e & 1) * 4) * 4);
        } else {
            r_ok[t] = t < RP && (wave + NW * t) < W_RAW_PIECES && e < 2 * W_WIN;
            const int px = r_ok[t] ? (e >> 1) : 0;
            r_img[t] = 0;
            r_off[t] = (unsigned)(((px / 18) * g.in_row + (px % 18) * g.Cin + (e & 1) * 4) * 4);
            if (ab == 7) r_off[t] = (unsigned)((e & 1) * 16);      // profiling: every lane fetches the same line (no memory-system cost)
        }
    }
    unsigned r_cur[3] = {r_off[0], r_off[1], r_off[2]};           // ... of the current item (quad: + the clamped image term)
    const unsigned lds_r0 = (unsigned)(size_t)(lptr_t)(sR);
    const unsigned lds_b0 = (unsigned)(size_t)(lptr_t)(sB);

    // ---- transform role: thread = (tile (ty, tx) = (tid >> 5, (tid >> 2) & 7), channel pair tid & 3) ----
    const int ltile = W8 ? tid >> 3 : tid >> 2, cp = W8 ? (tid & 7) >> 1 : tid & 3;
    const int ce = W8 ? tid & 1 : 0;                              // W8: the thread's channel of the pair (it transforms ONE channel)
    // float offset of patch pixel (0,0) in a raw window (quad: image (ty >> 2, tx >> 2), pixel (2 ly - 1, 2 lx - 1) of its 8 x 8
    // interior -- outside the interior for edge tiles, whose edge pixels read the zero pixel instead)
    const int t_ly = (ltile >> 3) & 3, t_lx = ltile & 3;
    const int w_slot = QUAD ? (((((ltile >> 5) * 2 + ((ltile >> 2) & 1)) * 64 + (2 * t_ly - 1) * 8 + (2 * t_lx - 1)) * WK) + cp * 2 + ce)
                            : ((2 * (ltile >> 3)) * 18 + 2 * (ltile & 7)) * WK + cp * 2 + ce;
    const bool e_top = t_ly == 0, e_bot = t_ly == 3, e_left = t_lx == 0, e_right = t_lx == 3;
    constexpr int W_ZERO = 4 * 64 * WK;                           // quad: a zero pixel sits behind each slot's 256 pixels
    // quad: float offsets (from the ring's start, slot 0) of the thread's 16 patch pixels, halo pixels redirected to the zero pixel,
    // two 16-bit offsets per register.  (Sixteen separate addresses -- what hipcc made of a select at each read, hoisted out of the
    // chunk loop -- did not fit beside 256 accumulators: they went to scratch, and every patch read became a scratch reload behind
    // s_waitcnt vmcnt(0), which also drained the DMAs in flight.)
    unsigned pk_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (QUAD) {
        if (tid < 3 * WK) sR[(tid / WK) * W_RAW + W_ZERO + tid % WK] = 0.0f;     // a zero pixel behind the 256 pixels of every slot (visible after the first barrier of the item loop)
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int i = n >> 2, j = n & 3;
            const bool halo = (i == 0 && e_top) || (i == 3 && e_bot) || (j == 0 && e_left) || (j == 3 && e_right);
            const unsigned off = (unsigned)(halo ? W_ZERO + cp * 2 + ce : w_slot + (i * 8 + j) * WK);
            pk_off[n >> 1] |= off << (16 * (n & 1));
        }
    }
    const int a_slot = ((cp >> 1) * 64 + ltile) * 4 + (cp & 1) * 2 + ce;                // ... of the (tile, pair) slot of position 0 in sA

    const float* fa = sA + (kl * 64 + wm * 32 + il) * 4;          // this lane's fragment slot of position 0, buffer 0
    const float* fb = sB + (kl * 64 + wn * 32 + il) * 4;

    // per-item state (wave-uniform)
    const float* x_item = nullptr;     // input window origin of the item, chunk 0
    const float* u_item = nullptr;     // transformed filters of the item's cout tile, chunk 0
    int nimg = 4;                      // quad: images of the item that exist (the last item of a batch may hold fewer)
    auto locate = [&](int item, int& ct, size_t& out_base) {
        if (QUAD) {
            const unsigned t = wino_div((unsigned)item, (unsigned)g.ksplit, g.magic_ks), ks = item - t * g.ksplit;
            const unsigned qd = wino_div(t, (unsigned)g.n_ct, g.magic_ct);
            ct = t - qd * g.n_ct;
            const int b0 = 4 * qd;
            nimg = min(4, g.B - b0);
            x_item = x + (size_t)b0 * g.in_img + (size_t)g.ipad * g.in_row + (size_t)g.ipad * g.Cin + (size_t)ks * g.cps * WK;
            u_item = u + ((size_t)ks * g.cps * g.n_ct + ct) * W_OPER;
            out_base = (size_t)ks * g.B * g.out_img + (size_t)b0 * g.out_img + (size_t)g.opad * g.out_row + (size_t)g.opad * g.Cout;
#pragma unroll
            for (int t2 = 0; t2 < RP; ++t2) r_cur[t2] = r_off[t2] + (unsigned)(min(r_img[t2], nimg - 1) * g.in_img * 4);
            return;
        }
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
        if (t < RP && r_ok[t]) lds_dma16(r_cur[t], x_item + chunk * WK, lds_r0 + (unsigned)((chunk % 3) * W_RAW * 4 + (wave + NW * t) * 1024));
    };
    constexpr int FP = 32 / NW;                               // filter pieces per wave and chunk
    auto dma_filter_piece = [&](int chunk, int q) {           // filters of `chunk` -> sB[chunk & 1]; wave w moves pieces FP w .. FP w + FP - 1
        if (ab == 3) return;
        lds_dma16((unsigned)((wave * FP + q) * 1024 + lane * 16), u_item + (size_t)chunk * g.n_ct * W_OPER,
                  lds_b0 + (unsigned)((chunk & 1) * W_OPER * 4 + (wave * FP + q) * 1024));
    };
    auto dma_raw = [&](int chunk) {
#pragma unroll
        for (int t = 0; t < 3; ++t) dma_raw_piece(chunk, t);
    };
    auto dma_filters = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < FP; ++q) dma_filter_piece(chunk, q);
    };
    // input transform of one chunk, in two parts so that the LDS round trip of the reads hides under MFMAs:
    //   t_read : the thread's 16 patch pixels (one channel pair) from the raw ring slot chunk % 3
    //   t_write: V = B^T d B (64 additions) and the 16 positions to sA[chunk & 1]
    float2 d[16];
    auto t_read2 = [&](int chunk, int q) {     // patch pixels 2 q and 2 q + 1 (q = 0..7)
        if (ab == 1 || ab == 5) return;
        const float* src = sR + (chunk % 3) * W_RAW + w_slot;
        if (QUAD) {
            unsigned pk = pk_off[q];
            asm volatile("" : "+v"(pk));                          // keep the unpacking here (see pk_off)
            const float* slot = sR + (chunk % 3) * W_RAW;
            if (W8) {
                d[2 * q].x = slot[pk & 0xffffu];
                d[2 * q + 1].x = slot[pk >> 16];
                return;
            }
            d[2 * q] = *reinterpret_cast<const float2*>(slot + (pk & 0xffffu));
            d[2 * q + 1] = *reinterpret_cast<const float2*>(slot + (pk >> 16));
            return;
        }
        if (W8) {                                                 // one channel: the .y halves stay unused
            d[2 * q].x = src[(((2 * q) >> 2) * 18 + ((2 * q) & 3)) * WK];
            d[2 * q + 1].x = src[(((2 * q + 1) >> 2) * 18 + ((2 * q + 1) & 3)) * WK];
            return;
        }
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
        if (W8) {                                                 // the same sums for the thread's one channel
            float t1[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t1[0 * 4 + j] = d[0 * 4 + j].x - d[2 * 4 + j].x;
                t1[1 * 4 + j] = d[1 * 4 + j].x + d[2 * 4 + j].x;
                t1[2 * 4 + j] = d[2 * 4 + j].x - d[1 * 4 + j].x;
                t1[3 * 4 + j] = d[1 * 4 + j].x - d[3 * 4 + j].x;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float o0 = t1[i * 4 + 0] - t1[i * 4 + 2], o1 = t1[i * 4 + 1] + t1[i * 4 + 2];
                float o2 = t1[i * 4 + 2] - t1[i * 4 + 1], o3 = t1[i * 4 + 1] - t1[i * 4 + 3];
                if (ab == 14) {                               // timing only: the VALU work of splitting every transformed value
                    auto split_pair = [](float& p, float& q) {   // into three bf16 pieces, two values per v_cvt_pk_bf16_f32 (results garbage)
                        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        f2 v = {p, q};
                        const bf2 h1 = __builtin_convertvector(v, bf2);
                        const f2 r1 = v - __builtin_convertvector(h1, f2);
                        const bf2 h2 = __builtin_convertvector(r1, bf2);
                        const f2 r2 = r1 - __builtin_convertvector(h2, f2);
                        const bf2 h3 = __builtin_convertvector(r2, bf2);
                        p = __uint_as_float(__builtin_bit_cast(unsigned, h1) ^ __builtin_bit_cast(unsigned, h3));
                        q = __uint_as_float(__builtin_bit_cast(unsigned, h2));
                    };
                    split_pair(o0, o1);
                    split_pair(o2, o3);
                }
                dst[(i * 4 + 0) * 512] = o0;
                dst[(i * 4 + 1) * 512] = o1;
                dst[(i * 4 + 2) * 512] = o2;
                dst[(i * 4 + 3) * 512] = o3;
            }
            return;
        }
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
    int nth_item = 0;
    int ct;
    size_t out_base;
    locate(item, ct, out_base);
    prologue_dma();

    for (;;) {
        constexpr int NP = W8 ? 8 : 16;                      // positions per wave (W8: acc[2 i + jj] is position 4 i + 2 pg + jj)
        f32x16 acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of windows 0..2 and filters 0 (and the last stores)
        WINO_STAMP(6);
        __syncthreads();                                     // ... everyone's
        WINO_STAMP(7);
        t_read(0);
        t_write(0);
        WINO_STAMP(8);
        for (int c = 0; c < nchunks; ++c) {
            if (c == 1) WINO_STAMP(9);
            if (c == 2) WINO_STAMP(10);
            const int buf = c & 1;
            if (c > 0) {
                // filters of chunk c were issued in iteration c - 1 BEFORE window c + 2: with in-order returns, all but the
                // wave's own pieces of that window (needed one iteration later) must have landed
                // (waves 0-2 move three pieces of a window, wave 3 two: the count is wave-uniform)
                if (c + 2 < nchunks && ab != 1 && ab != 6) {
                    if (W8) {                                // eleven window pieces over eight waves: waves 0-2 move two, the others one (quad: eight pieces, one each)
                        if (!QUAD && wave < 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    } else if (!QUAD && wave < 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (ab != 10) __syncthreads();                   // operand chunk c complete; sA / sB [buf ^ 1] and ring slot c % 3 are free (10: profiling, races)
            const bool more = c + 1 < nchunks;
            // one wave per SIMD: a latency is hidden only by this wave's own MFMAs.  Order of the interval: first fragments, the
            // DMAs, positions 0-8 with two patch pixels of chunk c + 1 requested per position, transform + store chunk c + 1 (VALU:
            // the only part that does not overlap), positions 9-15.
            const float* pa = fa + buf * W_OPER + (W8 ? pg * 2 * 512 : 0);      // W8: position 2 pg is the wave's first
            const float* pb = fb + buf * W_OPER + (W8 ? pg * 2 * 512 : 0);
            float4 a4[2], b4[2];
            a4[0] = *reinterpret_cast<const float4*>(pa);
            b4[0] = *reinterpret_cast<const float4*>(pb);
            if (W8) {
                // eight positions per wave and interval (the other eight run on the wave that shares the SIMD): four filter pieces,
                // then the window pieces; four patch pixels per position in positions 0-3, the transform + its stores in position 5
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) {             // pp = 2 i + jj: position 4 i + 2 pg + jj (pa / pb start at position 2 pg)
                    const int pnext = (4 * ((pp + 1) >> 1) + ((pp + 1) & 1)) * 512;
                    if (more) {
                        if (pp < FP) dma_filter_piece(c + 1, pp);
                        else if (pp >= 5 && pp < 5 + RP && c + 3 < nchunks) dma_raw_piece(c + 3, pp - 5);
                    }
                    if (pp < 7) {
                        a4[(pp + 1) & 1] = *reinterpret_cast<const float4*>(pa + pnext);
                        b4[(pp + 1) & 1] = *reinterpret_cast<const float4*>(pb + pnext);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float4 a = a4[pp & 1], b = b4[pp & 1];
                    if (ab == 12) __builtin_amdgcn_s_setprio(3);
                    if (ab == 13 || ab == 14) {               // timing only: the K = 8 of this position as bf16x3 piece products = 3 x v_mfma_f32_32x32x16_bf16
                        typedef __bf16 bf8 __attribute__((ext_vector_type(8)));          // (six per 16 k) on whatever bits the fp32 fragments hold
                        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc[pp], 0, 0, 0);
                        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, b), __builtin_bit_cast(bf8, a), acc[pp], 0, 0, 0);
                    } else {
                    acc[pp] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc[pp], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[pp], 0, 0, 0);
                    acc[pp] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc[pp], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[pp], 0, 0, 0);
                    }
                    if (ab == 12) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pp < 4 && more) { t_read2(c + 1, 2 * pp); t_read2(c + 1, 2 * pp + 1); }
                    if (pp == 5 && more) t_write(c + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ab == 12) __builtin_amdgcn_s_setprio(3);
                    if (ab == 13 || ab == 14) {
                        typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
                        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, a), acc[pp], 0, 0, 0);
                    } else {
                    acc[pp] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc[pp], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[pp], 0, 0, 0);
                    acc[pp] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc[pp], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[pp], 0, 0, 0);
                    }
                    if (ab == 12) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
            if (more && ab == 9) {                           // profiling: the burst this kernel used to issue
                dma_filters(c + 1);
                if (c + 3 < nchunks) dma_raw(c + 3);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (more && ab != 9) {                       // filters first, then the window: the vmcnt above counts on that order
                    if (p < 8) dma_filter_piece(c + 1, p);
                    else if (p >= 9 && p < 9 + RP && c + 3 < nchunks) dma_raw_piece(c + 3, p - 9);
                }
                if (p < 15) {
                    a4[(p + 1) & 1] = *reinterpret_cast<const float4*>(pa + (p + 1) * 512);
                    b4[(p + 1) & 1] = *reinterpret_cast<const float4*>(pb + (p + 1) * 512);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 a = a4[p & 1], b = b4[p & 1];
                if (ab == 2) {
                    if (p < 8 && more) t_read2(c + 1, p);
                    acc[p][0] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
                } else {
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[p], 0, 0, 0);
                    // two patch pixels per position, requested in the MIDDLE of the position's MFMAs: hipcc's wait in front of the
                    // next position's MFMAs allows only the two newest LDS reads to be outstanding, so a patch read issued together
                    // with the next fragments made every position wait out a fresh LDS round trip
                    __builtin_amdgcn_sched_barrier(0);
                    if (p < 8 && more) t_read2(c + 1, p);
                    if (p == 8 && more) t_write(c + 1);      // ... and the transform + its 16 LDS stores likewise (position 8)
                    __builtin_amdgcn_sched_barrier(0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[p], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ab == 2 && p == 8 && more) {
                    t_write(c + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // ---- the next item's first DMAs go out before this item's epilogue (which touches no LDS) ----
        const int cur_ct = ct, cur_nimg = nimg;
        const size_t cur_out = out_base;
        const int next = item + gridDim.x;
        ++nth_item;                                          // (stamps 0-5 belong to the end of the first item, 6-10 to the start of the second)
        WINO_STAMP(0);
        __syncthreads();                                     // everyone is done with this item's LDS
        WINO_STAMP(1);
        if (next < g.items) {
            locate(next, ct, out_base);
            prologue_dma();
        }
        WINO_STAMP(2);

        // ---- output transform Y = A^T M A, BatchNorm, residual, ReLU.  A lane owns one output channel (MFMA column) and 16 tiles
        //      (MFMA rows dr = (r & 3) + 8 (r >> 2), + 4 kl, of the wave's 32 = tile rows 4 wm + (dr >> 3), columns 4 kl + (dr & 3)):
        //      per output pixel a half-wave stores 128 contiguous bytes ----
        if (W8 && ab == 4) {                                 // profiling: no epilogue
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[p][r];
            if (t == 12345.678f) y[0] = t;
        } else if (W8 && TR) {
            // The lane's tile is (4 wm + (il >> 3), il & 7) of the item, its channels 8 g + 4 kl + (r & 3) of the wave's 32 for
            // register r = 4 g + (r & 3).  Row transform, exchange of one column with the partner wave and column transform exactly
            // as in the other eight-wave form (per register: the sums do not care which of tile / channel a register stands for).
            const int t_y = il >> 3, t_x = il & 7;
            const int cb = cur_ct * WC + wn * 32 + 4 * kl;
            const size_t lane_base = (QUAD ? cur_out + (size_t)(2 * wm + (t_x >> 2)) * g.out_img + (size_t)(2 * t_y) * g.out_row + (size_t)(2 * (t_x & 3) + pg) * g.Cout
                                           : cur_out + (size_t)(2 * (4 * wm + t_y)) * g.out_row + (size_t)(2 * t_x + pg) * g.Cout) + cb;
            const bool store_ok = !QUAD || 2 * wm + (t_x >> 2) < cur_nimg;
            float2* xch = reinterpret_cast<float2*>(sA);
            float k0[2][16], k1[2][16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    k0[jj][r] = acc[0 + jj][r] + acc[2 + jj][r] + acc[4 + jj][r];
                    k1[jj][r] = acc[2 + jj][r] - acc[4 + jj][r] - acc[6 + jj][r];
                }
            const int w4 = wave & 3;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 e = pg == 0 ? make_float2(k0[1][r], k1[1][r]) : make_float2(k0[0][r], k1[0][r]);
                xch[((pg * 16 + r) * 4 + w4) * 64 + lane] = e;
            }
            constexpr bool finish = !QUAD;                   // the 8 x 8 geometry always writes raw partial sums (wino_launch: g.raw = 1)
            float4 res[2][4], sc4[4], sh4[4];
            if (finish) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sc4[q] = *reinterpret_cast<const float4*>(scale + cb + 8 * q);
                    sh4[q] = *reinterpret_cast<const float4*>(shift + cb + 8 * q);
                }
                if (residual) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            res[a][q] = *reinterpret_cast<const float4*>(residual + lane_base + (size_t)a * g.out_row + 8 * q);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float out[2][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    const float2 o = xch[(((1 - pg) * 16 + r) * 4 + w4) * 64 + lane];
                    float yv[2];
                    if (pg == 0) {
                        yv[0] = k0[0][r] + k0[1][r] + o.x;
                        yv[1] = k1[0][r] + k1[1][r] + o.y;
                    } else {
                        yv[0] = o.x - k0[0][r] - k0[1][r];
                        yv[1] = o.y - k1[0][r] - k1[1][r];
                    }
                    const float sc = e == 0 ? sc4[q].x : e == 1 ? sc4[q].y : e == 2 ? sc4[q].z : sc4[q].w;
                    const float sh = e == 0 ? sh4[q].x : e == 1 ? sh4[q].y : e == 2 ? sh4[q].z : sh4[q].w;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        float v = yv[a];
                        if (finish) {
                            v = v * sc + sh;
                            if (residual) v += e == 0 ? res[a][q].x : e == 1 ? res[a][q].y : e == 2 ? res[a][q].z : res[a][q].w;
                            if (g.relu) v = fmaxf(v, 0.0f);
                        }
                        out[a][e] = v;
                    }
                }
                if (store_ok) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        *reinterpret_cast<float4*>(y + lane_base + (size_t)a * g.out_row + 8 * q) = make_float4(out[a][0], out[a][1], out[a][2], out[a][3]);
                }
            }
        } else if (W8) {
            // Row transform s = A^T M for the wave's two columns jj (identical sums, in the four-wave kernel's order), then the column
            // transform needs ONE column of the other pair: y(a, 0) = (s_a[0] + s_a[1]) + s_a[2] is formed by the waves of columns
            // 0-1 with s_a[2] from their partners, y(a, 1) = (s_a[1] - s_a[2]) - s_a[3] by the waves of columns 2-3 with s_a[1].
            // The exchange runs through sA (idle: the next item's first DMAs fill sB[0] and the raw ring).
            const int co = cur_ct * WC + wn * 32 + il;
            const float sc = scale[co], sh = shift[co];
            // quad: wave (wm, kl)'s 4 x 4 tiles are the whole map of image 2 wm + kl
            const size_t lane_base = (QUAD ? cur_out + (size_t)(2 * wm + kl) * g.out_img + co
                                           : cur_out + (size_t)(8 * wm) * g.out_row + (size_t)(8 * kl) * g.Cout + co) + (size_t)pg * g.Cout;
            const bool store_ok = !QUAD || 2 * wm + kl < cur_nimg;
            float2* xch = reinterpret_cast<float2*>(sA);
            float k0[2][16], k1[2][16];                      // s_0[jj], s_1[jj] of the wave's columns, per tile register r
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    k0[jj][r] = acc[0 + jj][r] + acc[2 + jj][r] + acc[4 + jj][r];
                    k1[jj][r] = acc[2 + jj][r] - acc[4 + jj][r] - acc[6 + jj][r];
                }
            const int w4 = wave & 3;
#pragma unroll
            for (int r = 0; r < 16; ++r) {                   // columns 0-1 export column 1, columns 2-3 export column 2 (their jj = 0)
                const float2 e = pg == 0 ? make_float2(k0[1][r], k1[1][r]) : make_float2(k0[0][r], k1[0][r]);
                xch[((pg * 16 + r) * 4 + w4) * 64 + lane] = e;
            }
            float res[16][2];
            if (residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        res[r][a] = residual[lane_base + (size_t)(2 * (r >> 2) + a) * g.out_row + (size_t)(2 * (r & 3)) * g.Cout];
            }
            WINO_STAMP(3);
            __syncthreads();
            WINO_STAMP(4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 o = xch[(((1 - pg) * 16 + r) * 4 + w4) * 64 + lane];
                float yv[2];
                if (pg == 0) {                               // y(a, 0) = (s_a[0] + s_a[1]) + s_a[2]
                    yv[0] = k0[0][r] + k0[1][r] + o.x;
                    yv[1] = k1[0][r] + k1[1][r] + o.y;
                } else {                                     // y(a, 1) = (s_a[1] - s_a[2]) - s_a[3]
                    yv[0] = o.x - k0[0][r] - k0[1][r];
                    yv[1] = o.y - k1[0][r] - k1[1][r];
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v = yv[a];
                    if (!QUAD || !g.raw) {
                        v = v * sc + sh;
                        if (residual) v += res[r][a];
                        if (g.relu) v = fmaxf(v, 0.0f);
                    }
                    if (store_ok) y[lane_base + (size_t)(2 * (r >> 2) + a) * g.out_row + (size_t)(2 * (r & 3)) * g.Cout] = v;
                }
            }
            WINO_STAMP(5);
        } else if (ab != 4) {
            const int co = cur_ct * WC + wn * 32 + il;
            const float sc = scale[co], sh = shift[co];
            // quad: wave (wm, kl)'s 4 x 4 tiles are the whole map of image 2 wm + kl
            const size_t lane_base = QUAD ? cur_out + (size_t)(2 * wm + kl) * g.out_img + co
                                          : cur_out + (size_t)(8 * wm) * g.out_row + (size_t)(8 * kl) * g.Cout + co;
            const bool store_ok = !QUAD || 2 * wm + kl < cur_nimg;
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
                    constexpr int S = W8 ? 0 : 4;            // (W8 never runs this branch; keeps the indices inside its eight accumulators)
                    s0[j] = acc[0 * S + j][r] + acc[1 * S + j][r] + acc[2 * S + j][r];
                    s1[j] = acc[1 * S + j][r] - acc[2 * S + j][r] - acc[3 * S + j][r];
                }
                float yv[4];
                yv[0] = s0[0] + s0[1] + s0[2];
                yv[1] = s0[1] - s0[2] - s0[3];
                yv[2] = s1[0] + s1[1] + s1[2];
                yv[3] = s1[1] - s1[2] - s1[3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = yv[q];
                    if (!QUAD || !g.raw) {
                        v = v * sc + sh;
                        if (residual) v += res[r & 1][q];
                        if (g.relu) v = fmaxf(v, 0.0f);
                    }
                    if (store_ok) y[lane_base + (size_t)(2 * (r >> 2) + (q >> 1)) * g.out_row + (size_t)(2 * (r & 3) + (q & 1)) * g.Cout] = v;
                }
            }
        } else {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[p][r];
            if (t == 12345.678f) y[0] = t;
        }
        if (next >= g.items) break;
        item = next;
    }
}

// second pass of the quad geometry: y = act(scale * (sum of the K slices, in slice order) + shift + residual) into the padded
// 8 x 8 output frames; thread per four channels
__global__ __launch_bounds__(256) void wino_splitk_epilogue_kernel(const float* __restrict__ partial, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ residual,
                                                                   float* __restrict__ y, int total4, int ksplit, int Cout, int opad,
                                                                   int relu) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 acc = reinterpret_cast<const float4*>(partial)[i];
    for (int k = 1; k < ksplit; ++k) {
        const float4 v = reinterpret_cast<const float4*>(partial)[(size_t)k * total4 + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const unsigned e = (unsigned)i * 4u;
    const unsigned m = e / (unsigned)Cout, co = e - m * Cout;               // m = (image * 8 + row) * 8 + column
    const unsigned b = m >> 6, py = (m >> 3) & 7, px = m & 7, fw = 8 + 2 * opad;
    const size_t o = ((size_t)(b * fw + py + opad) * fw + px + opad) * Cout + co;
    const float4 sc = *reinterpret_cast<const float4*>(scale + co), sh = *reinterpret_cast<const float4*>(shift + co);
    float4 v = make_float4(acc.x * sc.x + sh.x, acc.y * sc.y + sh.y, acc.z * sc.z + sh.z, acc.w * sc.w + sh.w);
    if (residual) {
        const float4 r = *reinterpret_cast<const float4*>(residual + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(y + o) = v;
}

#ifdef HPS_DEV_BUILD
// ------------------------------------------------------------------------------------------------------------------------------------
// The half-item form (dev library only; experiment of round 5, measured 12-16 % SLOWER than the product -- tests/dev/wino_half_check.py): the same arithmetic on work items of 4 x 8 tiles (8 x 16 output pixels) x 64
// output channels, FOUR waves per workgroup and 66 KB of LDS, so that TWO workgroups share a CU and one's barriers, DMA waits and epilogue
// run under the other's MFMAs.  K is streamed in half-chunks of four channels -- {0, 4, 1, 5} then {2, 6, 3, 7} of an eight-channel chunk,
// i.e. the k pairs (0, 4), (1, 5), (2, 6), (3, 7) in the product kernel's order: identical bits.  Waves = (column pair pg of the sixteen
// positions) x (32-channel half wn): eight positions, 128 accumulator registers each; the raw windows stay eight channels wide (a ring of
// three 10 x 18-pixel windows, each serving two half-chunks), the filters come in a second packing u4[half-chunk][cout tile][16 positions]
// [64 channels][kl][slot] (resnet.py), the input transform is split by output rows between the thread halves (rows 0-1: patch rows 0-2,
// rows 2-3: patch rows 1-3), and the epilogue is the eight-wave form's with the exchange buffer in sA + the idle filter buffer.
// What it showed: the epilogue does hide (4.1 -> 1.8 us per layer), but every MFMA now carries twice the LDS and DMA instructions (b64
// fragments, 16 KB of filters per 16 MFMAs and workgroup: ~10 TB/s of L2 -> LDS traffic chip-wide) and the K loop runs at 0.61 of the MFMA
// rate instead of 0.74.
constexpr int H_A = 16 * 32 * 4;              // floats of one transformed-input buffer   [position][tile][kl][slot]   (8 KB)
constexpr int H_B = 16 * 64 * 4;              // floats of one filter buffer              [position][channel][kl][slot] (16 KB)
constexpr int H_WIN = 10 * 18;                // pixels of an item's raw window
constexpr int H_RAW = 6 * 256;                // floats of a ring slot: six 1 KiB DMA pieces (the window is 5 760 bytes)

template <int AB>
__global__ __launch_bounds__(256, 2) void conv_wino_half_kernel(const float* __restrict__ x, const float* __restrict__ u4,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ residual, float* __restrict__ y, const WinoGeom g) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // sB[0] | sA[0] | sA[1] | sB[1] | ring[3]
    float* sB0 = smem;
    float* sA = smem + H_B;
    float* sB1 = smem + H_B + 2 * H_A;
    float* sR = smem + 2 * H_B + 2 * H_A;
    constexpr int ab = AB;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, il = lane & 31;
    const int pg = wave >> 1, wn = wave & 1;
    const int nh = g.Cin / 4, n8 = g.Cin / 8;

    // ---- raw-window DMA role: piece q covers window entries e = 64 q + lane (pixel e >> 1, 16-byte half e & 1); wave w moves piece w in the
    //      first half-step of a window's turn and (w < 2) piece 4 + w in the second ----
    unsigned r_off[2];
    bool r_ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int e = 64 * (wave + 4 * t) + lane;
        r_ok[t] = (t == 0 || wave < 2) && e < 2 * H_WIN;
        const int px = r_ok[t] ? (e >> 1) : 0;
        r_off[t] = (unsigned)(((px / 18) * g.in_row + (px % 18) * g.Cin + (e & 1) * 4) * 4);
    }
    const unsigned lds_r0 = (unsigned)(size_t)(lptr_t)(sR);
    const unsigned lds_b[2] = {(unsigned)(size_t)(lptr_t)(sB0), (unsigned)(size_t)(lptr_t)(sB1)};

    // ---- transform role: thread = (row half hh = tid >> 7, tile (ty, tx) = ((tid >> 5) & 3, (tid >> 2) & 7), channel slot ch = tid & 3 = 2 kl + slot) ----
    const int hh = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int ltile = (tid >> 2) & 31, ch = tid & 3;
    const int w_slot = ((2 * (ltile >> 3) + hh) * 18 + 2 * (ltile & 7)) * 8 + (ch & 1) + 4 * (ch >> 1);    // patch pixel (hh, 0), channel of half-chunk parity 0
    const int a_slot = ltile * 4 + ch + hh * 8 * 128;                                                   // (tile, kl, slot) of position 8 hh in sA

    const float* fa = sA + il * 4 + kl * 2;                        // this lane's fragment of position 0, buffer 0
    const int fb_off = (wn * 32 + il) * 4 + kl * 2;

    const float* x_item = nullptr;
    const float* u_item = nullptr;
    auto locate = [&](int item, int& ct, size_t& out_base) {
        const unsigned hb = wino_div((unsigned)item, (unsigned)g.n_ct, g.magic_ct);       // (image, block, half)
        ct = item - hb * g.n_ct;
        const unsigned blk = hb >> 1, half = hb & 1;
        const unsigned b = wino_div(blk, (unsigned)g.blocks_img, g.magic_img), rem = blk - b * g.blocks_img;
        const unsigned by = wino_div(rem, (unsigned)g.blocks_x, g.magic_x), bx = rem - by * g.blocks_x;
        x_item = x + (size_t)b * g.in_img + (size_t)(16 * by + 8 * half + g.ipad - 1) * g.in_row + (size_t)(16 * bx + g.ipad - 1) * g.Cin;
        u_item = u4 + (size_t)ct * H_B;
        out_base = (size_t)b * g.out_img + (size_t)(16 * by + 8 * half + g.opad) * g.out_row + (size_t)(16 * bx + g.opad) * g.Cout;
    };
    auto dma_raw_piece = [&](int c8, int t) {                  // window of chunk c8 -> ring slot c8 % 3, piece wave + 4 t
        if (r_ok[t]) lds_dma16(r_off[t], x_item + c8 * 8, lds_r0 + (unsigned)((c8 % 3) * H_RAW * 4 + (wave + 4 * t) * 1024));
    };
    auto dma_filter_piece = [&](int h, int q) {                // filters of half-chunk h -> sB[h & 1]; wave w moves pieces 4 w .. 4 w + 3
        lds_dma16((unsigned)((wave * 4 + q) * 1024 + lane * 16), u_item + (size_t)h * g.n_ct * H_B, lds_b[h & 1] + (unsigned)((wave * 4 + q) * 1024));
    };
    float d[12];                                               // the thread's three patch rows x four columns (one channel)
    auto t_read3 = [&](int h, int q) {                         // patch row q (0..2) of the thread's half
        const float* src = sR + (((h >> 1) % 3) * H_RAW) + w_slot + 2 * (h & 1) + q * 18 * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[q * 4 + j] = src[j * 8];
    };
    auto t_write = [&](int h) {
        float* dst = sA + (h & 1) * H_A + a_slot;
        float t[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (hh == 0) {                                     // rows 0, 1 of t = B^T d from patch rows 0, 1, 2
                t[j] = d[0 * 4 + j] - d[2 * 4 + j];
                t[4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
            } else {                                           // rows 2, 3 from patch rows 1, 2, 3 (the thread's local rows 0, 1, 2)
                t[j] = d[1 * 4 + j] - d[0 * 4 + j];
                t[4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            dst[(i * 4 + 0) * 128] = t[i * 4 + 0] - t[i * 4 + 2];
            dst[(i * 4 + 1) * 128] = t[i * 4 + 1] + t[i * 4 + 2];
            dst[(i * 4 + 2) * 128] = t[i * 4 + 2] - t[i * 4 + 1];
            dst[(i * 4 + 3) * 128] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };
    auto prologue_dma = [&]() {                                // first DMAs of an item: filters of half-chunk 0, windows 0..2
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_filter_piece(0, q);
#pragma unroll
        for (int c8 = 0; c8 < 3; ++c8)
            if (c8 < n8) { dma_raw_piece(c8, 0); dma_raw_piece(c8, 1); }
    };

    int item = blockIdx.x;
    if (item >= g.items) return;
    int ct;
    size_t out_base;
    locate(item, ct, out_base);
    prologue_dma();

    for (;;) {
        f32x16 acc[8];                                         // acc[2 i + jj] = position 4 i + 2 pg + jj
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) t_read3(0, q);
        t_write(0);
        for (int h = 0; h < nh; ++h) {
            const int buf = h & 1;
            if (h > 0) {
                // the filters of half-chunk h went out in half-step h - 1 BEFORE that step's window piece (needed three half-steps later)
                const int cfree = (h - 2) >> 1;                // the window whose slot half-step h - 1 refilled (h - 1 >= 1)
                const bool issued = h >= 2 && cfree + 3 < n8 && (((h - 1) & 1) || wave < 2);
                if (issued) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                                   // operands of half-chunk h complete; the other buffers and the ring slot of window (h - 1) / 2 - ... are free
            const bool more = h + 1 < nh;
            const int cnew = ((h - 1) >> 1) + 3;               // h >= 1: window to fetch into the slot that window (h - 1) / 2 left
            const bool fetch = h >= 1 && cnew < n8;
            const float* pa = fa + buf * H_A + pg * 2 * 128;
            const float* pb = (buf ? sB1 : sB0) + fb_off + pg * 2 * 256;
            float2 a2[2], b2[2];
            a2[0] = *reinterpret_cast<const float2*>(pa);
            b2[0] = *reinterpret_cast<const float2*>(pb);
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {                   // pp = 2 i + jj: position 4 i + 2 pg + jj
                const int pn = 4 * ((pp + 1) >> 1) + ((pp + 1) & 1);
                if (more && pp < 4) dma_filter_piece(h + 1, pp);
                if (fetch && pp == 5) dma_raw_piece(cnew, (h - 1) & 1 ? 1 : 0);
                if (pp < 7) {
                    a2[(pp + 1) & 1] = *reinterpret_cast<const float2*>(pa + pn * 128);
                    b2[(pp + 1) & 1] = *reinterpret_cast<const float2*>(pb + pn * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float2 a = a2[pp & 1], b = b2[pp & 1];
                acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[pp], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more && pp < 3) t_read3(h + 1, pp);
                if (more && pp == 5) t_write(h + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[pp], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        const int cur_ct = ct;
        const size_t cur_out = out_base;
        const int next = item + gridDim.x;
        __syncthreads();                                       // everyone is done with this item's LDS
        if (next < g.items) {
            locate(next, ct, out_base);
            prologue_dma();
        }
        if (ab == 4) {                                         // profiling: no epilogue
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[p][r];
            if (t == 12345.678f) y[0] = t;
        } else {
            // the eight-wave form's epilogue with one tile group: a lane owns channel co and the 16 tiles (r >> 2, 4 kl + (r & 3)) of the item
            const int co = cur_ct * WC + wn * 32 + il;
            const float sc = scale[co], sh = shift[co];
            const size_t lane_base = cur_out + (size_t)(8 * kl) * g.Cout + co + (size_t)pg * g.Cout;
            float2* xch = reinterpret_cast<float2*>(sA);       // 32 KB: sA[0], sA[1] and the idle filter buffer sB[1]
            float k0[2][16], k1[2][16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    k0[jj][r] = acc[0 + jj][r] + acc[2 + jj][r] + acc[4 + jj][r];
                    k1[jj][r] = acc[2 + jj][r] - acc[4 + jj][r] - acc[6 + jj][r];
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 e = pg == 0 ? make_float2(k0[1][r], k1[1][r]) : make_float2(k0[0][r], k1[0][r]);
                xch[((pg * 16 + r) * 2 + wn) * 64 + lane] = e;
            }
            float res[16][2];
            if (residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        res[r][a] = residual[lane_base + (size_t)(2 * (r >> 2) + a) * g.out_row + (size_t)(2 * (r & 3)) * g.Cout];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 o = xch[(((1 - pg) * 16 + r) * 2 + wn) * 64 + lane];
                float yv[2];
                if (pg == 0) {
                    yv[0] = k0[0][r] + k0[1][r] + o.x;
                    yv[1] = k1[0][r] + k1[1][r] + o.y;
                } else {
                    yv[0] = o.x - k0[0][r] - k0[1][r];
                    yv[1] = o.y - k1[0][r] - k1[1][r];
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float v = yv[a] * sc + sh;
                    if (residual) v += res[r][a];
                    if (g.relu) v = fmaxf(v, 0.0f);
                    y[lane_base + (size_t)(2 * (r >> 2) + a) * g.out_row + (size_t)(2 * (r & 3)) * g.Cout] = v;
                }
            }
        }
        if (next >= g.items) break;
        item = next;
    }
}
#endif  // HPS_DEV_BUILD

static unsigned wino_magic(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / d); }

}  // namespace hps

using namespace hps;

// K slices of the quad geometry: a rule on the layer only (never on the batch: the summation order of a pixel must not change
// with B).  Four slices when each still has >= 8 chunks (layer4: 64 chunks -> 4 x 16), else one.
#ifdef HPS_DEV_BUILD
static int g_wino_quad_ks = 0;           // dev library only (hps_dev_wino_quad_ksplit): 0 = the product rule
#else
constexpr int g_wino_quad_ks = 0;
#endif
static int wino_quad_ksplit(int Cin) {
    if (g_wino_quad_ks > 0 && (Cin / WK) % g_wino_quad_ks == 0) return g_wino_quad_ks;
    return (Cin / WK) % 4 == 0 && (Cin / WK) / 4 >= 8 ? 4 : 1;
}

extern "C" size_t hps_conv3x3_winograd_workspace(int B, int H, int W, int Cin, int Cout) {
    if (H != 8 || W != 8 || B <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return (size_t)wino_quad_ksplit(Cin) * B * 64 * Cout * sizeof(float);
}

static int wino_launch(const float* x, const float* u, const float* scale, const float* shift, const float* residual,
                       float* y, int B, int H, int W, int ipad, int Cin, int Cout, int opad, int relu, float* splitk_ws, int ablate,
                       hps_stream_t stream) {
    if (!x || !u || !scale || !shift || !y) return bad_arg("hps_conv3x3_winograd: null pointer");
    if (B <= 0) return HPS_OK;
    const bool quad = H == 8 && W == 8;
    if (!quad && (H <= 0 || W <= 0 || (H % 16) || (W % 16)))
        return bad_arg("hps_conv3x3_winograd: H and W must be multiples of 16 (8 x 8 blocks of 2 x 2 tiles), or H = W = 8");
    if (Cin <= 0 || Cin % WK != 0 || Cout <= 0 || Cout % WC != 0) return bad_arg("hps_conv3x3_winograd: Cin % 8 == 0 and Cout % 64 == 0 required");
    if (ipad < 1 || opad < 0) return bad_arg("hps_conv3x3_winograd: the input frame needs a halo of at least one pixel");
    if ((size_t)B * (H + 2 * ipad) * (W + 2 * ipad) * Cin * 4 >= 0xffffffffull || (size_t)B * (H + 2 * opad) * (W + 2 * opad) * Cout * 4 >= 0xffffffffull)
        return bad_arg("hps_conv3x3_winograd: tensor exceeds the 32-bit lane offsets");
    if (quad && !splitk_ws) return bad_arg("hps_conv3x3_winograd: 8 x 8 maps need splitk_ws (hps_conv3x3_winograd_workspace bytes)");
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
    g.B = B; g.ksplit = 1; g.cps = Cin / WK; g.raw = 0;
    if (quad) {                     // the kernel writes raw partial sums (slice, image, 8, 8, Cout); the second pass finishes
        g.ksplit = wino_quad_ksplit(Cin);
        g.cps = Cin / WK / g.ksplit;
        g.raw = 1;
        g.out_row = 8 * Cout; g.out_img = 64 * Cout; g.opad = 0;
        g.items = ((B + 3) / 4) * g.n_ct * g.ksplit;
    }
    g.magic_ct = wino_magic((unsigned)g.n_ct);
    g.magic_img = wino_magic((unsigned)g.blocks_img);
    g.magic_x = wino_magic((unsigned)g.blocks_x);
    g.magic_ks = wino_magic((unsigned)g.ksplit);
    const size_t lds = (size_t)(4 * W_OPER + 3 * W_RAW) * sizeof(float);           // 162 176 bytes
    // persistent grid: one workgroup per CU (256 on MI355X), items strided over the workgroups
    const dim3 grid((unsigned)(g.items < 256 ? g.items : 256));
    int grant_rc = HPS_OK;
    auto launch = [&](auto AB, auto Q) {
        constexpr int ab = decltype(AB)::value;
        constexpr bool q = decltype(Q)::value;
        if ((grant_rc = grant_lds<&conv_wino_kernel<ab, q>>(160 * 1024, "hps_conv3x3_winograd")) != HPS_OK) return;
        hipLaunchKernelGGL((conv_wino_kernel<ab, q>), grid, dim3(256), lds, (hipStream_t)stream, x, u, scale, shift, quad ? nullptr : residual,
                           quad ? splitk_ws : y, g);
    };
    auto launch8 = [&](auto Q, auto TRF, auto AB) {           // the eight-wave forms (the product: TRF = false; true = dev ablation 23, the lane-per-tile store form, 2-6 % slower)
        constexpr bool q = decltype(Q)::value, tr = decltype(TRF)::value;
        constexpr int ab = decltype(AB)::value;
        if ((grant_rc = grant_lds<&conv_wino_kernel<ab, q, true, tr>>(160 * 1024, "hps_conv3x3_winograd")) != HPS_OK) return;
        hipLaunchKernelGGL((conv_wino_kernel<ab, q, true, tr>), grid, dim3(512), lds, (hipStream_t)stream, x, u, scale, shift, quad ? nullptr : residual,
                           quad ? splitk_ws : y, g);
    };
    typedef std::integral_constant<int, 0> Z;
    typedef std::integral_constant<bool, false> F;
    typedef std::integral_constant<bool, true> T;
    if (quad) {
        if (ablate != 0 && ablate != 21 && ablate != 23) return bad_arg("hps_conv3x3_winograd: ablations exist for the 16 x 16-block geometry only");
#ifdef HPS_DEV_BUILD
        if (ablate == 21) launch(std::integral_constant<int, 0>(), T());      // the four-wave form
        else if (ablate == 23) launch8(T(), T(), Z());                        // eight waves, a lane owns a tile (16-byte stores: slower)
        else
#endif
            launch8(T(), F(), Z());
        if (grant_rc != HPS_OK) return grant_rc;
        const int rc = check_launch("hps_conv3x3_winograd");
        if (rc != HPS_OK) return rc;
        const int total4 = B * 64 * Cout / 4;
        hipLaunchKernelGGL(wino_splitk_epilogue_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, splitk_ws,
                           scale, shift, residual, y, total4, g.ksplit, Cout, opad, relu);
        return check_launch("hps_conv3x3_winograd (slices)");
    }
    switch (ablate) {
        case 0: launch8(F(), F(), Z()); break;                                  // the product form: eight waves, a lane owns a channel
#ifdef HPS_DEV_BUILD
        case 23: launch8(F(), T(), Z()); break;                                 // eight waves, a lane owns a tile (16-byte stores; identical bits, slower)
        case 24: launch8(F(), F(), std::integral_constant<int, 4>()); break;    // profiling: the product form without its epilogue
        case 31: launch8(F(), F(), std::integral_constant<int, 1>()); break;    // ... without patch reads / transform / window DMA
        case 33: launch8(F(), F(), std::integral_constant<int, 3>()); break;    // ... without filter DMA
        case 35: launch8(F(), F(), std::integral_constant<int, 5>()); break;    // ... window DMA but no transform
        case 36: launch8(F(), F(), std::integral_constant<int, 6>()); break;    // ... transform but no window DMA
        case 40: launch8(F(), F(), std::integral_constant<int, 10>()); break;   // ... no barrier per chunk (races)
        case 42: launch8(F(), F(), std::integral_constant<int, 12>()); break;   // experiment: raised wave priority around the MFMAs
        case 43: launch8(F(), F(), std::integral_constant<int, 13>()); break;   // timing only: the MFMAs as bf16x3 piece products (3 bf16 MFMAs per position and 8 k)
        case 44: launch8(F(), F(), std::integral_constant<int, 14>()); break;   // ... + the VALU work of splitting the transformed input into three bf16 pieces
        case 11: launch8(F(), F(), std::integral_constant<int, 11>()); break;   // profiling: lane = channel form with clock stamps (hps_dev_wino_stamps)
        case 21: launch(std::integral_constant<int, 0>(), F()); break;          // the four-wave form (identical bits; the ablations below are its)
        case 1: launch(std::integral_constant<int, 1>(), F()); break;
        case 2: launch(std::integral_constant<int, 2>(), F()); break;
        case 3: launch(std::integral_constant<int, 3>(), F()); break;
        case 4: launch(std::integral_constant<int, 4>(), F()); break;
        case 5: launch(std::integral_constant<int, 5>(), F()); break;
        case 6: launch(std::integral_constant<int, 6>(), F()); break;
        case 7: launch(std::integral_constant<int, 7>(), F()); break;
        case 8: launch(std::integral_constant<int, 8>(), F()); break;
        case 9: launch(std::integral_constant<int, 9>(), F()); break;
        case 10: launch(std::integral_constant<int, 10>(), F()); break;
#endif
        default: return bad_arg("hps_conv3x3_winograd: ablate");
    }
    if (grant_rc != HPS_OK) return grant_rc;
    return check_launch("hps_conv3x3_winograd");
}

extern "C" int hps_conv3x3_winograd(const float* x, const float* u, const float* scale, const float* shift, const float* residual,
                                    float* y, int B, int H, int W, int ipad, int Cin, int Cout, int opad, int relu,
                                    float* splitk_ws, hps_stream_t stream) {
    return wino_launch(x, u, scale, shift, residual, y, B, H, W, ipad, Cin, Cout, opad, relu, splitk_ws, 0, stream);
}

#ifdef HPS_DEV_BUILD
// experiment: the half-item form (two four-wave workgroups per CU); u4 = resnet.py's half-chunk packing of the transformed filters
extern "C" int hps_dev_conv3x3_winograd_half(const float* x, const float* u4, const float* scale, const float* shift, const float* residual,
                                             float* y, int B, int H, int W, int ipad, int Cin, int Cout, int opad, int relu, int ablate,
                                             int wgs_per_cu, hps_stream_t stream) {
    if (!x || !u4 || !scale || !shift || !y) return bad_arg("hps_dev_conv3x3_winograd_half: null pointer");
    if (B <= 0) return HPS_OK;
    if (H <= 0 || W <= 0 || (H % 16) || (W % 16) || Cin % 8 != 0 || Cout % WC != 0 || ipad < 1 || opad < 0)
        return bad_arg("hps_dev_conv3x3_winograd_half: shape");
    WinoGeom g;
    g.in_row = (W + 2 * ipad) * Cin;
    g.in_img = (H + 2 * ipad) * g.in_row;
    g.out_row = (W + 2 * opad) * Cout;
    g.out_img = (H + 2 * opad) * g.out_row;
    g.ipad = ipad; g.opad = opad;
    g.blocks_x = W / 16;
    g.blocks_img = (H / 16) * (W / 16);
    g.Cin = Cin; g.Cout = Cout; g.n_ct = Cout / WC; g.relu = relu;
    g.items = B * g.blocks_img * 2 * g.n_ct;
    g.B = B; g.ksplit = 1; g.cps = Cin / WK; g.raw = 0;
    g.magic_ct = wino_magic((unsigned)g.n_ct);
    g.magic_img = wino_magic((unsigned)g.blocks_img);
    g.magic_x = wino_magic((unsigned)g.blocks_x);
    g.magic_ks = wino_magic(1u);
    const size_t lds = (size_t)(2 * H_A + 2 * H_B + 3 * H_RAW) * sizeof(float);        // 67 584 bytes
    const int wgs = 256 * (wgs_per_cu > 0 ? wgs_per_cu : 2);
    const dim3 grid((unsigned)(g.items < wgs ? g.items : wgs));
    int rc = HPS_OK;
    if (ablate == 4) {
        if ((rc = grant_lds<&conv_wino_half_kernel<4>>(160 * 1024, "hps_dev_conv3x3_winograd_half")) != HPS_OK) return rc;
        hipLaunchKernelGGL((conv_wino_half_kernel<4>), grid, dim3(256), lds, (hipStream_t)stream, x, u4, scale, shift, residual, y, g);
    } else {
        if ((rc = grant_lds<&conv_wino_half_kernel<0>>(160 * 1024, "hps_dev_conv3x3_winograd_half")) != HPS_OK) return rc;
        hipLaunchKernelGGL((conv_wino_half_kernel<0>), grid, dim3(256), lds, (hipStream_t)stream, x, u4, scale, shift, residual, y, g);
    }
    return check_launch("hps_dev_conv3x3_winograd_half");
}
#endif
#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_wino_stamps(unsigned long long* host_out, int n) {      // n <= 4096 stamps of the last ablate-11 launch
    if (!host_out || n < 0 || n > 256 * 16) return bad_arg("hps_dev_wino_stamps");
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? HPS_OK : HPS_E_UNSUPPORTED;
}
extern "C" int hps_dev_wino_quad_ksplit(int ks) {       // experiment: K slices of the 8 x 8 geometry (0 = the product rule)
    g_wino_quad_ks = ks;
    return HPS_OK;
}
extern "C" int hps_dev_conv3x3_winograd(const float* x, const float* u, const float* scale, const float* shift, const float* residual,
                                        float* y, int B, int H, int W, int ipad, int Cin, int Cout, int opad, int relu,
                                        float* splitk_ws, int ablate, hps_stream_t stream) {
    return wino_launch(x, u, scale, shift, residual, y, B, H, W, ipad, Cin, Cout, opad, relu, splitk_ws, ablate, stream);
}
#endif

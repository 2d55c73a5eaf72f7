// ResNet-18 encoder, halo-padded NHWC generation (models/resnet.py:202-217).
//
// What the PMC counters said about the LDS-DMA kernel in conv.hip (128x128 tile, 128->128 3x3 layer): the MFMA pipe
// is busy 73 % of the cycles with the fill and 85 % without it, the clock is the same, and the extra time equals the
// extra *instruction issue* time (SQ_ACTIVE_INST_ANY +750 cycles per K-chunk per wave) -- the two workgroups of a CU
// run in lock step, so the per-chunk address arithmetic of one (tap decode, bounds tests, zero-source select, 64-bit
// lane addresses, the branches hipcc makes of them: ~190 instructions) is not hidden behind the MFMAs of the other.
// This generation removes that arithmetic instead of trying to hide it:
//   * activations live in HBM with a zero halo, (B, H + 2P, W + 2P, C): every filter tap of every output pixel is
//     in bounds, so there is no bounds test and no zero source;
//   * the source address of a DMA piece is  SGPR base (tensor + tap offset, advanced by scalar code per chunk)
//     + a per-lane 32-bit byte offset that never changes inside the kernel  (global_load_lds_dwordx4 v, s[..]);
//   * per chunk and wave that leaves the pieces themselves, three scalar adds and the barrier.
// The stem (7x7 / 2, 18 channels) runs on the same kernel in "row mode": in NHWC one filter row of a window is
// KW * C = 126 contiguous floats, so it is treated as one tap of 128 "channels" (the two extra floats meet zero
// filter entries): K = 7 * 128 = 896 instead of 7 * 7 * 20 = 980, no channel padding of the image.
#include <type_traits>

#include "hps_common.h"

namespace hps {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PBK = 32;   // K-chunk (floats); one 128-byte LDS row per tile row

// n / d for n < 2^32 with magic = floor(2^32 / d) (0xffffffff for d = 1): the estimate is at most one short
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned d, unsigned magic) {
    unsigned q = __umulhi(n, magic);
    if (n - q * d >= d) ++q;
    return q;
}
static unsigned div_magic(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / d); }

// one 1-KiB LDS-DMA piece: lane l fetches 16 bytes at sbase + voff and they land at lds_addr + 16 l.
// Issued from inline asm so that hipcc does not drain it in front of the next ds_read (conv.hip lds_dma16).
__device__ __forceinline__ void lds_dma16_sv(unsigned voff, const float* sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

struct PadGeom {
    int img_pitch, row_pitch, pix_pitch;   // input strides in floats (padded frame)
    int cin_k;                             // contiguous floats per tap, multiple of 32
    int kw;                                // taps per filter row
    int stride, off;                       // output (0,0) / tap (0,0) reads padded input (off, off)
    int Ho, Wo, Mtot, Kp, Cout;
    int opad;                              // halo of the output (and residual) frame
    int relu, tiles_m, ksplit;
    unsigned magic_howo, magic_wo;
    int ablate;                            // tuning only (hps_dev_conv_pad_ablate): 1 = no epilogue
};
// The 1x1 / stride-s down-sample convolution of a residual block's entry (models/resnet.py:71-72, 184-188) issued by EXTRA WORKGROUPS OF
// THE SAME LAUNCH as the block's 3x3 / stride-s convolution (hps_conv2d_bn_act_pad_down): the pixel a 1x1 / s / 0 window reads is the centre
// tap (pad, pad) of the 3x3 / s / pad window of the same output pixel, so both share the geometry (pixel -> input offset, output frame) and
// differ in the filter, the K range (that tap's Cin channels only), BatchNorm constants, destination and ReLU.  Workgroups [0, blocks_main)
// of the grid (x every K slice) are the main convolution's; the rest walk the centre tap with the down-sample's filter -- the same chunks
// in the same order as the separate launch: identical bits.  As launches of their own the three down-samples ran at 0.24-0.32 MFMA-busy
// (16-26 us for 1-2 us of MFMA work per CU: too few, too short workgroups); here they fill the tail of a launch that is there anyway.
struct DownArgs {
    const float* wn;                       // (Cout, Cin) n-major filter of the 1x1 convolution
    const float* scale;
    const float* shift;
    float* y;                              // its output frame (the main convolution's geometry)
    int blocks_main;
    int tap_kh, tap_kw;                    // the main window's tap it reads
};
struct NoDown {};

#ifdef HPS_DEV_BUILD
static int g_pad_ablate = 0;               // dev library only (hps_dev_conv_pad_ablate)
#else
constexpr int g_pad_ablate = 0;            // product library: no process-global switches
#endif

// offset (floats) of output pixel m, channel 0, in the output frame
__device__ __forceinline__ unsigned out_pixel_offset(unsigned m, const PadGeom& g) {
    const unsigned howo = (unsigned)(g.Ho * g.Wo);
    const unsigned b = fastdiv(m, howo, g.magic_howo), rem = m - b * howo;
    const unsigned ho = fastdiv(rem, (unsigned)g.Wo, g.magic_wo), wo = rem - ho * g.Wo;
    return ((b * (g.Ho + 2 * g.opad) + ho + g.opad) * (g.Wo + 2 * g.opad) + wo + g.opad) * (unsigned)g.Cout;
}

// ST = K-loop stages (LDS buffers per operand).  2: the next chunk's DMA pieces are issued during this chunk's MFMAs (the throughput
// form: several workgroups per CU cover each other's DMA latency).  3 / 4 (round 5, latency mode: hps_conv2d_bn_act_pad variant 5):
// two / three chunks ahead -- at batch 1 a 64 x 64 tile's chunk is 16 MFMAs per wave (0.43 us) with ONE workgroup on the CU, and a
// chunk fetched one ahead arrived ~0.6 us after it was needed: 1.0 us per chunk, 18.3 us for a layer1 convolution of 18 chunks
// (three stages: 15.3 us).
template <int BM, int BN, int WM, int WN, int ST = 2, bool DOWN = false>
__global__ __launch_bounds__(256) void conv_pad_kernel(const float* __restrict__ x, const float* __restrict__ wn,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ residual, float* __restrict__ y,
                                                       float* __restrict__ partial, const PadGeom g,
                                                       const std::conditional_t<DOWN, DownArgs, NoDown> d) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM / 32, B_LD = BN / 32;        // DMA pieces per wave per chunk
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                                   // [ST][BM][32]
    float* sB = smem + ST * BM * PBK;                   // [ST][BN][32]

    int bx = blockIdx.x;
    bool down = false;                                   // workgroup-uniform
    if constexpr (DOWN) {
        if (bx >= d.blocks_main) {                       // a workgroup of the 1x1 down-sample convolution (see DownArgs)
            if (blockIdx.y != 0) return;                 // it is never split over K
            down = true;
            bx -= d.blocks_main;
            wn = d.wn; scale = d.scale; shift = d.shift; y = d.y; residual = nullptr;
        }
    }
    const int filter_kp = down ? g.cin_k : g.Kp;         // K extent (floats) of a filter row
    const int ksplit = down ? 1 : g.ksplit;
    const int relu = down ? 0 : g.relu;
    const int tile_n = bx / g.tiles_m, tile_m = bx % g.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int kl = lane >> 5, il = lane & 31;
    // DMA role of this lane: tile row (32 r + 8 wave + lane / 8), LDS slot lane % 8 <- source quad slot ^ f(row),
    // f(row) = (row >> 1) & 7 (conv.hip: conflict-free ds_read_b128)
    const int drow = wave * 8 + (lane >> 3);
    const int dquad = ((lane & 7) ^ ((drow >> 1) & 7)) * 4;

    unsigned a_off[A_LD], b_off[B_LD];                  // per-lane byte offsets, constant for the whole kernel
    {
        const unsigned howo = (unsigned)(g.Ho * g.Wo);
#pragma unroll
        for (int r = 0; r < A_LD; ++r) {
            const unsigned m = (unsigned)(m0 + drow + 32 * r);
            unsigned o = 0;                              // tile overhang rows read pixel 0 (results discarded)
            if (m < (unsigned)g.Mtot) {
                const unsigned b = fastdiv(m, howo, g.magic_howo), rem = m - b * howo;
                const unsigned ho = fastdiv(rem, (unsigned)g.Wo, g.magic_wo), wo = rem - ho * g.Wo;
                o = b * g.img_pitch + (ho * g.stride + g.off) * g.row_pitch + (wo * g.stride + g.off) * g.pix_pitch;
            }
            a_off[r] = (o + dquad) * 4u;
        }
#pragma unroll
        for (int r = 0; r < B_LD; ++r) b_off[r] = ((unsigned)(n0 + drow + 32 * r) * filter_kp + dquad) * 4u;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // split-K: blockIdx.y owns the chunk range [c_begin, c_end) and writes raw partial sums (summed in slice order by
    // splitk_pad_epilogue_kernel: deterministic, no atomics)
    const int chunks_total = filter_kp / PBK;
    const int per_split = chunks_total / ksplit;
    const int c_begin = down ? 0 : blockIdx.y * per_split, c_end = c_begin + per_split;
    // scalar walk over (kh, kw, ci0): the A base pointer advances by 32 floats per chunk and jumps at tap / row ends
    const int cpt = g.cin_k / PBK;
    int ci = c_begin % cpt, tap = c_begin / cpt;
    int kw = tap % g.kw, kh = tap / g.kw;
    if constexpr (DOWN) {
        if (down) { kh = d.tap_kh; kw = d.tap_kw; }      // the one tap the 1x1 window reads; ci = 0 .. cpt - 1
    }
    const float* a_src = x + (size_t)kh * g.row_pitch + (size_t)kw * g.pix_pitch + (size_t)ci * PBK;
    const float* b_src = wn + (size_t)c_begin * PBK;
    const int tap_jump = g.pix_pitch - g.cin_k + PBK;                    // from the last chunk of a tap to the next tap
    const int row_jump = g.row_pitch - (g.kw - 1) * g.pix_pitch - g.cin_k + PBK;   // ... to the next filter row

    const unsigned lds_a = (unsigned)(size_t)(lptr_t)(sA + wave * 8 * PBK);
    const unsigned lds_b = (unsigned)(size_t)(lptr_t)(sB + wave * 8 * PBK);
    auto advance = [&]() {                               // source pointers of the next chunk (scalar code)
        b_src += PBK;
        if (++ci < cpt) a_src += PBK;
        else {
            ci = 0;
            if (++kw < g.kw) a_src += tap_jump;
            else { kw = 0; a_src += row_jump; }
        }
    };
    auto dma_chunk = [&](int buf) {
        const unsigned la = __builtin_amdgcn_readfirstlane(lds_a + buf * BM * PBK * 4);
        const unsigned lb = __builtin_amdgcn_readfirstlane(lds_b + buf * BN * PBK * 4);
#pragma unroll
        for (int r = 0; r < A_LD; ++r) lds_dma16_sv(a_off[r], a_src, la + r * 32 * PBK * 4);
#pragma unroll
        for (int r = 0; r < B_LD; ++r) lds_dma16_sv(b_off[r], b_src, lb + r * 32 * PBK * 4);
        advance();
    };
    // one piece of the chunk whose sources are (pa_src, pb_src): piece p < A_LD is an A piece, the rest B pieces
    auto dma_piece = [&](int p, const float* pa_src, const float* pb_src, unsigned la, unsigned lb) {
        if (p < A_LD) lds_dma16_sv(a_off[p < A_LD ? p : 0], pa_src, la + p * 32 * PBK * 4);
        else if (p < A_LD + B_LD) lds_dma16_sv(b_off[p >= A_LD && p < A_LD + B_LD ? p - A_LD : 0], pb_src, lb + (p - A_LD) * 32 * PBK * 4);
    };

    const int fsw = (il >> 1) & 7;                       // f(row) of the fragment rows this lane reads
#ifdef HPS_DEV_BUILD
    if (g.ablate >= 2) {       // experiment: de-phase the workgroups that share a CU (ablate - 1 sleeps of 3.4 us for the second slot)
        if (((blockIdx.x >> 8) & 1) && blockIdx.x < 512)
            for (int i = 0; i < (g.ablate - 1) * 1; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    dma_chunk(0);
#pragma unroll
    for (int q = 1; q < ST - 1; ++q)
        if (c_begin + q < c_end) dma_chunk(q);
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = ST == 2 ? (c - c_begin) & 1 : (c - c_begin) % ST;
        // this wave's pieces of chunk c have landed; with more than two stages the pieces of up to ST - 2 younger chunks, issued after
        // them (A_LD + B_LD per chunk), may still be in flight
        {
            const int younger = min(ST - 2, c_end - 1 - c);
            if (ST >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (A_LD + B_LD)) : "memory");
            else if (ST >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + B_LD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                     // ... everyone's have, and the buffer refilled below is no longer being read
        // The next chunk's DMA pieces are issued ONE PER FOUR MFMAs, not in a burst after the barrier: every 1 KiB piece costs the
        // SIMD 36-57 cycles that no other wave's MFMAs cover (tools/mfma_dma_overlap.hip: the loop skeleton runs at 139.4 TF/s with
        // the burst and 143.0 spread); on the stem 0.906 -> 0.888 ms (tests/dev/stem_ablate.py, mode 3 = burst).
        static_assert(A_LD + B_LD <= (PBK / 8) * TM * TN, "one DMA piece per accumulator block of a chunk at most");
#ifdef HPS_DEV_BUILD
        const bool spread = g.ablate != 3 || ST != 2;    // hps_dev_conv_pad_ablate(3): the earlier burst (two-stage form only)
#else
        constexpr bool spread = true;
#endif
        const bool more = c + (ST - 1) < c_end;          // the chunk fetched during this one: c + 1 (two stages), c + 2, c + 3
        const float* na_src = a_src;
        const float* nb_src = b_src;
        const int nbuf = ST == 2 ? buf ^ 1 : (c - c_begin + ST - 1) % ST;
        const unsigned nla = __builtin_amdgcn_readfirstlane(lds_a + nbuf * BM * PBK * 4);
        const unsigned nlb = __builtin_amdgcn_readfirstlane(lds_b + nbuf * BN * PBK * 4);
        if (more) {
            if (spread) advance();
            else dma_chunk(buf ^ 1);
        }
        const float* pa = sA + (size_t)buf * BM * PBK + (wm0 + il) * PBK;
        const float* pb = sB + (size_t)buf * BN * PBK + (wn0 + il) * PBK;
        // Fragments are double-buffered by hand: group gq + 1 is requested before the MFMAs of group gq are issued (the
        // sched_barriers keep hipcc from sinking the reads to their uses -- left alone it reloads the same 16 registers after
        // the group's last MFMA and waits out the LDS round trip in front of the next group).
        float4 a4[2][TM], b4[2][TN];
        auto load_frags = [&](int gq, int s) {
            const int slot = ((2 * gq + kl) ^ fsw) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[s][i] = *reinterpret_cast<const float4*>(pa + i * 32 * PBK + slot);
#pragma unroll
            for (int j = 0; j < TN; ++j) b4[s][j] = *reinterpret_cast<const float4*>(pb + j * 32 * PBK + slot);
        };
        load_frags(0, 0);
#pragma unroll
        for (int gq = 0; gq < PBK / 8; ++gq) {
            if (gq + 1 < PBK / 8) load_frags(gq + 1, (gq + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            // k order inside the group is x, y, z, w for every accumulator (same summation order as conv.hip)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[gq & 1][j].x, a4[gq & 1][i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[gq & 1][j].y, a4[gq & 1][i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[gq & 1][j].z, a4[gq & 1][i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[gq & 1][j].w, a4[gq & 1][i].w, acc[i][j], 0, 0, 0);
                    if (spread && more) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma_piece((gq * TM + i) * TN + j, na_src, nb_src, nla, nlb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // The filter fragment is the MFMA's row operand and the pixel fragment its column operand (products commute, the
    // k order is unchanged), so a lane ends up with ONE pixel (column = lane & 31) and, per register quad q, four
    // consecutive output channels: row = (r & 3) + 8 q + 4 (lane >> 5)  ->  128-bit accesses along Cout.
    if (ksplit > 1) {
        float* dst = partial + (size_t)blockIdx.y * g.Mtot * g.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm0 + i * 32 + il;
            if (m >= g.Mtot) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(dst + (size_t)m * g.Cout + n0 + wn0 + j * 32 + 8 * q + 4 * kl) =
                        make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        }
        return;
    }

    if (g.ablate == 1) {        // tuning: keep the accumulators alive, write (almost) nothing
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) y[0] = t;
        return;
    }

    // Epilogue: BatchNorm (scale, shift), residual, ReLU, stored as whole pixel rows.  Each wave transposes its 32-pixel
    // sub-tiles through a private LDS patch [32][WN + 4] (the +4 keeps both the 128-bit writes -- 16 lanes = 16 pixels,
    // bank step 4 -- and the row reads conflict free), then WN/4 consecutive lanes own one pixel's WN channels: residual
    // loads and output stores are 16 bytes per lane, WN*4 contiguous bytes per pixel (64 stores -> 16 per wave).
    constexpr int EP = WN + 4;                 // patch pitch (floats)
    constexpr int LPP = WN / 4;                // lanes per pixel
    constexpr int PPI = 64 / LPP;              // pixels per wave instruction
    static_assert(4 * 32 * EP <= ST * (BM + BN) * PBK, "the epilogue patches fit in the K-loop buffers");
    __syncthreads();                           // every wave is done with the K-loop buffers
    float* patch = smem + wave * 32 * EP;
    const int c4 = (lane % LPP) * 4;
    const int co = n0 + wn0 + c4;
    const float4 sc = *reinterpret_cast<const float4*>(scale + co), sh = *reinterpret_cast<const float4*>(shift + co);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(patch + il * EP + j * 32 + 8 * q + 4 * kl) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        // wave-private patch: no barrier, the LDS queue is in order per wave
        constexpr int U = 32 / PPI;
        if (m0 + BM <= g.Mtot) {
            // Full tile (every tile of the benchmark shapes): no per-pixel guards, so this is straight-line code -- all patch rows
            // are requested before the first is used and the stores follow back to back.  (With the guards hipcc put every
            // row's read, wait and store into a block of its own: 16 exposed LDS round trips per sub-tile.)
            unsigned po[U];
            float4 a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                po[u] = out_pixel_offset((unsigned)(m0 + wm0 + i * 32 + u * PPI + lane / LPP), g) + (unsigned)co;
                a[u] = *reinterpret_cast<const float4*>(patch + (u * PPI + lane / LPP) * EP + c4);
            }
            if (residual) {
                float4 res[U];
#pragma unroll
                for (int u = 0; u < U; ++u) res[u] = *reinterpret_cast<const float4*>(residual + po[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float4 v = make_float4(a[u].x * sc.x + sh.x + res[u].x, a[u].y * sc.y + sh.y + res[u].y,
                                           a[u].z * sc.z + sh.z + res[u].z, a[u].w * sc.w + sh.w + res[u].w);
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<float4*>(y + po[u]) = v;
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    // "+ 0" as in the guarded path (residual absent = zero addend): the same bits
                    float4 v = make_float4(a[u].x * sc.x + sh.x + 0.f, a[u].y * sc.y + sh.y + 0.f, a[u].z * sc.z + sh.z + 0.f,
                                           a[u].w * sc.w + sh.w + 0.f);
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<float4*>(y + po[u]) = v;
                }
            }
            continue;
        }
        unsigned po[U];
        bool live[U];
        float4 res[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + wm0 + i * 32 + u * PPI + lane / LPP;
            live[u] = m < g.Mtot;
            po[u] = live[u] ? out_pixel_offset((unsigned)m, g) + (unsigned)co : 0u;
            res[u] = (residual && live[u]) ? *reinterpret_cast<const float4*>(residual + po[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 a = *reinterpret_cast<const float4*>(patch + (u * PPI + lane / LPP) * EP + c4);
            float4 v = make_float4(a.x * sc.x + sh.x + res[u].x, a.y * sc.y + sh.y + res[u].y, a.z * sc.z + sh.z + res[u].z,
                                   a.w * sc.w + sh.w + res[u].w);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (live[u]) *reinterpret_cast<float4*>(y + po[u]) = v;
        }
    }
}

// second pass of a split-K convolution: y = act(scale * (sum of the slices, in slice order) + shift + residual),
// y / residual in the padded output frame
__global__ __launch_bounds__(256) void splitk_pad_epilogue_kernel(const float* __restrict__ partial,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  const float* __restrict__ residual,
                                                                  float* __restrict__ y, long total4, const PadGeom g) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    // Six slices requested before the first is added (the additions stay in slice order: the same bits).  As a plain loop hipcc
    // waited out one L2 round trip per slice: 5-7 us for the latency mode's 12-18 slices, eleven times per image.
    const float4* p4 = reinterpret_cast<const float4*>(partial) + i;
    float4 acc = p4[0];
    int k = 1;
    for (; k + 6 <= g.ksplit; k += 6) {
        float4 v[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = p4[(size_t)(k + q) * total4];
#pragma unroll
        for (int q = 0; q < 6; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    }
    for (; k < g.ksplit; ++k) {
        const float4 v = p4[(size_t)k * total4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const unsigned e = (unsigned)(i * 4);
    const unsigned m = e / (unsigned)g.Cout, co = e - m * g.Cout;
    const size_t o = (size_t)out_pixel_offset(m, g) + co;
    const float4 sc = *reinterpret_cast<const float4*>(scale + co), sh = *reinterpret_cast<const float4*>(shift + co);
    float4 v = make_float4(acc.x * sc.x + sh.x, acc.y * sc.y + sh.y, acc.z * sc.z + sh.z, acc.w * sc.w + sh.w);
    if (residual) {
        const float4 r = *reinterpret_cast<const float4*>(residual + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(y + o) = v;
}

// (B,C,H,W) -> (B, H + 2P, W + 2P, C) interior (the halo is zeroed once by the owner of the buffer).  A workgroup moves
// one run of up to 256 pixels of an image row: lanes along w read each channel plane coalesced into LDS [pixel][C], then
// the run's 256*C contiguous output floats leave as lane-contiguous 8-byte stores (pixel records are 8-byte aligned
// for even C; 512 contiguous bytes per wave instruction instead of 64 scattered records).
template <int C>
__global__ __launch_bounds__(256) void nchw_to_padded_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int H, int W, int P, int runs_per_row) {
    __shared__ float sp[256 * C];
    const int t = threadIdx.x;
    const int run = blockIdx.x % runs_per_row;
    const long row = blockIdx.x / runs_per_row;              // b * H + h
    const long b = row / H;
    const int h = (int)(row - b * H);
    const int w0 = run * 256, n = min(256, W - w0);
    const long hw = (long)H * W;
    if (t < n) {
        const float* src = x + (b * C) * hw + (long)h * W + w0 + t;
#pragma unroll
        for (int c = 0; c < C; ++c) sp[t * C + c] = src[c * hw];
    }
    __syncthreads();
    float2* dst = reinterpret_cast<float2*>(y + ((b * (H + 2 * P) + h + P) * (long)(W + 2 * P) + w0 + P) * C);
    const float2* s2 = reinterpret_cast<const float2*>(sp);
    for (int i = t; i < n * C / 2; i += 256) dst[i] = s2[i];
}

// Any channel count and width (the reference's ResNet takes any in_channels and image size, models/resnet.py:127-176): thread per
// (pixel, channel), channel c < C of pixel (h, w) goes to the interior of a (B, H + 2P, WF + 2P, CP) frame whose other entries
// (channels C..CP-1, columns W..WF-1, the halo) the owner zeroed once.  Not a hot path: the released model's shapes take the
// kernels above / the phase split.
__global__ __launch_bounds__(256) void nchw_to_padded_nhwc_generic_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                                          int CP, int H, int W, int WF, int P, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int w = (int)(i % W);
    long r = i / W;
    const int h = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const long b = r / C;
    y[((b * (H + 2 * P) + h + P) * (long)(WF + 2 * P) + w + P) * CP + c] = x[i];
}

// MaxPool2d(3, 2, 1) on NHWC, thread per (2 x 2 output pixels, 4 channels); output written into a frame with halo opad.
// The four windows of a 2 x 2 output block share a 5 x 5 input patch: 25 loads for four outputs instead of 36 (the kernel is
// bound by the requests its loads put to the vector cache, not by HBM: every input pixel used to be fetched 2.25 times), and
// the row maxima are formed once per patch row and column triple.
__global__ __launch_bounds__(256) void maxpool_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                          int C, int Ho, int Wo, int opad, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = C / 4, Ho2 = (Ho + 1) / 2, Wo2 = (Wo + 1) / 2;
    const int cq = (int)(i % c4);
    long p = i / c4;
    const int wo = 2 * (int)(p % Wo2); p /= Wo2;
    const int ho = 2 * (int)(p % Ho2);
    const long b = p / Ho2;
    // Guard-free: a tap outside the map is clamped onto the nearest row / column, which belongs to the window anyway (max is
    // idempotent: exact).  All loads go out before the first comparison; with `if (outside) continue` every load sat in a
    // block of its own behind s_waitcnt vmcnt(0).
    float4 v[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int hi = min(max(ho * 2 - 1 + r, 0), H - 1);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int wi = min(max(wo * 2 - 1 + q, 0), W - 1);
            v[r][q] = *reinterpret_cast<const float4*>(x + ((b * H + hi) * W + wi) * C + cq * 4);
        }
    }
    auto max4 = [](const float4& a, const float4& c) { return make_float4(fmaxf(a.x, c.x), fmaxf(a.y, c.y), fmaxf(a.z, c.z), fmaxf(a.w, c.w)); };
    float4 hm[5][2];               // maxima over the columns 0..2 / 2..4 of every patch row
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        hm[r][0] = max4(max4(v[r][0], v[r][1]), v[r][2]);
        hm[r][1] = max4(max4(v[r][2], v[r][3]), v[r][4]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (ho + a >= Ho || wo + c >= Wo) continue;
            const float4 m = max4(max4(hm[2 * a][c], hm[2 * a + 1][c]), hm[2 * a + 2][c]);
            *reinterpret_cast<float4*>(y + ((b * (Ho + 2 * opad) + ho + a + opad) * (long)(Wo + 2 * opad) + wo + c + opad) * C + cq * 4) = m;
        }
}

// global average pool over the interior of a padded frame: thread per (b, c), coalesced over c, pixels in row-major order
__global__ __launch_bounds__(256) void avgpool_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                          int C, int P, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = i / C, c = i % C;
    const int Wp = W + 2 * P;
    const float* s = x + ((size_t)b * (H + 2 * P) + P) * Wp * C + (size_t)P * C + c;
    // eight pixels requested before the first is added (row-major order kept: the same bits); a column beyond W re-reads
    // column W - 1 and adds +0
    float acc = 0.0f;
    for (int h = 0; h < H; ++h)
        for (int w0 = 0; w0 < W; w0 += 8) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = s[((size_t)h * Wp + min(w0 + q, W - 1)) * C];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += (w0 + q < W) ? t[q] : 0.0f;
        }
    y[i] = acc / (float)(H * W);
}

template <int BM, int BN, int WM, int WN, int ST = 2>
static int launch_conv_pad(const float* x, const float* wn, const float* scale, const float* shift, const float* residual,
                           float* y, float* partial, PadGeom g, hipStream_t s, const DownArgs* down = nullptr) {
    g.tiles_m = ceil_div(g.Mtot, BM);
    const int tiles_n = g.Cout / BN;
    const size_t lds = (size_t)ST * (BM + BN) * PBK * sizeof(float);
    if (down) {                 // the block's 1x1 down-sample rides in the same launch: as many workgroups again, one K slice
        DownArgs d = *down;
        d.blocks_main = g.tiles_m * tiles_n;
        if (lds > 64 * 1024)
            if (int rc = grant_lds<&conv_pad_kernel<BM, BN, WM, WN, ST, true>>((int)lds, "hps_conv2d_bn_act_pad_down")) return rc;
        hipLaunchKernelGGL((conv_pad_kernel<BM, BN, WM, WN, ST, true>), dim3(2 * d.blocks_main, g.ksplit), dim3(256), lds, s, x, wn, scale,
                           shift, residual, y, partial, g, d);
    } else {
        if (lds > 64 * 1024)
            if (int rc = grant_lds<&conv_pad_kernel<BM, BN, WM, WN, ST>>((int)lds, "hps_conv2d_bn_act_pad")) return rc;
        hipLaunchKernelGGL((conv_pad_kernel<BM, BN, WM, WN, ST>), dim3(g.tiles_m * tiles_n, g.ksplit), dim3(256), lds, s, x, wn, scale,
                           shift, residual, y, partial, g, NoDown());
    }
    if (g.ksplit > 1) {
        const long total4 = (long)g.Mtot * g.Cout / 4;
        hipLaunchKernelGGL(splitk_pad_epilogue_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, partial, scale,
                           shift, residual, y, total4, g);
    }
    return check_launch("hps_conv2d_bn_act_pad");
}

}  // namespace hps

using namespace hps;

static int conv_pad_entry(const float* x, const float* wn, const float* scale, const float* shift,
                          const float* residual, float* y, int B, int H, int W, int ipad, int Cin, int Cout,
                          int KH, int KW, int stride, int pad, int opad, int relu, int row_mode, int variant,
                          int ksplit, float* splitk_ws, hps_stream_t stream, DownArgs* down) {
    if (!x || !wn || !scale || !shift || !y) return bad_arg("hps_conv2d_bn_act_pad: null pointer");
    if (ipad < pad || opad < 0) return bad_arg("hps_conv2d_bn_act_pad: the input halo must cover the convolution padding");
    if (Cout % 64 != 0) return bad_arg("hps_conv2d_bn_act_pad: Cout % 64 == 0 required");
    if (B <= 0) return HPS_OK;
    PadGeom g;
    const int Hp = H + 2 * ipad, Wp = W + 2 * ipad;
    g.pix_pitch = Cin;
    g.row_pitch = Wp * Cin;
    g.img_pitch = Hp * g.row_pitch;
    g.stride = stride;
    g.off = ipad - pad;
    g.Ho = (H + 2 * pad - KH) / stride + 1;
    g.Wo = (W + 2 * pad - KW) / stride + 1;
    g.Mtot = B * g.Ho * g.Wo;
    g.Cout = Cout;
    g.opad = opad;
    g.relu = relu;
    g.ksplit = ksplit < 1 ? 1 : ksplit;
    if (row_mode) {
        // a filter row = KW * Cin contiguous floats, rounded up to the chunk; the DMA needs 16-byte aligned windows
        g.cin_k = ceil_div(KW * Cin, PBK) * PBK;
        g.kw = 1;
        if ((stride * Cin) % 4 != 0 || g.row_pitch % 4 != 0 || (g.off * Cin) % 4 != 0)
            return bad_arg("hps_conv2d_bn_act_pad: row mode needs 16-byte aligned window starts");
        if (((g.Wo - 1) * stride + g.off) * Cin + g.cin_k > g.row_pitch)
            return bad_arg("hps_conv2d_bn_act_pad: row mode window overruns the padded row");
    } else {
        if (Cin % PBK != 0) return bad_arg("hps_conv2d_bn_act_pad: Cin % 32 == 0 required (or row mode)");
        g.cin_k = Cin;
        g.kw = KW;
    }
    g.Kp = KH * g.kw * g.cin_k;
    if ((size_t)B * g.img_pitch * 4 >= 0xffffffffull || (size_t)Cout * g.Kp * 4 >= 0xffffffffull)
        return bad_arg("hps_conv2d_bn_act_pad: tensor exceeds the 32-bit lane offsets");
    if ((g.Kp / PBK) % g.ksplit != 0 || (g.ksplit > 1 && !splitk_ws)) return bad_arg("hps_conv2d_bn_act_pad: ksplit");
    g.magic_howo = div_magic((unsigned)(g.Ho * g.Wo));
    g.magic_wo = div_magic((unsigned)g.Wo);
    g.ablate = g_pad_ablate;
    hipStream_t s = (hipStream_t)stream;
    // split-K runs on the 128-row tiles unless the caller asks for the 64 x 64 ones (variants 3 / 5: the latency mode, where a
    // single image's 8 x 8 ... 32 x 32 maps leave half of a 128-row tile empty and a quarter as many workgroups on the chip);
    // the summation order of an output does not depend on the tile shape: the same bits
    if (g.ksplit > 1 && variant != 3 && variant != 5) variant = Cout % 128 == 0 ? 1 : 2;
    if (variant == 0) {
        // same rule as hps_conv2d_bn_act_v3 (measured per layer): largest tile that still gives every CU a workgroup
        if (Cout % 128 == 0 && ((long)g.Mtot / 128) * (Cout / 128) >= 256) variant = 1;
        else if (Cout == 64 && g.Mtot / 256 >= 512) variant = 4;
        else variant = 3;
    }
    if (variant == 1 && Cout % 128 != 0) variant = 2;
    if (down) {
        // the tile shapes a residual block's entry convolution takes (Cout >= 128): 128 x 128, 128 x 64, 64 x 64 (two- and four-stage)
        switch (variant) {
            case 1: return launch_conv_pad<128, 128, 64, 64>(x, wn, scale, shift, residual, y, splitk_ws, g, s, down);
            case 2: return launch_conv_pad<128, 64, 64, 32>(x, wn, scale, shift, residual, y, splitk_ws, g, s, down);
            case 3: return launch_conv_pad<64, 64, 32, 32>(x, wn, scale, shift, residual, y, splitk_ws, g, s, down);
            case 5: return launch_conv_pad<64, 64, 32, 32, 4>(x, wn, scale, shift, residual, y, splitk_ws, g, s, down);
            default: set_error("hps_conv2d_bn_act_pad_down: tile variant %d carries no down-sample (issue the two convolutions separately)", variant);
                     return HPS_E_UNSUPPORTED;
        }
    }
    switch (variant) {
        case 1: return launch_conv_pad<128, 128, 64, 64>(x, wn, scale, shift, residual, y, splitk_ws, g, s);
        case 2: return launch_conv_pad<128, 64, 64, 32>(x, wn, scale, shift, residual, y, splitk_ws, g, s);
        case 3: return launch_conv_pad<64, 64, 32, 32>(x, wn, scale, shift, residual, y, splitk_ws, g, s);
        case 4: return launch_conv_pad<256, 64, 64, 64>(x, wn, scale, shift, residual, y, splitk_ws, g, s);
        case 5: return launch_conv_pad<64, 64, 32, 32, 4>(x, wn, scale, shift, residual, y, splitk_ws, g, s);      // four-stage K loop (latency mode)
#ifdef HPS_DEV_BUILD
        case 6: return launch_conv_pad<64, 64, 32, 32, 3>(x, wn, scale, shift, residual, y, splitk_ws, g, s);      // three stages (A/B)
#endif
        default: return bad_arg("hps_conv2d_bn_act_pad: variant");
    }
}

extern "C" int hps_conv2d_bn_act_pad(const float* x, const float* wn, const float* scale, const float* shift,
                                     const float* residual, float* y, int B, int H, int W, int ipad, int Cin, int Cout,
                                     int KH, int KW, int stride, int pad, int opad, int relu, int row_mode, int variant,
                                     int ksplit, float* splitk_ws, hps_stream_t stream) {
    return conv_pad_entry(x, wn, scale, shift, residual, y, B, H, W, ipad, Cin, Cout, KH, KW, stride, pad, opad, relu, row_mode, variant,
                          ksplit, splitk_ws, stream, nullptr);
}

extern "C" int hps_conv2d_bn_act_pad_down(const float* x, const float* wn, const float* scale, const float* shift, float* y,
                                          const float* wn_down, const float* scale_down, const float* shift_down, float* y_down,
                                          int B, int H, int W, int ipad, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                          int opad, int relu, int variant, int ksplit, float* splitk_ws, hps_stream_t stream) {
    if (!wn_down || !scale_down || !shift_down || !y_down) return bad_arg("hps_conv2d_bn_act_pad_down: null pointer");
    if (KH != KW || KH % 2 != 1 || pad != KH / 2 || stride < 1)
        return bad_arg("hps_conv2d_bn_act_pad_down: the main convolution must be k x k / s / (k / 2) (its centre tap is the 1x1 / s / 0 window)");
    DownArgs d;
    d.wn = wn_down; d.scale = scale_down; d.shift = shift_down; d.y = y_down;
    d.blocks_main = 0;
    d.tap_kh = pad; d.tap_kw = pad;
    return conv_pad_entry(x, wn, scale, shift, nullptr, y, B, H, W, ipad, Cin, Cout, KH, KW, stride, pad, opad, relu, 0, variant, ksplit,
                          splitk_ws, stream, &d);
}

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_conv_pad_ablate(int mode) {
    g_pad_ablate = mode;
    return HPS_OK;
}
#endif

extern "C" int hps_nchw_to_padded_nhwc(const float* x, float* y, int B, int C, int H, int W, int P, hps_stream_t stream) {
    if (!x || !y) return bad_arg("hps_nchw_to_padded_nhwc: null pointer");
    if (B <= 0 || H <= 0 || W <= 0) return HPS_OK;
    const int runs = ceil_div(W, 256);
    const long blocks = (long)B * H * runs;
    if (blocks > 0x7fffffffL) return bad_arg("hps_nchw_to_padded_nhwc: too many rows");
    hipStream_t s = (hipStream_t)stream;
    if (C == 18) hipLaunchKernelGGL(nchw_to_padded_nhwc_kernel<18>, dim3((unsigned)blocks), dim3(256), 0, s, x, y, H, W, P, runs);
    else if (C == 4) hipLaunchKernelGGL(nchw_to_padded_nhwc_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, x, y, H, W, P, runs);
    else if (C == 64) hipLaunchKernelGGL(nchw_to_padded_nhwc_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, s, x, y, H, W, P, runs);
    else return bad_arg("hps_nchw_to_padded_nhwc: C must be 4, 18 or 64");
    return check_launch("hps_nchw_to_padded_nhwc");
}

extern "C" int hps_nchw_to_padded_nhwc_generic(const float* x, float* y, int B, int C, int CP, int H, int W, int WF, int P,
                                               hps_stream_t stream) {
    if (!x || !y) return bad_arg("hps_nchw_to_padded_nhwc_generic: null pointer");
    if (C <= 0 || CP < C || WF < W || P < 0) return bad_arg("hps_nchw_to_padded_nhwc_generic: need CP >= C > 0, WF >= W, P >= 0");
    if (B <= 0 || H <= 0 || W <= 0) return HPS_OK;
    const long total = (long)B * C * H * W;
    if ((total + 255) / 256 > 0x7fffffffL) return bad_arg("hps_nchw_to_padded_nhwc_generic: tensor too large");
    hipLaunchKernelGGL(nchw_to_padded_nhwc_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       C, CP, H, W, WF, P, total);
    return check_launch("hps_nchw_to_padded_nhwc_generic");
}

extern "C" int hps_maxpool3x3s2_pad(const float* x, float* y, int B, int H, int W, int C, int opad, hps_stream_t stream) {
    if (!x || !y) return bad_arg("hps_maxpool3x3s2_pad: null pointer");
    if (C % 4 != 0) return bad_arg("hps_maxpool3x3s2_pad: C % 4 == 0 required");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)B * ((Ho + 1) / 2) * ((Wo + 1) / 2) * (C / 4);
    if (total <= 0) return HPS_OK;
    hipLaunchKernelGGL(maxpool_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W,
                       C, Ho, Wo, opad, total);
    return check_launch("hps_maxpool3x3s2_pad");
}

extern "C" int hps_global_avgpool_pad(const float* x, float* y, int B, int H, int W, int C, int P, hps_stream_t stream) {
    if (!x || !y) return bad_arg("hps_global_avgpool_pad: null pointer");
    const int total = B * C;
    if (total <= 0) return HPS_OK;
    hipLaunchKernelGGL(avgpool_pad_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, P,
                       total);
    return check_launch("hps_global_avgpool_pad");
}

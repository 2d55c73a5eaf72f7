// Winograd form of the ResNet stem (models/resnet.py:147-150, :203-206: conv1 7x7 / stride 2 / pad 3 on the 18-channel proxy
// representation, bn1, relu) on the fp32 MFMA pipe.
//
// Why: the stem is a third of the encoder (0.88 of 2.58 ms at B = 64) and its direct implicit GEMM already runs its K loop at
// 141 of the 153.7 TF/s the pipe sustains (conv_pad.hip, row mode) -- the only lever left is fewer multiplications.
//
// A stride-2 correlation splits into four stride-1 correlations on the even / odd sub-lattices of the input ("phases"):
//   y[oy, ox] = sum_{ry, rx} sum_{a, b} X_{ry,rx}[oy + a, ox + b] * w[2a + ry, 2b + rx],    X_{ry,rx}[i, j] = x[2i + ry - 3, 2j + rx - 3]
// with 4 taps per axis for the even phase (ky = 0, 2, 4, 6) and 3 for the odd one.  Each phase is computed as Winograd
// F(2x2, r x s): a 2x2 output tile from (2 + r - 1) x (2 + s - 1) inputs with as many multiplications -- 25 + 20 + 20 + 16 = 81 per
// tile and (cin, cout) pair instead of 4 * 49 = 196: 2.4 x fewer MFMAs.  F(2, 4) uses the points 0, 1, -1, 2, inf and F(2, 3) the
// points 0, 1, -1, inf: B^T and A^T hold small integers only, and in fp32 the result is as close to the fp64 sum as the direct
// fp32 sum is (5e-7 of the output scale, tools/stem_wino/accuracy.py) -- K = 18 per position, so little rounding accumulates.
//
//   V_p[tile, c]  = (B_y^T d B_x)_p               input transform of the tile's 9 x 9 input patch (81 pixels -> 81 positions)
//   M_p[co, tile] = sum_c U_p[c, co] V_p[tile, c]   81 GEMMs with K = 18: nine v_mfma_f32_32x32x2_f32 per 32 x 32 block
//   Y = sum_phases A_y^T M A_x                     output transform, then bn1 + relu
//   U = G_y g G_x^T is prepared on the host (resnet.py).
//
// The shape of this problem is unusual: K is tiny, so every transformed input value feeds only 64 output channels.  Staging V
// through LDS (as conv_wino.hip does) would cost more LDS traffic than the MFMAs take; instead the lane that needs V as an MFMA
// operand computes it: MFMA lane (il, kl) supplies B[k = kl][column il], so lane (il, kl) transforms tile il's channels of
// parity kl (2m + kl for the m-th MFMA of a position) and holds them in registers -- no LDS round trip for V at all.
// The positions are processed in 18 "rows" (one row i of one phase: 5 or 4 positions).  Per row a wave
//   * transforms: per channel pair, the vertical combinations c[b] = sum_a B_y^T[i][a] d[a][b] of its patch pixels from the
//     raw window in LDS (coefficients from a constant table: 2-4 pixels per output, the next pair's pixels in flight), then
//     a_j = sum_b B_x^T[j][b] c[b] (compile-time coefficients) -- 45 operand registers;
//   * after the row's barrier (filters of the row in LDS): 45 MFMAs back to back, filters read from LDS one position ahead;
//   * folds the row into the four output accumulators: t = M A_x (compile-time), Y[a][.] += A_y^T[a][i] t.
// One run of VALU work and one run of MFMAs per row, not a fine interleave: on gfx950 VALU instructions and fp32 MFMAs of a SIMD
// exclude each other, and every switch between the two costs about 25 cycles in each direction (tools/mfma_valu_ops.hip; a
// version that hid every window read behind two MFMAs of the same wave was slower, 0.62 against 0.59 ms).  The same tool priced
// the instructions: with distinct register operands every VALU instruction, packed or not, costs about 5.7 cycles -- what counts is
// their number; forcing everything into v_pk_fma_f32 by inline asm cost registers (spills) and measured slower than what hipcc
// makes of the v2f expressions below, so they are left to it.
// Live registers: M (5 x 16), Y (4 x 16), the 45 operands -- two waves per SIMD.
//
// Work item = 8 x 8 tiles (16 x 16 output pixels) of one image x 64 output channels; a workgroup is two teams of four waves
// (2 tile halves x 2 channel halves), each team on its own item, both teams sharing the filters: LDS holds, per team, two raw
// phase windows (19 rows of 19 pixels, double buffered: the next phase arrives by LDS-DMA while this one is transformed) and, for
// the workgroup, two filter rows (5 positions x 4.5 KiB, double buffered) -- 154 KiB, one workgroup per CU, persistent.
// The input comes from hps_stem_phase_split: the four phase images of every input image as separate NHWC frames, so that a
// window row is one contiguous run and the channels of a pixel sit in the order the lanes read them.
#include <type_traits>

#include "hps_common.h"

namespace hps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int SW_C = 18;                         // input channels
constexpr int SW_CO = 64;                        // output channels
constexpr int SW_PAIR = 38;                      // floats per pixel PAIR: 2 x 18 channels + 2 floats of padding (see stem_phase_split_kernel)
constexpr int SW_SLOTS = 90;                     // 16-byte slots per window row: 19 pixels = 9.5 pairs x 152 bytes = 1440
constexpr int SW_ROWF = SW_SLOTS * 4;            // floats per window row in LDS
constexpr int SW_RAW_PIECES = 27;                // 1 KiB DMA pieces per window: 19 rows x 90 slots = 1710 slots (the tail over-reads row 19)
constexpr int SW_RAW_F = SW_RAW_PIECES * 256;    // floats per window buffer
constexpr int SW_POS_F = 1152;                   // floats of one position's filters: [co half][k 0-7 | k 8-15 | k 16-17] x 32 co
constexpr int SW_U_PIECES = 23;                  // a row of five positions = 22.5 KiB
constexpr int SW_U_F = SW_U_PIECES * 256;
constexpr int SW_ROWS = 18;
constexpr int SW_LDS_F = 4 * SW_RAW_F + 2 * SW_U_F;      // 157 696 bytes

// one row of positions: phase, number of x points, the vertical combination, the output fold
struct StemRow {
    int phase, first, nxp, ne;
    int upos;                   // index of the row's first position in the filter array
    int aoff[4];                // window rows of the vertical combination, as float offsets (a * SW_ROWF)
    float coef[4];
    float e0, e1;               // A_y^T[0][i], A_y^T[1][i]
};
// F(2, 4), points 0, 1, -1, 2, inf:  B^T = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1],  A^T = [1 1 1 1 0; 0 1 -1 2 1]
// F(2, 3), points 0, 1, -1, inf:     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1],                          A^T = [1 1 1 0; 0 1 -1 1]
// Phases in the order (even rows, even columns), (even, odd), (odd, even), (odd, odd); ne = taps of the row's vertical combination
// (unused entries repeat tap 0 with coefficient 0).
__constant__ StemRow c_stem_rows[SW_ROWS] = {
    {0, 1, 5, 4,  0, {0 * SW_ROWF, 1 * SW_ROWF, 2 * SW_ROWF, 3 * SW_ROWF}, {2.0f, -1.0f, -2.0f, 1.0f}, 1.0f, 0.0f},
    {0, 0, 5, 3,  5, {1 * SW_ROWF, 2 * SW_ROWF, 3 * SW_ROWF, 1 * SW_ROWF}, {-2.0f, -1.0f, 1.0f, 0.0f}, 1.0f, 1.0f},
    {0, 0, 5, 3, 10, {1 * SW_ROWF, 2 * SW_ROWF, 3 * SW_ROWF, 1 * SW_ROWF}, {2.0f, -3.0f, 1.0f, 0.0f}, 1.0f, -1.0f},
    {0, 0, 5, 2, 15, {1 * SW_ROWF, 3 * SW_ROWF, 1 * SW_ROWF, 1 * SW_ROWF}, {-1.0f, 1.0f, 0.0f, 0.0f}, 1.0f, 2.0f},
    {0, 0, 5, 4, 20, {1 * SW_ROWF, 2 * SW_ROWF, 3 * SW_ROWF, 4 * SW_ROWF}, {2.0f, -1.0f, -2.0f, 1.0f}, 0.0f, 1.0f},
    {1, 1, 4, 4, 25, {0 * SW_ROWF, 1 * SW_ROWF, 2 * SW_ROWF, 3 * SW_ROWF}, {2.0f, -1.0f, -2.0f, 1.0f}, 1.0f, 0.0f},
    {1, 0, 4, 3, 29, {1 * SW_ROWF, 2 * SW_ROWF, 3 * SW_ROWF, 1 * SW_ROWF}, {-2.0f, -1.0f, 1.0f, 0.0f}, 1.0f, 1.0f},
    {1, 0, 4, 3, 33, {1 * SW_ROWF, 2 * SW_ROWF, 3 * SW_ROWF, 1 * SW_ROWF}, {2.0f, -3.0f, 1.0f, 0.0f}, 1.0f, -1.0f},
    {1, 0, 4, 2, 37, {1 * SW_ROWF, 3 * SW_ROWF, 1 * SW_ROWF, 1 * SW_ROWF}, {-1.0f, 1.0f, 0.0f, 0.0f}, 1.0f, 2.0f},
    {1, 0, 4, 4, 41, {1 * SW_ROWF, 2 * SW_ROWF, 3 * SW_ROWF, 4 * SW_ROWF}, {2.0f, -1.0f, -2.0f, 1.0f}, 0.0f, 1.0f},
    {2, 1, 5, 2, 45, {0 * SW_ROWF, 2 * SW_ROWF, 0 * SW_ROWF, 0 * SW_ROWF}, {1.0f, -1.0f, 0.0f, 0.0f}, 1.0f, 0.0f},
    {2, 0, 5, 2, 50, {1 * SW_ROWF, 2 * SW_ROWF, 1 * SW_ROWF, 1 * SW_ROWF}, {1.0f, 1.0f, 0.0f, 0.0f}, 1.0f, 1.0f},
    {2, 0, 5, 2, 55, {1 * SW_ROWF, 2 * SW_ROWF, 1 * SW_ROWF, 1 * SW_ROWF}, {-1.0f, 1.0f, 0.0f, 0.0f}, 1.0f, -1.0f},
    {2, 0, 5, 2, 60, {1 * SW_ROWF, 3 * SW_ROWF, 1 * SW_ROWF, 1 * SW_ROWF}, {-1.0f, 1.0f, 0.0f, 0.0f}, 0.0f, 1.0f},
    {3, 1, 4, 2, 65, {0 * SW_ROWF, 2 * SW_ROWF, 0 * SW_ROWF, 0 * SW_ROWF}, {1.0f, -1.0f, 0.0f, 0.0f}, 1.0f, 0.0f},
    {3, 0, 4, 2, 69, {1 * SW_ROWF, 2 * SW_ROWF, 1 * SW_ROWF, 1 * SW_ROWF}, {1.0f, 1.0f, 0.0f, 0.0f}, 1.0f, 1.0f},
    {3, 0, 4, 2, 73, {1 * SW_ROWF, 2 * SW_ROWF, 1 * SW_ROWF, 1 * SW_ROWF}, {-1.0f, 1.0f, 0.0f, 0.0f}, 1.0f, -1.0f},
    {3, 0, 4, 2, 77, {1 * SW_ROWF, 3 * SW_ROWF, 1 * SW_ROWF, 1 * SW_ROWF}, {-1.0f, 1.0f, 0.0f, 0.0f}, 0.0f, 1.0f}};

struct StemGeom {
    int fr_rowf, fr_phasef, fr_imgf;      // phase frame pitches in floats: row, phase frame, image (4 phase frames)
    int blocks_x, blocks_img, n_items;
    int out_row, out_img, opad, relu;     // output frame (B, Ho + 2 opad, Wo + 2 opad, 64)
    unsigned magic_img, magic_x;
    int in_h, in_w;                       // NCHW-fed form: the input image (B, 18, in_h, in_w)
};

__device__ __forceinline__ unsigned sw_div(unsigned n, unsigned d, unsigned magic) {
    unsigned q = __umulhi(n, magic);
    if (n - q * d >= d) ++q;
    return q;
}
static unsigned sw_magic(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / d); }

// B_x^T applied to the vertical combinations c[0..NXP): position j of the row
template <int NXP, typename T>
__device__ __forceinline__ T sw_horiz(int j, const T (&c)[5]) {
    if (NXP == 5) {
        switch (j) {
            case 0: return 2.0f * (c[0] - c[2]) + (c[3] - c[1]);
            case 1: return (c[3] - c[2]) - 2.0f * c[1];
            case 2: return (2.0f * c[1] + c[3]) - 3.0f * c[2];
            case 3: return c[3] - c[1];
            default: return 2.0f * (c[1] - c[3]) + (c[4] - c[2]);
        }
    }
    switch (j) {
        case 0: return c[0] - c[2];
        case 1: return c[1] + c[2];
        case 2: return c[2] - c[1];
        default: return c[3] - c[1];
    }
}

// One 8-byte LDS read per patch pixel and channel pair, as ds_read_b64 (volatile: hipcc would pair two of them into a
// ds_read2_b64, which the LDS serves at half the rate -- 16-lane groups, banks mod 32 -- and with the tiles' 144-byte pitch at
// half of that again)
__device__ __forceinline__ v2f sw_ld2(const float* p) {
    typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_t;
    return *(lds_v2f_t)(const __attribute__((address_space(3))) float*)p;
}

template <int AB>
__device__ __forceinline__ f32x16 sw_mfma(float a, float b, f32x16 c) {
    if (AB == 2) {
        c[0] += a * b;
        return c;
    }
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Input transform of one row of positions (see the header): A[j][m] = the lane's channel 2 m + kl of position j, m = 0..8.
// rawp: this lane's patch origin in the raw window (its channel pairs), raws: the channels 16, 17 of that pixel (odd: the lane
// keeps 17).  NXP: positions of the row, NE: window rows in its vertical combination.
// AB: profiling ablations (dev library only; the product instantiates AB = 0): 1 = patch pixels not read (constants), 2 = no MFMAs,
// 3 = no output transform, 4 = no barriers (races), 5 = no filter fragment reads, 6 = no stores
// G0, G1 (profiling, AB = 12 / 13): only the channel groups [G0, G1) are transformed, the other operands stay constants
template <int NXP, int NE, int AB, int G0 = 0, int G1 = 5>
__device__ __forceinline__ void sw_transform(const float* rawp, const float* raws, const StemRow& row, bool odd, float (&A)[5][9]) {
    const int o[4] = {row.aoff[0], row.aoff[1], row.aoff[2], row.aoff[3]};
    const float k[4] = {row.coef[0], row.coef[1], row.coef[2], row.coef[3]};
    v2f ld[2][5][4];               // [group parity][patch column][tap]: the reads of the next group are in flight while one is combined
    auto issue = [&](int g) {      // group g = channel pair 0..3, 4 = the single channels 16, 17
#pragma unroll
        for (int b = 0; b < NXP; ++b)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float* p = (g < 4 ? rawp + g * 4 : raws) + (b >> 1) * SW_PAIR + (b & 1) * SW_C + o[e];
                if (AB == 7) {       // profiling: the channel pair as two 4-byte reads (what a planar, NCHW-fed window would need)
                    typedef const volatile __attribute__((address_space(3))) float* lds_f_t;
                    const lds_f_t q = (lds_f_t)(const __attribute__((address_space(3))) float*)p;
                    ld[g & 1][b][e] = (v2f){q[0], q[1]};
                } else
                    ld[g & 1][b][e] = AB == 1 ? (v2f){(float)(unsigned)(size_t)p, 1.0f} : sw_ld2(p);
            }
    };
    if (G0 > 0 || G1 < 5) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int m = 0; m < 9; ++m) A[j][m] = (float)(j + m);
    }
    issue(G0);
#pragma unroll
    for (int g = G0; g < G1; ++g) {
        if (g + 1 < G1) issue(g + 1);
        v2f c[5];
#pragma unroll
        for (int b = 0; b < NXP; ++b) {
            v2f v = k[0] * ld[g & 1][b][0];
#pragma unroll
            for (int e = 1; e < NE; ++e) v = __builtin_elementwise_fma((v2f){k[e], k[e]}, ld[g & 1][b][e], v);
            c[b] = v;
        }
#pragma unroll
        for (int j = 0; j < NXP; ++j) {
            const v2f a = sw_horiz<NXP>(j, c);
            if (g < 4) {
                A[j][2 * g] = a.x;
                A[j][2 * g + 1] = a.y;
            } else {
                A[j][8] = odd ? a.y : a.x;
            }
        }
    }
}

// The MFMAs of one row (nine per position, filters of the position read from LDS one position ahead) and its output transform
// t = M A_x, Y[a][.] += A_y^T[a][i] t.  ub / ub1: this lane's filter fragment slots of position 0 of the row (k 0-15, k 16-17).
template <int NXP, int AB, typename Spread>
__device__ __forceinline__ void sw_gemm(const float* ub, const float* ub1, const StemRow& row, const float (&A)[5][9], f32x16 (&Y)[2][2],
                                        Spread spread) {
    f32x16 M[5];
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 u0 = AB == 5 ? make_float4(1.f, 2.f, 3.f, 4.f) : *reinterpret_cast<const float4*>(ub);
    float4 u1 = AB == 5 ? make_float4(1.f, 2.f, 3.f, 4.f) : *reinterpret_cast<const float4*>(ub + 256);
    float u2 = AB == 5 ? 1.0f : ub1[0];
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        float4 n0 = u0, n1 = u1;
        float n2 = u2;
        if (j + 1 < NXP && AB != 5) {
            n0 = *reinterpret_cast<const float4*>(ub + (j + 1) * SW_POS_F);
            n1 = *reinterpret_cast<const float4*>(ub + (j + 1) * SW_POS_F + 256);
            n2 = ub1[(j + 1) * SW_POS_F];
        }
        M[j] = sw_mfma<AB>(u0.x, A[j][0], z);
        M[j] = sw_mfma<AB>(u0.y, A[j][1], M[j]);
        M[j] = sw_mfma<AB>(u0.z, A[j][2], M[j]);
        M[j] = sw_mfma<AB>(u0.w, A[j][3], M[j]);
        M[j] = sw_mfma<AB>(u1.x, A[j][4], M[j]);
        M[j] = sw_mfma<AB>(u1.y, A[j][5], M[j]);
        M[j] = sw_mfma<AB>(u1.z, A[j][6], M[j]);
        M[j] = sw_mfma<AB>(u1.w, A[j][7], M[j]);
        M[j] = sw_mfma<AB>(u2, A[j][8], M[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sl = 2 * j; sl < (j + 1 < NXP ? 2 * j + 2 : 10); ++sl) spread(sl);       // DMA pieces of the next row / phase (see the caller)
        __builtin_amdgcn_sched_barrier(0);
        u0 = n0;
        u1 = n1;
        u2 = n2;
    }
    if (AB == 3) {
#pragma unroll
        for (int j = 0; j < NXP; ++j) Y[j & 1][(j >> 1) & 1][j] += M[j][j];
        return;
    }
    f32x16 t0, t1;
    if (NXP == 5) {
        t0 = (M[0] + M[1]) + (M[2] + M[3]);
        t1 = (M[1] - M[2]) + (2.0f * M[3] + M[4]);
    } else {
        t0 = (M[0] + M[1]) + M[2];
        t1 = (M[1] - M[2]) + M[3];
    }
    const float e0 = row.e0, e1 = row.e1;
    Y[0][0] += e0 * t0;
    Y[0][1] += e0 * t1;
    Y[1][0] += e1 * t0;
    Y[1][1] += e1 * t1;
}

// POOL: the 3 x 3 / 2 max pool that follows the stem (models/resnet.py:150, :206) is formed in the epilogue and only the pooled map is
// written (the stem's full-resolution output has no other reader).  Pooled pixel (py, px) is the maximum over rows 2 py - 1 .. 2 py + 1
// and columns 2 px - 1 .. 2 px + 1 of the stem's output, i.e. over tile (py, px) entirely, the bottom row of tile (py - 1, px), the right
// column of tile (py, px - 1) and the corner pixel of tile (py - 1, px - 1): a lane owns a tile, so its left / upper neighbours are the
// lanes il - 1 / il - 8 / il - 9 (shuffles), the wave above (through LDS) -- or another work item.  Those last terms are left out here:
// the item writes the bottom row's and right column's contributions (bottom-row / right-column maxima and corner pixels of its tile
// row 7 / tile column 7) to `side` ([item][bottom | right][8 tiles][2][64 channels]) and stem_pool_borders_kernel completes the 15
// pooled pixels of every item that touch a neighbour.  Maxima of the same values in another order: identical to hps_maxpool3x3s2_pad
// on the stem's output.  g.out_* / g.opad then describe the POOLED frame.
// NCHW: the raw phase windows are gathered from the (B, 18, H, W) network input itself -- hps_stem_phase_split and its four frames per image
// disappear.  The LDS windows keep their layout (pixel pairs of 38 floats, channels in the lanes' order), so nothing downstream changes:
// lane tt = 10 wi + pj (tt < 190) of a team owns pixel pair (wi, pj) of a window and moves it in four chunks (pixel, channel half) of 8 / 10
// floats: global loads (exec-masked where the pixel lies outside the image: it stays zero) in one row's MFMA run, the 8-byte LDS stores at the
// top of the next row.  Window p + 1 is stored during the first four rows of phase p (its first chunk is loaded in the last row of phase
// p - 1, when the buffer's previous window has been read for the last time); xf is then the NCHW tensor.
template <int AB, bool POOL = false, bool NCHW = false>
__global__ __launch_bounds__(512) void stem_wino_kernel(const float* __restrict__ xf, const float* __restrict__ u,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ y, float* __restrict__ side, const StemGeom g) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // raw[team][buf][SW_RAW_F] | filters[buf][SW_U_F]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    const int kl = lane >> 5, il = lane & 31;
    const int ty = 4 * wm + (il >> 3), tx = il & 7;                  // this lane's tile in the item's 8 x 8 block

    // raw-window DMA role: piece t = w4 + 4 k of the team's window, slot s = 64 t + lane -> (window row s / 90, 16-byte slot s % 90);
    // the lane offsets are recomputed per piece (seven registers the row loop has no room for)
    auto raw_voff = [&](int k) {
        unsigned ln = (unsigned)lane;
        asm volatile("" : "+v"(ln));                                   // recomputed where it is used: hoisted out of the loops it was spilled
        const unsigned s = 64u * (unsigned)(w4 + 4 * k) + ln;
        const unsigned wr = __umulhi(s, 0x2D82D83u);                       // s / 90 for s < 2^16 (ceil(2^32 / 90))
        const unsigned q = s - wr * SW_SLOTS;
        return (wr * (unsigned)g.fr_rowf + q * 4u) * 4u;
    };
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lds_raw = lds0 + (unsigned)(team * 2 * SW_RAW_F * 4);
    const unsigned lds_u = lds0 + (unsigned)(4 * SW_RAW_F * 4);

    auto window_src = [&](int item, int phase) -> const float* {
        const unsigned b = sw_div((unsigned)item, (unsigned)g.blocks_img, g.magic_img), rem = item - b * g.blocks_img;
        const unsigned by = sw_div(rem, (unsigned)g.blocks_x, g.magic_x), bx = rem - by * g.blocks_x;
        return xf + (size_t)b * g.fr_imgf + (size_t)phase * g.fr_phasef + (size_t)(16 * by) * g.fr_rowf + (size_t)(8 * bx) * SW_PAIR;
    };
    // piece k (0..6) of the phase window whose origin is src -> the team's buffer phase & 1; this wave moves pieces w4, w4 + 4, ...
    auto dma_raw_piece = [&](const float* src, int phase, int k) {
        if (w4 + 4 * k < SW_RAW_PIECES)
            lds_dma16(raw_voff(k), src, lds_raw + (unsigned)((phase & 1) * SW_RAW_F * 4 + (w4 + 4 * k) * 1024));
    };
    auto dma_raw = [&](int item, int phase) {
        const float* src = window_src(item, phase);
#pragma unroll
        for (int k = 0; k < 7; ++k) dma_raw_piece(src, phase, k);
    };
    // piece k (0..2) of the filters of row r -> buffer r & 1; wave w moves pieces w, w + 8, w + 16
    auto dma_filter_piece = [&](int r, int k) {
        const int pieces = c_stem_rows[r].nxp == 5 ? SW_U_PIECES : 18;
        if (wave + 8 * k < pieces)
            lds_dma16((unsigned)(lane * 16), u + (size_t)c_stem_rows[r].upos * SW_POS_F + (wave + 8 * k) * 256,
                      lds_u + (unsigned)((r & 1) * SW_U_F * 4 + (wave + 8 * k) * 1024));
    };
    auto dma_filters = [&](int r) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dma_filter_piece(r, k);
    };

    // LDS read roles
    const float* raw_team = smem + team * 2 * SW_RAW_F;
    const int patch0 = (2 * ty) * SW_ROWF + tx * SW_PAIR;                          // the tile's patch origin in a window (pixel 2 tx = pair tx)
    const float* frag0 = smem + 4 * SW_RAW_F + wn * 576 + kl * 128 + il * 4;      // filter fragment (k 0-7) of position 0, buffer 0
    const float* frag1 = smem + 4 * SW_RAW_F + wn * 576 + 512 + kl * 32 + il;     // ... (k 16-17)

    // ---- NCHW gather role (see the template comment) ----
    const int tt = tid & 255;
    const int n_wi = tt / 10, n_pj = tt - 10 * n_wi;
    const bool n_act = NCHW && tt < 190;
    const unsigned n_goff = (unsigned)((2 * n_wi * g.in_w + 4 * n_pj) * 4);      // bytes from the (item, phase) origin to pixel 0 of the pair
    const int n_lds = n_wi * SW_ROWF + n_pj * SW_PAIR;                            // the pair's float offset in a window buffer
    const long n_plane = (long)g.in_h * g.in_w;
    float lr[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // the chunk in flight: lr[2 s + e] = channel s of the chunk, pixel e of the pair
    const float* nc_origin = xf; int nc_y0 = 0, nc_x0 = 0; bool nc_border = false;        // the current item
    const float* nn_origin = xf; int nn_y0 = 0, nn_x0 = 0; bool nn_border = false;        // the next one
    auto nchw_item = [&](int it, const float*& origin, int& y0, int& x0, bool& border) {
        const unsigned b = sw_div((unsigned)it, (unsigned)g.blocks_img, g.magic_img), rem = it - b * g.blocks_img;
        const unsigned by = sw_div(rem, (unsigned)g.blocks_x, g.magic_x), bx = rem - by * g.blocks_x;
        y0 = 32 * (int)by - 3; x0 = 32 * (int)bx - 3;
        origin = xf + (long)b * SW_C * n_plane + (long)y0 * g.in_w + x0;
        border = by == 0 || bx == 0 || y0 + 38 > g.in_h || x0 + 38 > g.in_w;
    };
    const bool n_act1 = n_act && n_pj < 9;                                         // the pair's second pixel exists (19 pixels per window row)
    const float* n_base = xf;                                                      // wave-uniform: pixel 0 of pair (0, 0), first channel of the chunk
    bool n_ok0 = false, n_ok1 = false, n_zero = false;                             // the lane loads pixel 0 / 1; the chunk needs zeros where it does not
    // chunk q = channels 4 q .. 4 q + 3 (q = 3: 12 .. 17) of BOTH pixels of the pair: the two loads of a channel are neighbours in the
    // instruction stream and hit the same 128-byte lines (chunks split by pixel fetched every line twice: 0.72 against 0.69 ms)
    auto nchw_aim = [&](bool next, int phase, int q) {
        const float* origin = next ? nn_origin : nc_origin;
        const bool border = next ? nn_border : nc_border;
        n_base = origin + (long)(4 * q) * n_plane + (long)(phase >> 1) * g.in_w + (phase & 1);
        n_ok0 = n_act; n_ok1 = n_act1;
        n_zero = border;
        if (border) {
            const int yy = (next ? nn_y0 : nc_y0) + (phase >> 1) + 2 * n_wi, xx = (next ? nn_x0 : nc_x0) + (phase & 1) + 4 * n_pj;
            const bool vy = (unsigned)yy < (unsigned)g.in_h;
            n_ok0 = n_ok0 && vy && (unsigned)xx < (unsigned)g.in_w;
            n_ok1 = n_ok1 && vy && (unsigned)(xx + 2) < (unsigned)g.in_w;
        }
    };
    auto nchw_load2 = [&](int q, int sc) {                   // channel sc of the chunk aimed at, both pixels
        if (sc < (q == 3 ? 6 : 4) && AB != 8) {             // (8: profiling, no gather loads)
            if (n_zero) { lr[2 * sc] = 0.0f; lr[2 * sc + 1] = 0.0f; }
            const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(n_base + sc * n_plane) + n_goff);
            if (n_ok0) lr[2 * sc] = p[0];
            if (n_ok1) lr[2 * sc + 1] = p[2];
        }
    };
    auto nchw_store = [&](int phase, int q) {                // the chunk in lr -> the team's window buffer phase & 1 (slots: c c+2 | c+1 c+3 per four channels)
        if (AB == 9) return;                                 // (9: profiling, no window stores)
        float* d = smem + team * 2 * SW_RAW_F + (phase & 1) * SW_RAW_F + n_lds + 4 * q;
        if (n_act) {
            *reinterpret_cast<v2f*>(d + 0) = (v2f){lr[0], lr[4]};
            *reinterpret_cast<v2f*>(d + 2) = (v2f){lr[2], lr[6]};
            if (q == 3) *reinterpret_cast<v2f*>(d + 4) = (v2f){lr[8], lr[10]};       // channels 16, 17
        }
        if (n_act1) {
            *reinterpret_cast<v2f*>(d + SW_C + 0) = (v2f){lr[1], lr[5]};
            *reinterpret_cast<v2f*>(d + SW_C + 2) = (v2f){lr[3], lr[7]};
            if (q == 3) *reinterpret_cast<v2f*>(d + SW_C + 4) = (v2f){lr[9], lr[11]};
        }
    };

    int pair = blockIdx.x;
    if (2 * pair >= g.n_items) return;
    int item = min(2 * pair + team, g.n_items - 1);
    bool live = 2 * pair + team < g.n_items;
    if (NCHW) {
        nchw_item(item, nc_origin, nc_y0, nc_x0, nc_border);
        for (int q = 0; q < 4; ++q) {                        // the first window, chunk by chunk; then the first chunk of the second one
            nchw_aim(false, 0, q);
#pragma unroll
            for (int sc = 0; sc < 6; ++sc) nchw_load2(q, sc);
            nchw_store(0, q);
        }
        nchw_aim(false, 1, 0);
#pragma unroll
        for (int sc = 0; sc < 6; ++sc) nchw_load2(0, sc);
    } else {
        dma_raw(item, 0);
    }
    dma_filters(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first row's transform reads the window in front of the row's barrier
    __syncthreads();

    for (;;) {
        f32x16 Y[2][2];                // the four output pixels of the lane's tile x its 16 channels
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[a][b][r] = 0.0f;
        const int next_pair = pair + gridDim.x;
        const bool has_next = 2 * next_pair < g.n_items;
        const int next_item = min(2 * next_pair + team, g.n_items - 1);
        if (NCHW && has_next) nchw_item(next_item, nn_origin, nn_y0, nn_x0, nn_border);

        for (int r = 0; r < SW_ROWS; ++r) {
            const StemRow& row = c_stem_rows[r];
            const int phase = row.phase;
            // NCHW: row idx of its phase stores chunk idx of the NEXT window (loaded during the previous row) and loads chunk idx + 1; the
            // phase's last row loads chunk 0 of the window after next
            const int n_idx = r - (phase == 0 ? 0 : phase == 1 ? 5 : phase == 2 ? 10 : 14);
            const bool n_last = r == SW_ROWS - 1 || c_stem_rows[r + 1].first;
            int n_lq = -1;                                       // chunk to load in this row's MFMA run
            if (NCHW) {
                int l_phase = phase + 1;
                bool l_next = false;
                if (n_idx + 1 < 4) n_lq = n_idx + 1;
                else if (n_last) { n_lq = 0; l_phase = phase + 2; }
                if (l_phase > 3) { l_phase -= 4; l_next = true; if (!has_next) n_lq = -1; }
                if (n_lq >= 0) nchw_aim(l_next, l_phase, n_lq);
            }
            // ---- the row's VALU run: input transform (the window of its phase landed at least a row ago) ----
            const float* rawp = raw_team + (phase & 1) * SW_RAW_F + patch0 + 2 * kl;
            const float* raws = raw_team + (phase & 1) * SW_RAW_F + patch0 + 16;
            const bool odd = kl != 0;
            float A[5][9];
            // Profiling (dev library; results are garbage): what removing the DUPLICATE input transform could buy at most.  The waves of a
            // channel-half pair (wn = 0 / 1) transform the same tiles.  AB = 10: the wn = 1 waves skip the transform outright (operands =
            // constants) -- the upper bound, a hand-over that costs nothing.  AB = 11: they read their 45 operands from LDS instead (the
            // team's window buffer as a stand-in for an exchange buffer: 45 four-byte reads per lane) behind one more barrier per row --
            // the cheapest hand-over there could be (the wn = 0 waves' 45 stores are NOT charged).
            // AB = 12 / 13: the SPLIT form -- each wave of the pair transforms about half of the channels (wn = 0: channel pairs 0-3 of
            // the even / odd parity = groups 0, 1; wn = 1: groups 2, 3 and the singles) and would get the other half from its partner.
            // 12: no exchange at all (the upper bound of the split); 13: + 23 LDS stores, 23 LDS reads per lane and row and a second barrier
            // (stand-ins in the team's window buffer: the cheapest exchange there could be, its 2 x 23 KB of LDS not even found yet).
            if (AB == 12 || AB == 13) {
                if (row.nxp == 5) {
                    if (wn == 0) { if (row.ne == 4) sw_transform<5, 4, AB, 0, 2>(rawp, raws, row, odd, A); else if (row.ne == 3) sw_transform<5, 3, AB, 0, 2>(rawp, raws, row, odd, A); else sw_transform<5, 2, AB, 0, 2>(rawp, raws, row, odd, A); }
                    else { if (row.ne == 4) sw_transform<5, 4, AB, 2, 5>(rawp, raws, row, odd, A); else if (row.ne == 3) sw_transform<5, 3, AB, 2, 5>(rawp, raws, row, odd, A); else sw_transform<5, 2, AB, 2, 5>(rawp, raws, row, odd, A); }
                } else {
                    if (wn == 0) { if (row.ne == 4) sw_transform<4, 4, AB, 0, 2>(rawp, raws, row, odd, A); else if (row.ne == 3) sw_transform<4, 3, AB, 0, 2>(rawp, raws, row, odd, A); else sw_transform<4, 2, AB, 0, 2>(rawp, raws, row, odd, A); }
                    else { if (row.ne == 4) sw_transform<4, 4, AB, 2, 5>(rawp, raws, row, odd, A); else if (row.ne == 3) sw_transform<4, 3, AB, 2, 5>(rawp, raws, row, odd, A); else sw_transform<4, 2, AB, 2, 5>(rawp, raws, row, odd, A); }
                }
                if (AB == 13) {
                    float* xq = const_cast<float*>(raw_team) + (phase & 1) * SW_RAW_F + (w4 * 23) * 64 + lane;
                    if (wn) {
#pragma unroll
                        for (int e = 0; e < 23; ++e) xq[e * 64] = A[e / 9 + 2][e % 9];
                    } else {
#pragma unroll
                        for (int e = 0; e < 23; ++e) xq[e * 64] = A[e / 9][e % 9];
                    }
                }
            } else
            if ((AB == 10 || AB == 11) && wn == 1) {
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int m = 0; m < 9; ++m) A[j][m] = AB == 10 ? (float)(j + m) : raw_team[(phase & 1) * SW_RAW_F + (j * 9 + m) * 64 + lane];
            } else
            if (row.nxp == 5) {
                if (row.ne == 4) sw_transform<5, 4, AB>(rawp, raws, row, odd, A);
                else if (row.ne == 3) sw_transform<5, 3, AB>(rawp, raws, row, odd, A);
                else sw_transform<5, 2, AB>(rawp, raws, row, odd, A);
            } else {
                if (row.ne == 4) sw_transform<4, 4, AB>(rawp, raws, row, odd, A);
                else if (row.ne == 3) sw_transform<4, 3, AB>(rawp, raws, row, odd, A);
                else sw_transform<4, 2, AB>(rawp, raws, row, odd, A);
            }
            // NCHW: the chunk loaded during the previous row's MFMA run goes to LDS here, behind the transform (its loads have had the
            // transform's time to arrive; stored at the top of the row they were waited for) and in front of the row's barrier
            if (NCHW && n_idx < 4 && (phase < 3 || has_next)) nchw_store(phase + 1, n_idx);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of row r's filters (and of the next phase window)
            if (AB != 4) __syncthreads();                        // ... everyone's; the other filter buffer and window buffer are free
            if (AB == 11) __syncthreads();                       // (profiling: the hand-over's second barrier)
            if (AB == 13) {                                      // (profiling: the exchange's second half -- read the partner's 23 operands, one more barrier)
                const float* xq = raw_team + (phase & 1) * SW_RAW_F + ((w4 ^ 1) * 23) * 64 + lane;
                if (wn) {
#pragma unroll
                    for (int e = 0; e < 23; ++e) A[e / 9][e % 9] = xq[e * 64];
                } else {
#pragma unroll
                    for (int e = 0; e < 23; ++e) A[e / 9 + 2][e % 9] = xq[e * 64];
                }
                __syncthreads();
            }
            // The DMAs of the next row's filters (and, in the first row of a phase, of the next phase's window) are issued INSIDE the
            // MFMA run, two pieces behind each position's MFMAs: in front of the run their issue (about 50 cycles a piece, up to ten
            // pieces) held the whole SIMD back while the MFMA pipe was idle.
            const int f_row = r + 1 < SW_ROWS ? r + 1 : (has_next ? 0 : -1);
            const int w_phase = phase < 3 ? phase + 1 : 0;
            const bool w_fetch = !NCHW && row.first && (phase < 3 || has_next);
            const float* w_src = xf;
            if (w_fetch) w_src = window_src(phase < 3 ? item : next_item, w_phase);      // (four rows in eighteen: two divisions)
            // the filter row's source and piece count once per row, not per piece
            const float* f_src = u + (size_t)c_stem_rows[f_row >= 0 ? f_row : 0].upos * SW_POS_F + wave * 256;
            const int f_pieces = f_row < 0 ? 0 : (c_stem_rows[f_row].nxp == 5 ? SW_U_PIECES : 18);
            const unsigned f_dst = lds_u + (unsigned)(((r + 1) & 1) * SW_U_F * 4 + wave * 1024);
            auto spread = [&](int slot) {                        // slots 0-2: filter pieces, 3-9: window pieces
                if (slot < 3) {
                    if (wave + 8 * slot < f_pieces) lds_dma16((unsigned)(lane * 16), f_src + slot * 8 * 256, f_dst + (unsigned)(slot * 8 * 1024));
                } else if (NCHW) {
                    if (n_lq >= 0 && slot < 9) nchw_load2(n_lq, slot - 3);       // (issued in a burst at the head of the run instead: 0.70 against 0.69 ms)
                } else if (w_fetch) {
                    dma_raw_piece(w_src, w_phase, slot - 3);
                }
            };
            // ---- the row's MFMA run, then its output transform (the head of the next VALU run) ----
            const float* ub = frag0 + (r & 1) * SW_U_F;
            const float* ub1 = frag1 + (r & 1) * SW_U_F;
            if (row.nxp == 5) sw_gemm<5, AB>(ub, ub1, row, A, Y, spread);
            else sw_gemm<4, AB>(ub, ub1, row, A, Y, spread);
        }

        // ---- bn1 + relu, stores: lane = one tile (MFMA column), register quad q = channels wn 32 + 8 q + 4 kl .. + 3 ----
        if (AB == 6) {                 // profiling: results kept alive, nothing stored
            float t = 0.0f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += Y[a][bb][r];
            if (t == 12345.678f) y[0] = t;
        } else if (POOL) {
            const int co = wn * 32 + 4 * kl;
            float4 sc[4], sh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sc[q] = *reinterpret_cast<const float4*>(scale + co + 8 * q);
                sh[q] = *reinterpret_cast<const float4*>(shift + co + 8 * q);
            }
            const int trow = il >> 3;                                    // the tile's row inside the wave's four
            const bool has_left = tx > 0, up_in_wave = trow > 0;
            // exchange buffer of the team: its window buffer 1 (phase 3's window, read for the last time before row 17's barrier; the next
            // DMA into it is issued behind the next item's first barrier): [wn][kl][bottom maxima | corners][16 channels][8 tiles]
            float* xb = smem + (team * 2 + 1) * SW_RAW_F + ((wn * 2 + kl) * 2) * 128;
            float P[16], Bt[16], R[16], Cn[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = r >> 2, e = r & 3;
                const float s1 = e == 0 ? sc[q].x : e == 1 ? sc[q].y : e == 2 ? sc[q].z : sc[q].w;
                const float h1 = e == 0 ? sh[q].x : e == 1 ? sh[q].y : e == 2 ? sh[q].z : sh[q].w;
                float v00 = Y[0][0][r] * s1 + h1, v01 = Y[0][1][r] * s1 + h1, v10 = Y[1][0][r] * s1 + h1, v11 = Y[1][1][r] * s1 + h1;
                if (g.relu) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
                Bt[r] = fmaxf(v10, v11);
                R[r] = fmaxf(v01, v11);
                Cn[r] = v11;
                P[r] = fmaxf(fmaxf(v00, v01), Bt[r]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rl = __shfl_up(R[r], 1), bu = __shfl_up(Bt[r], 8), cu = __shfl_up(Cn[r], 9);
                if (has_left) P[r] = fmaxf(P[r], rl);
                if (up_in_wave) {
                    P[r] = fmaxf(P[r], bu);
                    if (has_left) P[r] = fmaxf(P[r], cu);
                }
            }
            if (wm == 0 && trow == 3) {                                  // tile row 3 of the item: what tile row 4 (the wave below) needs
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    xb[r * 8 + tx] = Bt[r];
                    xb[128 + r * 8 + tx] = Cn[r];
                }
            }
            __syncthreads();
            if (wm == 1 && trow == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    P[r] = fmaxf(P[r], xb[r * 8 + tx]);
                    if (has_left) P[r] = fmaxf(P[r], xb[128 + r * 8 + tx - 1]);
                }
            }
            if (NCHW) __syncthreads();       // the next item's first row STORES into this buffer at once (the frame-fed form's DMA into it waits for a barrier)
            if (live) {
                const unsigned b = sw_div((unsigned)item, (unsigned)g.blocks_img, g.magic_img), rem = item - b * g.blocks_img;
                const unsigned by = sw_div(rem, (unsigned)g.blocks_x, g.magic_x), bx = rem - by * g.blocks_x;
                float* yp = y + (size_t)b * g.out_img + (size_t)(8 * by + ty + g.opad) * g.out_row + (size_t)(8 * bx + tx + g.opad) * SW_CO + co;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(yp + 8 * q) = make_float4(P[4 * q], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3]);
                float* sd = side + (size_t)item * 2048 + co;
                if (ty == 7) {                                           // bottom edge: [tx][maxima | corners][64]
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        *reinterpret_cast<float4*>(sd + tx * 128 + 8 * q) = make_float4(Bt[4 * q], Bt[4 * q + 1], Bt[4 * q + 2], Bt[4 * q + 3]);
                        *reinterpret_cast<float4*>(sd + tx * 128 + 64 + 8 * q) = make_float4(Cn[4 * q], Cn[4 * q + 1], Cn[4 * q + 2], Cn[4 * q + 3]);
                    }
                }
                if (tx == 7) {                                           // right edge: [ty][maxima | corners][64]
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        *reinterpret_cast<float4*>(sd + 1024 + ty * 128 + 8 * q) = make_float4(R[4 * q], R[4 * q + 1], R[4 * q + 2], R[4 * q + 3]);
                        *reinterpret_cast<float4*>(sd + 1024 + ty * 128 + 64 + 8 * q) = make_float4(Cn[4 * q], Cn[4 * q + 1], Cn[4 * q + 2], Cn[4 * q + 3]);
                    }
                }
            }
        } else if (live) {
            const unsigned b = sw_div((unsigned)item, (unsigned)g.blocks_img, g.magic_img), rem = item - b * g.blocks_img;
            const unsigned by = sw_div(rem, (unsigned)g.blocks_x, g.magic_x), bx = rem - by * g.blocks_x;
            const int co = wn * 32 + 4 * kl;
            float* yp = y + (size_t)b * g.out_img + (size_t)(16 * by + 2 * ty + g.opad) * g.out_row +
                        (size_t)(16 * bx + 2 * tx + g.opad) * SW_CO + co;
            float4 sc[4], sh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sc[q] = *reinterpret_cast<const float4*>(scale + co + 8 * q);
                sh[q] = *reinterpret_cast<const float4*>(shift + co + 8 * q);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = make_float4(Y[a][bb][4 * q] * sc[q].x + sh[q].x, Y[a][bb][4 * q + 1] * sc[q].y + sh[q].y,
                                               Y[a][bb][4 * q + 2] * sc[q].z + sh[q].z, Y[a][bb][4 * q + 3] * sc[q].w + sh[q].w);
                        if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        *reinterpret_cast<float4*>(yp + (size_t)a * g.out_row + bb * SW_CO + 8 * q) = v;
                    }
        }
        if (!has_next) break;
        pair = next_pair;
        item = next_item;
        nc_origin = nn_origin; nc_y0 = nn_y0; nc_x0 = nn_x0; nc_border = nn_border;
        live = 2 * pair + team < g.n_items;
    }
}

// Pixels w0 .. w0 + n - 1 (w0 even) of image row h of image b, staged in LDS as sp[pixel][18 channel slots] (slot order of the
// frames: 0 2 1 3 | 4 6 5 7 | ... | 16 17), -> the two phase frames the row belongs to.  t = thread of a 256-thread workgroup.
__device__ __forceinline__ void phase_row_store(const float* sp, float* __restrict__ frames, long b, int h, int w0, int n, int H, int W, int t) {
    constexpr int C = SW_C;
    const int FR = H / 2 + 4, FC = W / 2 + 4;
    const int ry = (h + 3) & 1, i = (h + 3) >> 1;
    const v2f* s2 = reinterpret_cast<const v2f*>(sp);
#pragma unroll
    for (int rx = 0; rx < 2; ++rx) {
        // pixels w0 + wl with (w0 + wl + 3) & 1 == rx (w0 is even): wl = 1 - rx, 3 - rx, ...
        const int first = 1 - rx, count = (n - first + 1) / 2;
        const int j0 = (w0 + first + 3) >> 1;
        v2f* row = reinterpret_cast<v2f*>(frames + ((b * 4 + ry * 2 + rx) * FR + i) * (long)(FC / 2) * SW_PAIR);
        // ten 8-byte words per pixel: its nine channel pairs and, behind the second pixel of a pair, the padding (written as zero
        // although it never changes: a run that leaves 8-byte holes in its 64-byte sectors makes the memory side read-modify-write
        // them -- 0.17 instead of 0.11 ms)
        for (int e = t; e < count * 10; e += 256) {
            const int pi = e / 10, sl = e - pi * 10;
            const int j = j0 + pi;                                   // pixel of the phase row: pair j / 2, 19 8-byte words per pair
            if (sl < C / 2) row[(j >> 1) * (SW_PAIR / 2) + (j & 1) * (C / 2) + sl] = s2[(first + 2 * pi) * (C / 2) + sl];
            else if (j & 1) row[(j >> 1) * (SW_PAIR / 2) + C] = (v2f){0.0f, 0.0f};
        }
    }
}

// (B, 18, H, W) -> the four phase frames of every image: frames[b][2 ry + rx][i][j][18] = x[b][:, 2 i + ry - 3, 2 j + rx - 3],
// frame = (H / 2 + 4) x (W / 2 + 4) pixels (out-of-image pixels stay zero: the owner zeroes the buffer once).  The 18 channels
// of a pixel are stored in the order 0 2 1 3 | 4 6 5 7 | 8 10 9 11 | 12 14 13 15 | 16 17: the MFMA lane of parity kl reads the
// pairs (kl, kl + 2), (kl + 4, kl + 6), ... as 8-byte words.  Two pixels are followed by two floats of padding (38 floats per
// pixel pair): the tiles of a wave start at even pixels, and with 36 floats between them every ds_read_b64 of the stem kernel
// reached only half of the LDS banks (2-way conflicts, SQ_LDS_BANK_CONFLICT = 42 % of the LDS cycles); with 38 the 32 lanes of a
// read group cover the 64 banks exactly once.  A workgroup moves one run of up to 256 pixels of an input row.
__global__ __launch_bounds__(256) void stem_phase_split_kernel(const float* __restrict__ x, float* __restrict__ frames, int H, int W,
                                                               int runs_per_row) {
    constexpr int C = SW_C;
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
        for (int c = 0; c < C; ++c) {
            const int slot = c >= 16 ? c : (c & ~3) + ((c & 1) << 1) + ((c >> 1) & 1);      // 0 2 1 3 within each group of four
            sp[t * C + slot] = src[c * hw];
        }
    }
    __syncthreads();
    phase_row_store(sp, frames, b, h, w0, n, H, W, t);
}

// The proxy representation (predict/predict_poseMF_shapeGaussian_net.py:93-100: channel 0 = edge map, channels 1..17 = visibility-masked
// Gaussian heat-maps, utils/label_conversions.py:105-124) written STRAIGHT into the four phase frames of the Winograd stem: what
// hps_proxy_rep followed by hps_stem_phase_split produces, bit for bit, without the (B,18,H,W) tensor in between (-302 MB written,
// -302 MB read per 64 images: the front end generates every input pixel anyway -- VERDICT r4 item 3b).  The Gaussian is evaluated with
// proxy_rep_kernel's operations on the same operands (row term once per workgroup and row, column term once per thread and joint).
// A workgroup = FRAME_ROWS rows of up to 256 columns of one image; thread = column.
constexpr int FRAME_ROWS = 8;
__global__ __launch_bounds__(256) void proxy_rep_frames_kernel(const float* __restrict__ edge, const float* __restrict__ joints2d,
                                                               const float* __restrict__ visib, float* __restrict__ frames, int H, int W,
                                                               float std, int runs_per_row) {
    constexpr int C = SW_C, K = SW_C - 1;
    __shared__ float sp[256 * C];
    __shared__ float s_row[FRAME_ROWS][K];           // ((y - v) / std)^2 / 2
    __shared__ float s_vis[K];
    const int t = threadIdx.x;
    const int run = blockIdx.x % runs_per_row;
    const int y0 = (blockIdx.x / runs_per_row) * FRAME_ROWS;
    const long b = blockIdx.y;
    const int w0 = run * 256, n = min(256, W - w0), x = w0 + t;
    for (int i = t; i < FRAME_ROWS * K; i += 256) {
        const int r = i / K, k = i - r * K;
        const float v = joints2d[((size_t)b * K + k) * 2 + 1];
        const float a = ((float)(y0 + r) - v) / std;
        s_row[r][k] = (a * a) / 2.0f;
    }
    if (t < K) s_vis[t] = visib ? visib[(size_t)b * K + t] : 1.0f;
    float c2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float u = joints2d[((size_t)b * K + k) * 2 + 0];
        const float c = ((float)x - u) / std;
        c2[k] = (c * c) / 2.0f;
    }
    const int rows = min(FRAME_ROWS, H - y0);
    for (int r = 0; r < rows; ++r) {
        __syncthreads();                              // s_row / s_vis written (first round); sp free again (later rounds)
        if (t < n) {
            sp[t * C + 0] = edge ? edge[((size_t)b * H + y0 + r) * W + x] : 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int c = k + 1;
                const int slot = c >= 16 ? c : (c & ~3) + ((c & 1) << 1) + ((c >> 1) & 1);
                const float hv = expf(-s_row[r][k] - c2[k]);
                sp[t * C + slot] = visib ? hv * s_vis[k] : hv;
            }
        }
        __syncthreads();
        phase_row_store(sp, frames, b, y0 + r, w0, n, H, W, t);
    }
}


// Second pass of the pooled stem: the pooled pixels of an item that lie on its top row or left column (15 of 64) also see the neighbouring
// items' bottom rows / right columns (`side`, written by stem_wino_kernel<., true>).  Thread = (item, one of the 15 pixels, four channels).
__global__ __launch_bounds__(256) void stem_pool_borders_kernel(float* __restrict__ y, const float* __restrict__ side, const StemGeom g, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cq = (int)(i & 15);
    long p = i >> 4;
    const int j = (int)(p % 15);
    const int item = (int)(p / 15);
    const int ty = j < 8 ? 0 : j - 7, tx = j < 8 ? j : 0;
    const unsigned b = sw_div((unsigned)item, (unsigned)g.blocks_img, g.magic_img), rem = item - b * g.blocks_img;
    const unsigned by = sw_div(rem, (unsigned)g.blocks_x, g.magic_x), bx = rem - by * g.blocks_x;
    if (by == 0 && bx == 0) return;                                      // nothing above, nothing to the left
    float4* yp = reinterpret_cast<float4*>(y + (size_t)b * g.out_img + (size_t)(8 * by + ty + g.opad) * g.out_row +
                                           (size_t)(8 * bx + tx + g.opad) * SW_CO + cq * 4);
    float4 v = *yp;
    auto take = [&](int it, int edge, int pos, int val) {
        const float4 o = *reinterpret_cast<const float4*>(side + (size_t)it * 2048 + edge * 1024 + pos * 128 + val * 64 + cq * 4);
        v = make_float4(fmaxf(v.x, o.x), fmaxf(v.y, o.y), fmaxf(v.z, o.z), fmaxf(v.w, o.w));
    };
    if (ty == 0) {
        if (by > 0) {
            take(item - g.blocks_x, 0, tx, 0);                           // bottom-row maxima of the tile above
            if (tx > 0) take(item - g.blocks_x, 0, tx - 1, 1);           // corner pixel of the tile above and to the left
        }
        if (tx == 0 && bx > 0) {
            take(item - 1, 1, 0, 0);                                     // right-column maxima of the tile to the left
            if (by > 0) take(item - g.blocks_x - 1, 0, 7, 1);            // corner pixel of the diagonal item's last tile
        }
    } else if (bx > 0) {
        take(item - 1, 1, ty, 0);
        take(item - 1, 1, ty - 1, 1);
    }
    *yp = v;
}

}  // namespace hps

using namespace hps;

extern "C" size_t hps_stem_phase_frames_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || (H % 32) || (W % 32)) return 0;
    // + one window of slack: the last DMA piece of a window over-reads into the rows that follow
    const size_t rowf = (size_t)(W / 2 + 4) / 2 * SW_PAIR;
    return ((size_t)B * 4 * (H / 2 + 4) * rowf + 20 * rowf) * sizeof(float);
}

extern "C" int hps_stem_phase_split(const float* x, float* frames, int B, int C, int H, int W, hps_stream_t stream) {
    if (!x || !frames) return bad_arg("hps_stem_phase_split: null pointer");
    if (C != SW_C) return bad_arg("hps_stem_phase_split: 18 input channels (the proxy representation)");
    if (H <= 0 || W <= 0 || (H % 32) || (W % 32)) return bad_arg("hps_stem_phase_split: H and W must be multiples of 32");
    if (B <= 0) return HPS_OK;
    const int runs = ceil_div(W, 256);
    const long blocks = (long)B * H * runs;
    if (blocks > 0x7fffffffL) return bad_arg("hps_stem_phase_split: too many rows");
    hipLaunchKernelGGL(stem_phase_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, frames, H, W, runs);
    return check_launch("hps_stem_phase_split");
}

extern "C" int hps_proxy_rep_phase_frames(const float* edge, const float* joints2d, const float* visib, float* frames, int B, int K,
                                          int H, int W, float std, hps_stream_t stream) {
    if (!joints2d || !frames) return bad_arg("hps_proxy_rep_phase_frames: null pointer");
    if (K != SW_C - 1) return bad_arg("hps_proxy_rep_phase_frames: 17 joints (the 18-channel proxy representation)");
    if (H <= 0 || W <= 0 || (H % 32) || (W % 32)) return bad_arg("hps_proxy_rep_phase_frames: H and W must be multiples of 32");
    if (B <= 0) return HPS_OK;
    if (B > 65535) return bad_arg("hps_proxy_rep_phase_frames: at most 65 535 images per call");
    const int runs = ceil_div(W, 256);
    hipLaunchKernelGGL(proxy_rep_frames_kernel, dim3((unsigned)(runs * ceil_div(H, FRAME_ROWS)), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       edge, joints2d, visib, frames, H, W, std, runs);
    return check_launch("hps_proxy_rep_phase_frames");
}

static int stem_wino_launch(const float* frames, const float* u, const float* scale, const float* shift, float* y, int B, int H,
                            int W, int opad, int relu, int ablate, hps_stream_t stream, float* side = nullptr, bool nchw = false) {
    const bool pool = side != nullptr;
    if (!frames || !u || !scale || !shift || !y) return bad_arg("hps_stem_winograd: null pointer");
    if (H <= 0 || W <= 0 || (H % 32) || (W % 32)) return bad_arg("hps_stem_winograd: H and W must be multiples of 32 (8 x 8 blocks of 2 x 2-pixel tiles at stride 2)");
    if (opad < 0) return bad_arg("hps_stem_winograd: opad");
    if (B <= 0) return HPS_OK;
    const int Ho = H / 2, Wo = W / 2;
    StemGeom g;
    g.fr_rowf = (Wo + 4) / 2 * SW_PAIR;
    g.fr_phasef = (Ho + 4) * g.fr_rowf;
    g.fr_imgf = 4 * g.fr_phasef;
    g.blocks_x = Wo / 16;
    g.blocks_img = (Ho / 16) * g.blocks_x;
    g.n_items = B * g.blocks_img;
    g.out_row = ((pool ? Wo / 2 : Wo) + 2 * opad) * SW_CO;       // pool: the frame of the pooled map
    g.out_img = ((pool ? Ho / 2 : Ho) + 2 * opad) * g.out_row;
    g.opad = opad;
    g.relu = relu;
    g.magic_img = sw_magic((unsigned)g.blocks_img);
    g.magic_x = sw_magic((unsigned)g.blocks_x);
    g.in_h = H; g.in_w = W;
    if ((size_t)B * g.fr_imgf * 4 >= 0xffffffffull || (size_t)B * g.out_img * 4 >= 0xffffffffull)
        return bad_arg("hps_stem_winograd: tensor exceeds the 32-bit lane offsets");
    const size_t lds = (size_t)SW_LDS_F * sizeof(float);
    const int pairs = (g.n_items + 1) / 2;
    int rc = HPS_OK;
    auto launch = [&](auto AB) {
        constexpr int ab = decltype(AB)::value;
        if ((rc = grant_lds<&stem_wino_kernel<ab>>((int)lds, "hps_stem_winograd")) != HPS_OK) return;
        hipLaunchKernelGGL((stem_wino_kernel<ab>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                           frames, u, scale, shift, y, (float*)nullptr, g);
    };
    if (pool) {
        if (ablate != 0 && !(nchw && ablate >= 8 && ablate <= 13)) return bad_arg("hps_stem_winograd_pooled: no ablations");
        if (nchw) {
#ifdef HPS_DEV_BUILD
            if (ablate == 12 || ablate == 13) {              // profiling: the input transform SPLIT between the waves of a channel-half pair
                if (ablate == 12) {
                    if ((rc = grant_lds<&stem_wino_kernel<12, true, true>>((int)lds, "hps_dev_stem_winograd_pooled_nchw")) != HPS_OK) return rc;
                    hipLaunchKernelGGL((stem_wino_kernel<12, true, true>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                                       frames, u, scale, shift, y, side, g);
                } else {
                    if ((rc = grant_lds<&stem_wino_kernel<13, true, true>>((int)lds, "hps_dev_stem_winograd_pooled_nchw")) != HPS_OK) return rc;
                    hipLaunchKernelGGL((stem_wino_kernel<13, true, true>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                                       frames, u, scale, shift, y, side, g);
                }
                return check_launch("hps_dev_stem_winograd_pooled_nchw");
            }
            if (ablate == 10 || ablate == 11) {              // profiling: the duplicate input transform removed (see the kernel)
                if (ablate == 10) {
                    if ((rc = grant_lds<&stem_wino_kernel<10, true, true>>((int)lds, "hps_dev_stem_winograd_pooled_nchw")) != HPS_OK) return rc;
                    hipLaunchKernelGGL((stem_wino_kernel<10, true, true>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                                       frames, u, scale, shift, y, side, g);
                } else {
                    if ((rc = grant_lds<&stem_wino_kernel<11, true, true>>((int)lds, "hps_dev_stem_winograd_pooled_nchw")) != HPS_OK) return rc;
                    hipLaunchKernelGGL((stem_wino_kernel<11, true, true>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                                       frames, u, scale, shift, y, side, g);
                }
                return check_launch("hps_dev_stem_winograd_pooled_nchw");
            }
            if (ablate == 8 || ablate == 9) {                // profiling: the gather without its loads / without its LDS stores (results are garbage)
                if (ablate == 8) {
                    if ((rc = grant_lds<&stem_wino_kernel<8, true, true>>((int)lds, "hps_dev_stem_winograd_pooled_nchw")) != HPS_OK) return rc;
                    hipLaunchKernelGGL((stem_wino_kernel<8, true, true>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                                       frames, u, scale, shift, y, side, g);
                } else {
                    if ((rc = grant_lds<&stem_wino_kernel<9, true, true>>((int)lds, "hps_dev_stem_winograd_pooled_nchw")) != HPS_OK) return rc;
                    hipLaunchKernelGGL((stem_wino_kernel<9, true, true>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                                       frames, u, scale, shift, y, side, g);
                }
                return check_launch("hps_dev_stem_winograd_pooled_nchw");
            }
#endif
            if ((rc = grant_lds<&stem_wino_kernel<0, true, true>>((int)lds, "hps_stem_winograd_pooled_nchw")) != HPS_OK) return rc;
            hipLaunchKernelGGL((stem_wino_kernel<0, true, true>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                               frames, u, scale, shift, y, side, g);
        } else {
            if ((rc = grant_lds<&stem_wino_kernel<0, true>>((int)lds, "hps_stem_winograd_pooled")) != HPS_OK) return rc;
            hipLaunchKernelGGL((stem_wino_kernel<0, true>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), lds, (hipStream_t)stream,
                               frames, u, scale, shift, y, side, g);
        }
        if ((rc = check_launch("hps_stem_winograd_pooled")) != HPS_OK) return rc;
        const long total = (long)g.n_items * 15 * 16;
        hipLaunchKernelGGL(stem_pool_borders_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, side, g, total);
        return check_launch("hps_stem_winograd_pooled (borders)");
    }
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
#endif
        default: return bad_arg("hps_stem_winograd: ablate");
    }
    if (rc != HPS_OK) return rc;
    return check_launch("hps_stem_winograd");
}

extern "C" int hps_stem_winograd(const float* frames, const float* u, const float* scale, const float* shift, float* y, int B, int H,
                                 int W, int opad, int relu, hps_stream_t stream) {
    return stem_wino_launch(frames, u, scale, shift, y, B, H, W, opad, relu, 0, stream);
}

extern "C" size_t hps_stem_pool_side_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || (H % 32) || (W % 32)) return 0;
    return (size_t)B * (H / 32) * (W / 32) * 2048 * sizeof(float);
}

extern "C" int hps_stem_winograd_pooled(const float* frames, const float* u, const float* scale, const float* shift, float* pooled, float* side,
                                        int B, int H, int W, int opad, int relu, hps_stream_t stream) {
    if (!side) return bad_arg("hps_stem_winograd_pooled: null pointer");
    return stem_wino_launch(frames, u, scale, shift, pooled, B, H, W, opad, relu, 0, stream, side);
}

extern "C" int hps_stem_winograd_pooled_nchw(const float* x, const float* u, const float* scale, const float* shift, float* pooled, float* side,
                                             int B, int H, int W, int opad, int relu, hps_stream_t stream) {
    if (!side) return bad_arg("hps_stem_winograd_pooled_nchw: null pointer");
    if ((size_t)B * SW_C * H * W * 4 >= 0x7fffffffull * 4) return bad_arg("hps_stem_winograd_pooled_nchw: input too large");
    return stem_wino_launch(x, u, scale, shift, pooled, B, H, W, opad, relu, 0, stream, side, true);
}

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_stem_winograd_pooled_nchw(const float* x, const float* u, const float* scale, const float* shift, float* pooled, float* side,
                                                 int B, int H, int W, int opad, int relu, int ablate, hps_stream_t stream) {
    return stem_wino_launch(x, u, scale, shift, pooled, B, H, W, opad, relu, ablate, stream, side, true);
}
extern "C" int hps_dev_stem_winograd(const float* frames, const float* u, const float* scale, const float* shift, float* y, int B,
                                     int H, int W, int opad, int relu, int ablate, hps_stream_t stream) {
    return stem_wino_launch(frames, u, scale, shift, y, B, H, W, opad, relu, ablate, stream);
}
#endif

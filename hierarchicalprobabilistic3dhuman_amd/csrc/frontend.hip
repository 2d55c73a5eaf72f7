// Proxy-representation front end: Canny edge detection
// (models/canny_edge_detector.py:104-166) and 2D-joint Gaussian heat-maps
// (utils/label_conversions.py:105-124), i.e. what turns an RGB crop + 17 keypoints into the 18-channel network
// input (predict/predict_poseMF_shapeGaussian_net.py:88-100).  Pure stencil / element-wise work: HBM bound.
#include <type_traits>

#include "hps_common.h"

namespace hps {

constexpr int CT = 32;                 // output tile edge
constexpr int G = 5, GH = G / 2;       // Gaussian taps (DATA.EDGE_GAUSSIAN_SIZE = 5)
// Tile frame: logical column c = 0..39 <-> image x = x0 - 4 + c, row r = 0..39 <-> y = y0 - 4 + r (x0, y0: the tile's origin).
//   input          c, r in [0, 40)
//   horizontal     c in [2, 38), every row          (Gaussian along x)
//   blurred        c, r in [2, 38)                  (Gaussian along y)
//   gradient / mag c, r in [3, 37)                  (Sobel; one pixel of halo for the non-max suppression)
//   outputs        c, r in [4, 36)
// Every stage works on aligned GROUPS of four columns (group g = columns 4 g .. 4 g + 3, g = 0..9): one 16-byte LDS word per
// item, index arithmetic and border tests once per four pixels.  LDS rows are 48 floats: column c at float c + 4, so that the
// groups g - 1 and g + 1 an item also reads exist for g = 0 and g = 9 (padding, zeroed once; values computed from it are
// never used by a valid output).
constexpr int TF = CT + 8;             // 40: frame edge
constexpr int TG = TF / 4;             // 10 groups per row
constexpr int TP = TF + 8;             // 48: LDS row pitch in floats

// Orientation bin k = round((atan2(gy, gx) * 180 / pi + 180) / 45) in 0..8 (models/canny_edge_detector.py:128-129; the
// reference's float value is 45 k) WITHOUT evaluating atan2: the bin is the 45-degree sector (gx, gy) lies in, decided by
// comparing |gy| with tan(22.5) |gx| and tan(67.5) |gx| and by the two sign bits (atan2's conventions for signed zeros
// included: atan2(+0, -x) = pi -> 8, atan2(-0, -x) = -pi -> 0, atan2(+-0, +0) = 0 -> 4).  A gradient within 1e-4 (relative) of a
// sector boundary -- where the reference's own answer hangs on atan2f's last bit -- takes the reference's formula.
// torch evaluates `atan2(...) * (180.0 / np.pi)` with the double 180 / pi rounded ONCE to fp32 (57.29578f); 180.0f / (float)pi
// is 57.295776f, one unit in the last place lower -- and this fallback exists precisely for last-bit cases.
constexpr float RAD2DEG = 57.29577951308232f;
__device__ __forceinline__ int orientation_bin(float gx, float gy) {
    const float ax = fabsf(gx), ay = fabsf(gy);
    const float T1 = 0.41421356237309503f, T2 = 2.4142135623730951f;      // tan(22.5 deg), tan(67.5 deg)
    const float b1 = T1 * ax, b2 = T2 * ax;
    if (__builtin_expect(fabsf(ay - b1) <= 1e-4f * b1 || fabsf(ay - b2) <= 1e-4f * b2, 0)) {
        if (!(ax == 0.0f && ay == 0.0f)) {
            const float ori = atan2f(gy, gx) * RAD2DEG + 180.0f;
            return (int)rintf(ori / 45.0f);
        }
    }
    // k = 4 +- d: d = 4 [gx < 0] in the horizontal sector (k = 4, 8, 0), 2 in the vertical one (6, 2), 1 + 2 [gx < 0] on the
    // diagonals (5, 7 / 3, 1); the sign is that of gy (sign BITS: atan2's conventions for signed zeros)
    const int sx = (int)(__builtin_bit_cast(unsigned, gx) >> 31), sy = (int)(__builtin_bit_cast(unsigned, gy) >> 31);
    const int d = ay <= b1 ? 4 * sx : (ay >= b2 ? 2 : 1 + 2 * sx);
    return 4 + (1 - 2 * sy) * d;
}

// The same decision for the row-marching kernel, in two straight-line pieces: the sector arithmetic for every pixel (selects only,
// no divergent control flow -- a wave that owns a whole strip runs one per SIMD, so every exec-mask region costs issue slots nothing
// hides), and `near`: the pixel must take the reference's formula instead (orientation_bin_slow), which the caller evaluates under ONE
// branch per group of four pixels.  Same conditions and the same results as orientation_bin.
__device__ __forceinline__ int orientation_bin_fast(float gx, float gy, bool& near) {
    const float ax = fabsf(gx), ay = fabsf(gy);
    const float T1 = 0.41421356237309503f, T2 = 2.4142135623730951f;
    const float b1 = T1 * ax, b2 = T2 * ax;
    const bool on1 = fabsf(ay - b1) <= 1e-4f * b1, on2 = fabsf(ay - b2) <= 1e-4f * b2, zero = ax == 0.0f && ay == 0.0f;
    near = (on1 | on2) & !zero;
    const int sx = (int)(__builtin_bit_cast(unsigned, gx) >> 31), sy = (int)(__builtin_bit_cast(unsigned, gy) >> 31);
    const int d_h = 4 * sx, d_d = 1 + 2 * sx;
    const bool h = ay <= b1, v = ay >= b2;
    const int d_vd = v ? 2 : d_d;
    const int d = h ? d_h : d_vd;
    return 4 + (1 - 2 * sy) * d;
}
__device__ __forceinline__ int orientation_bin_slow(float gx, float gy) {
    const float ori = atan2f(gy, gx) * RAD2DEG + 180.0f;
    return (int)rintf(ori / 45.0f);
}

// One workgroup = one 32x32 tile of one image.  Every stage reproduces the zero padding of the reference's
// chain of nn.Conv2d calls: each convolution sees zeros outside the IMAGE, not outside the tile.
// The earlier version of this kernel worked pixel by pixel (one LDS word, one index decode and one border test per value): about
// 2 800 instructions per thread and tile, instruction-bound at 0.092 ms for 64 crops; by groups of four it is about a quarter of that.
__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__global__ __launch_bounds__(256) void canny_kernel(const float* __restrict__ img, float g0, float g1, float g2,
                                                    float g3, float g4, float* __restrict__ blurred,
                                                    float* __restrict__ grad_mag, float* __restrict__ grad_ori,
                                                    float* __restrict__ thr_mag, float* __restrict__ thin,
                                                    float* __restrict__ thr_thin, float* __restrict__ edge_out,
                                                    size_t edge_batch_stride, int C, int H, int W,
                                                    float threshold, int nms) {
    __shared__ __attribute__((aligned(16))) float sIn[TF * TP];      // input tile; the magnitudes after the channel loop
    __shared__ __attribute__((aligned(16))) float sH[TF * TP];
    __shared__ __attribute__((aligned(16))) float sBl[TF * TP];
    const float gk[G] = {g0, g1, g2, g3, g4};
    const int b = blockIdx.z, y0 = blockIdx.y * CT, x0 = blockIdx.x * CT;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W;
    const bool vec = (W & 3) == 0;                                    // rows are 16-byte aligned: whole groups move as float4

    for (int i = tid; i < TF * TP; i += 256) { sIn[i] = 0.f; sH[i] = 0.f; sBl[i] = 0.f; }

    // Sobel items (r = 3 + i / 10, g = i % 10), i = tid and tid + 256 (< 340): the gradients of an item's four pixels stay in
    // this thread's registers over the channel loop and into the last stage
    float gx[2][4], gy[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { gx[k][e] = 0.f; gy[k][e] = 0.f; }

    for (int c = 0; c < C; ++c) {
        const float* src = img + ((size_t)b * C + c) * plane;
        __syncthreads();
        // ---- input tile, zero outside the image ----
        for (int i = tid; i < TF * TG; i += 256) {
            const int r = i / TG, g = i - r * TG;
            const int y = y0 - 4 + r, x = x0 - 4 + 4 * g;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < H) {
                const float* row = src + (size_t)y * W;
                if (vec && x >= 0 && x + 3 < W) {
                    v = *reinterpret_cast<const float4*>(row + x);
                } else {
                    if (x >= 0 && x < W) v.x = row[x];
                    if (x + 1 >= 0 && x + 1 < W) v.y = row[x + 1];
                    if (x + 2 >= 0 && x + 2 < W) v.z = row[x + 2];
                    if (x + 3 >= 0 && x + 3 < W) v.w = row[x + 3];
                }
            }
            *reinterpret_cast<float4*>(&sIn[r * TP + 4 * g + 4]) = v;
        }
        __syncthreads();
        // ---- horizontal Gaussian (:118); it only exists inside the image ----
        for (int i = tid; i < TF * TG; i += 256) {
            const int r = i / TG, g = i - r * TG;
            const float* p = &sIn[r * TP + 4 * g];
            const float4 a = lds4(p), m = lds4(p + 4), z = lds4(p + 8);
            const float v[12] = {a.x, a.y, a.z, a.w, m.x, m.y, m.z, m.w, z.x, z.y, z.z, z.w};
            const int x = x0 - 4 + 4 * g;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < G; ++t) acc += gk[t] * v[2 + e + t];
                o[e] = (x + e >= 0 && x + e < W) ? acc : 0.f;
            }
            *reinterpret_cast<float4*>(&sH[r * TP + 4 * g + 4]) = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
        // ---- vertical Gaussian on rows 2..37; zero outside the image (that is what the Sobel convs pad with) ----
        for (int i = tid; i < (TF - 4) * TG; i += 256) {
            const int rr = i / TG, g = i - rr * TG, r = rr + 2;
            const float* p = &sH[(r - 2) * TP + 4 * g + 4];
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const float4 h = lds4(p + t * TP);
                o[0] += gk[t] * h.x; o[1] += gk[t] * h.y; o[2] += gk[t] * h.z; o[3] += gk[t] * h.w;
            }
            const int y = y0 - 4 + r, x = x0 - 4 + 4 * g;
            const bool yin = y >= 0 && y < H;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (yin && x + e >= 0 && x + e < W) ? o[e] : 0.f;
            *reinterpret_cast<float4*>(&sBl[r * TP + 4 * g + 4]) = make_float4(o[0], o[1], o[2], o[3]);
            if (blurred && yin && r >= 4 && r < 4 + CT && g >= 1 && g <= 8) {                                    // :119
                float* dst = blurred + ((size_t)b * C + c) * plane + (size_t)y * W + x;
                if (vec && x + 3 < W) {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (x + e < W) dst[e] = o[e];
                }
            }
        }
        __syncthreads();
        // ---- Sobel (:122-123) on rows / columns 3..36, accumulated over channels in this thread's registers ----
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 256 * k;
            if (i >= (TF - 6) * TG) break;
            const int rr = i / TG, g = i - rr * TG, r = rr + 3;
            float a[3][6];                                            // rows r - 1 .. r + 1, columns 4 g - 1 .. 4 g + 4
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float* p = &sBl[(r - 1 + d) * TP + 4 * g + 4];
                const float4 m = lds4(p);
                a[d][0] = p[-1]; a[d][1] = m.x; a[d][2] = m.y; a[d][3] = m.z; a[d][4] = m.w; a[d][5] = p[4];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // cross-correlation with [[1,0,-1],[2,0,-2],[1,0,-1]] and its transpose
                gx[k][e] += (a[0][e] - a[0][e + 2]) + 2.f * (a[1][e] - a[1][e + 2]) + (a[2][e] - a[2][e + 2]);
                gy[k][e] += (a[0][e] - a[2][e]) + 2.f * (a[0][e + 1] - a[2][e + 1]) + (a[0][e + 2] - a[2][e + 2]);
            }
        }
    }
    // ---- gradient magnitude (:126-127) on the halo region, zero outside the image (padding of the directional filters) ----
    float* sMag = sIn;                                                // its last reader was the last horizontal pass
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 256 * k;
        if (i >= (TF - 6) * TG) break;
        const int rr = i / TG, g = i - rr * TG, r = rr + 3;
        const int y = y0 - 4 + r, x = x0 - 4 + 4 * g;
        float m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gx[k][e] = gx[k][e] / (float)C;
            gy[k][e] = gy[k][e] / (float)C;
            m[e] = (y >= 0 && y < H && x + e >= 0 && x + e < W) ? sqrtf(gx[k][e] * gx[k][e] + gy[k][e] * gy[k][e]) : 0.f;
        }
        *reinterpret_cast<float4*>(&sMag[r * TP + 4 * g + 4]) = make_float4(m[0], m[1], m[2], m[3]);
    }
    __syncthreads();
    // ---- orientation bins, threshold, non-max suppression: the interior items (rows 4..35, groups 1..8) ----
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 256 * k;
        if (i >= (TF - 6) * TG) break;
        const int rr = i / TG, g = i - rr * TG, r = rr + 3;
        const int y = y0 - 4 + r, x = x0 - 4 + 4 * g;
        if (r < 4 || r >= 4 + CT || g < 1 || g > 8 || y >= H || x >= W) continue;
        float nb[3][6];                                               // magnitudes, rows r - 1 .. r + 1, columns 4 g - 1 .. 4 g + 4
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float* p = &sMag[(r - 1 + d) * TP + 4 * g + 4];
            const float4 m = lds4(p);
            nb[d][0] = p[-1]; nb[d][1] = m.x; nb[d][2] = m.y; nb[d][3] = m.z; nb[d][4] = m.w; nb[d][5] = p[4];
        }
        float o_mag[4], o_ori[4], o_thr[4], o_thin[4], o_edge[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m = nb[1][e + 1];
            const int kbin = orientation_bin(gx[k][e], gy[k][e]);                               // :128-129
            const float mt = (m < threshold) ? 0.f : m;                                         // :132-133
            o_mag[e] = m; o_ori[e] = 45.0f * (float)kbin; o_thr[e] = mt; o_thin[e] = 0.f; o_edge[e] = mt;
            if (nms) {
                // directional differences centre - neighbour, order 0,45,...,315 degrees (:56-102); positive_idx = k mod 8 (:144),
                // and the pair (pos_i, pos_i + 4) it selects is k mod 4: E/W, SE/NW, S/N, SW/NE
                const int pos = kbin & 3;
                const float na = pos == 0 ? nb[1][e + 2] : pos == 1 ? nb[2][e + 2] : pos == 2 ? nb[2][e + 1] : nb[2][e];
                const float nc = pos == 0 ? nb[1][e] : pos == 1 ? nb[0][e] : pos == 2 ? nb[0][e + 1] : nb[0][e + 2];
                const bool is_max = fminf(m - na, m - nc) > 0.0f;                                // :154
                const float t = is_max ? m : 0.f;                                               // :158-159
                o_thin[e] = t;
                o_edge[e] = (t < threshold) ? 0.f : t;                                          // :160-161
            }
        }
        const size_t o = (size_t)b * plane + (size_t)y * W + x;
        auto put = [&](float* base, size_t off, const float (&v)[4]) {
            if (!base) return;
            if (vec && x + 3 < W) {
                *reinterpret_cast<float4*>(base + off) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (x + e < W) base[off + e] = v[e];
            }
        };
        put(grad_mag, o, o_mag);
        put(grad_ori, o, o_ori);
        put(thr_mag, o, o_thr);
        if (nms) { put(thin, o, o_thin); put(thr_thin, o, o_edge); }
        put(edge_out, (size_t)b * edge_batch_stride + (size_t)y * W + x, o_edge);
    }
}

// ---- The same detector, marching down the rows: the product form for 1 and 3 channels ---------------------------------------
// The tile kernel above spends its time on instruction issue (about 2 200 instructions per thread and tile: index arithmetic,
// border tests and one LDS access per value in every stage, 56 % halo overhead in the early stages, five barriers per channel:
// 0.090 ms for 64 crops, a quarter of the HBM roofline).  Here one WAVE owns a strip of rows of up to 256 columns -- four
// adjacent pixels per lane -- and walks down it; every stage's vertical neighbourhood lives in the lane's own registers (a ring
// of the last five horizontally filtered rows, the last three blurred rows, the last three magnitude rows), the horizontal
// neighbours come from the adjacent lanes by DPP wavefront shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1; the lane without a
// neighbour receives 0, which is exactly the zero padding of the reference's convolutions).  No LDS, no barriers, no index
// decoding; the arithmetic of every pixel is the tile kernel's, in the same order.
//   step with input row y:  h(y) -> blurred(y - 2) -> gradient, magnitude(y - 3) -> outputs(y - 4)
// Images wider than 256 are cut into column blocks of 248 valid columns with four columns of overlap on each side.
// (bound_ctrl: the lane without a source reads 0 -- no zeroed destination register to set up per shift)
__device__ __forceinline__ float dpp_from_left(float v) {      // lane i <- lane i - 1; lane 0 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {     // lane i <- lane i + 1; lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// grad / num_channels (:125), correctly rounded (div3_rn: hps_common.h).  One channel: nothing to do.
template <int C>
__device__ __forceinline__ float div_by_channels(float v) {
    if (C == 1) return v;
    if (C == 3) return div3_rn(v);
    return v / (float)C;
}

struct CannyOut {
    float* blurred; float* grad_mag; float* grad_ori; float* thr_mag; float* thin; float* thr_thin; float* edge;
    size_t edge_batch_stride;
};

template <int C, bool FULL, bool VEC, bool NMS>
__global__ __launch_bounds__(256) void canny_rows_kernel(const float* __restrict__ img, float g0, float g1, float g2, float g3,
                                                         float g4, CannyOut out, int H, int W, float threshold,
                                                         int rows_per_strip, int strips, int col_blocks, int n_items) {
    // the wave's number as a SCALAR (readfirstlane: hipcc cannot know that threadIdx.x >> 6 is the same in all 64 lanes): the strip,
    // its rows and every row test below then live in scalar registers, row addresses are scalar bases + one per-lane column offset
    const int item = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= n_items) return;                                   // whole waves only: every lane of a wave stays active below
    const int lane = threadIdx.x & 63;
    const int s = item % strips, cb = (item / strips) % col_blocks, b = item / (strips * col_blocks);
    const float gk[G] = {g0, g1, g2, g3, g4};
    const size_t plane = (size_t)H * W;
    // columns: this lane's four pixels x .. x + 3; outputs are stored for columns [out_lo, out_hi).  VEC: W % 4 == 0 -- rows are
    // 16-byte aligned and a lane's four pixels are all inside or all outside the image.
    const int x0 = col_blocks == 1 ? 0 : cb * 248 - 4;
    const int out_lo = col_blocks == 1 ? 0 : cb * 248, out_hi = col_blocks == 1 ? W : min(W, cb * 248 + 248);
    const int x = x0 + 4 * lane;
    bool cin[4], cst[4];                                           // column inside the image / column stored by this lane
#pragma unroll
    for (int e = 0; e < 4; ++e) { cin[e] = x + e >= 0 && x + e < W; cst[e] = x + e >= out_lo && x + e < out_hi; }
    const bool all_in = cin[0] && cin[3], all_st = cst[0] && cst[3];
    const int Y0 = s * rows_per_strip, Y1 = min(H, Y0 + rows_per_strip);
    const int steps = (Y1 - Y0) + 8;

    // Memory accesses are BUFFER instructions on a descriptor of ONE ROW (base = the row's address, a scalar; num_records = its
    // W * 4 bytes, or 0 when the row does not exist / must not be written): a lane whose byte offset is not below num_records reads
    // zeros and its stores are dropped -- by the address unit, not by the exec mask.  So every load and store is issued
    // unconditionally: no clamped addresses and zero-selects, and no branch around any of them.  With `if (lane stores)` regions
    // around global stores hipcc could not count the stores in flight (vmcnt counts loads AND stores on gfx9; a skipped region
    // issues none), assumed none, and waited s_waitcnt vmcnt(6) for the row loaded two steps earlier -- i.e. for all but the last
    // six memory operations, the previous step's nine stores included; now the waits are vmcnt(17..21).  (Measured: 240 fewer
    // VALU instructions per six steps, no AGPR spills, 0.0416 -> 0.0405 ms; the 30 % of the wave cycles that SQ_WAIT_ANY reports
    // -- tools/canny_pmc.sh -- did NOT move, so they are not waits for stores.)
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    constexpr int RSRC_FLAGS = 0x00020000;                         // raw buffer, 32-bit data format (gfx90a / gfx94x / gfx950)
    const unsigned row_bytes = 4u * (unsigned)W;
    // per-lane byte offsets, constant for the whole kernel.  Loads: 4 x as an unsigned number -- columns left of the image wrap to
    // 2^32 - 16, columns right of it are >= row_bytes: both out of range.  Stores: out of range unless the lane stores the column.
    const unsigned ld_off = 4u * (unsigned)x;
    unsigned st_off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) st_off[e] = cst[e] ? 4u * (unsigned)(x + e) : 0xffffffffu;
    const unsigned st_off4 = all_st ? 4u * (unsigned)x : 0xffffffffu;
    auto load_row = [&](int c, int y) __attribute__((always_inline)) -> float4 {
        const bool yin = y >= 0 && y < H;
        const float* row = img + ((size_t)b * C + c) * plane + (size_t)min(max(y, 0), H - 1) * W;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, yin ? row_bytes : 0u, RSRC_FLAGS);
        float4 v;
        if (VEC) {
            // (the whole vector is bit-cast: __builtin_bit_cast(float, q.y) on an ELEMENT of the loaded vector is miscompiled by this
            // hipcc -- it narrows the load to one dword and uses undefined values for y, z, w; tools/buffer_probe.hip)
            const v4f q = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, ld_off, 0, 0));
            v = make_float4(q.x, q.y, q.z, q.w);
        } else {
            v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, ld_off, 0, 0));
            v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, ld_off + 4u, 0, 0));
            v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, ld_off + 8u, 0, 0));
            v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, ld_off + 12u, 0, 0));
        }
        return v;
    };
    // base: never null where this is called; row_off and `wanted` are scalars
    auto store4 = [&](float* base, size_t row_off, bool wanted, const float (&v)[4]) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base + row_off, 0, wanted ? row_bytes : 0u, RSRC_FLAGS);
        if (VEC) {
            const v4f q = {v[0], v[1], v[2], v[3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, q), r, st_off4, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), r, st_off[e], 0, 0);
        }
    };

    // Every vertical window is a ring indexed by the step number modulo its length; the step loop is unrolled six times (a multiple
    // of every ring length, the five-row Gaussian ring carrying one spare slot), so that every ring index is a compile-time
    // constant and no window is ever shifted through register moves.  Every slot is written before its first use (the eight
    // warm-up steps below fill the rings in the order the stages consume them), so nothing is initialised.
    float hw[C][6][4];             // horizontally filtered rows: row y_in lands in slot t % 6
    float ar[C][3][6];             // blurred rows, columns x - 1 .. x + 4: row y_bl lands in slot t % 3
    float mr[3][6];                // magnitude rows, columns x - 1 .. x + 4: row y_g lands in slot t % 3
    float gr[2][2][4];             // gradient (x, y) of row y_g in slot t % 2
    float4 inr[C][3];              // input rows: row y_in sits in slot t % 3, row y_in + 2 is loaded into slot (t + 2) % 3
    // The first six rows are requested at once, before anything is computed: the warm-up steps are short (a horizontal Gaussian is
    // a quarter of a microsecond), a row requested two steps ahead arrives after its step began.  (The registers are free: the
    // other rings are not in use yet.  Worth less than expected: within the noise of the 0.040 ms.)
    float4 pre[C][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) pre[c][r] = load_row(c, Y0 - 4 + r);

    // STAGES: how much of the chain a step runs -- 1: horizontal Gaussian only, 2: + vertical Gaussian (a blurred row), 3: + Sobel
    // and magnitude, 4: + the outputs of a row.  The strip's first output row Y0 needs magnitudes from Y0 - 1, blurred rows from
    // Y0 - 2, horizontally filtered rows from Y0 - 4: the warm-up steps t = 0..3 run stage 1, t = 4, 5 stages 1-2, t = 6, 7 stages
    // 1-3 (2.5 instead of 5.5 steps' worth of instructions for the eight rows of halo every strip re-reads).
    auto step = [&](auto ph, auto stages, auto pre_row, int t) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph)::value;                    // t % 6
        constexpr int STAGES = decltype(stages)::value;
        constexpr int PRE = decltype(pre_row)::value;              // steps 0..5: their row is pre[.][PRE]; -1: the ring's
        constexpr int S3 = PH % 3, S2 = PH % 2;
        const int y_in = Y0 - 4 + t, y_bl = y_in - 2, y_g = y_in - 3, y_o = y_in - 4;
        const bool bl_in = y_bl >= 0 && y_bl < H, g_in = y_g >= 0 && y_g < H;
        float gx[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 cur = PRE >= 0 ? pre[c][PRE >= 0 ? PRE : 0] : inr[c][S3];
            if (PRE < 0 || PRE >= 4) inr[c][(S3 + 2) % 3] = load_row(c, y_in + 2);     // rows 6, 7, ...: two steps ahead
            // ---- horizontal Gaussian (:118) of row y_in: columns x - 2 .. x + 5 ----
            const float v[8] = {dpp_from_left(cur.z), dpp_from_left(cur.w), cur.x, cur.y, cur.z, cur.w,
                                dpp_from_right(cur.x), dpp_from_right(cur.y)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < G; ++k) acc += gk[k] * v[e + k];
                hw[c][PH][e] = acc;    // outside the image's columns this is not zero as the reference's is -- but the vertical pass
                                       // does not mix columns, and the blurred row is zeroed there below
            }
            if (STAGES < 2) continue;
            // ---- vertical Gaussian: blurred row y_bl from the rows y_in - 4 .. y_in of the ring ----
            float bl[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < G; ++k) acc += gk[k] * hw[c][(PH + 2 + k) % 6][e];
                bl[e] = (bl_in && cin[e]) ? acc : 0.f;             // zero outside the image: what the Sobel convolutions pad with
            }
            if (FULL && STAGES >= 3)       // :119 (the first blurred row of the strip is formed at step 6; num_records = 0 drops the others)
                store4(out.blurred, ((size_t)b * C + c) * plane + (size_t)min(max(y_bl, 0), H - 1) * W, bl_in && y_bl >= Y0 && y_bl < Y1, bl);
            // ---- Sobel (:122-123) of row y_g from the blurred rows y_g - 1 (a0), y_g (a1), y_g + 1 (a2, new) ----
            float (&a2)[6] = ar[c][S3];
            const float (&a1)[6] = ar[c][(S3 + 2) % 3];
            const float (&a0)[6] = ar[c][(S3 + 1) % 3];
            a2[0] = dpp_from_left(bl[3]); a2[1] = bl[0]; a2[2] = bl[1]; a2[3] = bl[2]; a2[4] = bl[3]; a2[5] = dpp_from_right(bl[0]);
            if (STAGES < 3) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gx[e] += (a0[e] - a0[e + 2]) + 2.f * (a1[e] - a1[e + 2]) + (a2[e] - a2[e + 2]);
                gy[e] += (a0[e] - a2[e]) + 2.f * (a0[e + 1] - a2[e + 1]) + (a0[e + 2] - a2[e + 2]);
            }
        }
        if (STAGES < 3) return;
        // ---- gradient magnitude (:126-127) of row y_g, zero outside the image ----
        float (&m2)[6] = mr[S3];
        const float (&m1)[6] = mr[(S3 + 2) % 3];
        const float (&m0)[6] = mr[(S3 + 1) % 3];
        float mg[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gx[e] = div_by_channels<C>(gx[e]);
            gy[e] = div_by_channels<C>(gy[e]);
            gr[S2][0][e] = gx[e];
            gr[S2][1][e] = gy[e];
            mg[e] = (g_in && cin[e]) ? sqrtf(gx[e] * gx[e] + gy[e] * gy[e]) : 0.f;
        }
        m2[0] = dpp_from_left(mg[3]); m2[1] = mg[0]; m2[2] = mg[1]; m2[3] = mg[2]; m2[4] = mg[3]; m2[5] = dpp_from_right(mg[0]);
        if (STAGES < 4) return;
        // ---- outputs of row y_o = y_g - 1: magnitudes m0 / m1 / m2 = rows y_o - 1, y_o, y_o + 1, gradient of the previous step ----
        {
            const float (&gxp)[4] = gr[(S2 + 1) % 2][0];
            const float (&gyp)[4] = gr[(S2 + 1) % 2][1];
            float o_mag[4], o_ori[4], o_thr[4], o_thin[4], o_edge[4];
            int kbin[4];
            bool near[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) kbin[e] = orientation_bin_fast(gxp[e], gyp[e], near[e]);       // :128-129
            if (near[0] | near[1] | near[2] | near[3]) {               // rare: a gradient on a sector boundary
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ks = orientation_bin_slow(gxp[e], gyp[e]);
                    kbin[e] = near[e] ? ks : kbin[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m = m1[e + 1];
                const float mt = (m < threshold) ? 0.f : m;                                     // :132-133
                o_mag[e] = m; o_ori[e] = 45.0f * (float)kbin[e]; o_thr[e] = mt; o_thin[e] = 0.f; o_edge[e] = mt;
                if (NMS) {
                    // the direction pair E/W, SE/NW, S/N or SW/NE (:56-102) that positive_idx mod 4 (:144) picks.  The eight
                    // neighbours are named values BEFORE the selects: written as pos == 0 ? m1[e + 2] : ... hipcc merged the
                    // array reads into one read at a selected ADDRESS -- the window went to scratch memory.
                    const int pos = kbin[e] & 3;
                    const float n_e = m1[e + 2], n_w = m1[e], n_se = m2[e + 2], n_nw = m0[e];
                    const float n_s = m2[e + 1], n_n = m0[e + 1], n_sw = m2[e], n_ne = m0[e + 2];
                    const bool p0 = pos == 0, p1 = pos == 1, p2 = pos == 2;
                    const float na2 = p2 ? n_s : n_sw, nc2 = p2 ? n_n : n_ne;
                    const float na1 = p1 ? n_se : na2, nc1 = p1 ? n_nw : nc2;
                    const float na = p0 ? n_e : na1, nc = p0 ? n_w : nc1;
                    const bool is_max = fminf(m - na, m - nc) > 0.0f;                            // :154
                    const float tt = is_max ? m : 0.f;                                          // :158-159
                    o_thin[e] = tt;
                    o_edge[e] = (tt < threshold) ? 0.f : tt;                                    // :160-161
                }
            }
            const size_t o = (size_t)b * plane + (size_t)y_o * W;
            if (FULL) {                    // hps_canny_edges: every output of the reference's dict, no edge map
                store4(out.grad_mag, o, true, o_mag);
                store4(out.grad_ori, o, true, o_ori);
                store4(out.thr_mag, o, true, o_thr);
                if (NMS) { store4(out.thin, o, true, o_thin); store4(out.thr_thin, o, true, o_edge); }
            } else {                       // hps_canny_edge_map: the edge map alone
                store4(out.edge, (size_t)b * out.edge_batch_stride + (size_t)y_o * W, true, o_edge);
            }
        }
    };
    using std::integral_constant;
    using none = integral_constant<int, -1>;
    step(integral_constant<int, 0>(), integral_constant<int, 1>(), integral_constant<int, 0>(), 0);
    step(integral_constant<int, 1>(), integral_constant<int, 1>(), integral_constant<int, 1>(), 1);
    step(integral_constant<int, 2>(), integral_constant<int, 1>(), integral_constant<int, 2>(), 2);
    step(integral_constant<int, 3>(), integral_constant<int, 1>(), integral_constant<int, 3>(), 3);
    step(integral_constant<int, 4>(), integral_constant<int, 2>(), integral_constant<int, 4>(), 4);
    step(integral_constant<int, 5>(), integral_constant<int, 2>(), integral_constant<int, 5>(), 5);
    step(integral_constant<int, 0>(), integral_constant<int, 3>(), none(), 6);
    step(integral_constant<int, 1>(), integral_constant<int, 3>(), none(), 7);
    // The rows of the strip: step t = 8 + i writes row Y0 + i.  The only branches are the exits (nothing joins the straight line
    // again, so the waits for the rows loaded two steps earlier still count the memory operations issued since: behind a branch
    // AROUND every step hipcc fell back to s_waitcnt vmcnt(0) -- also draining the loads just issued, i.e. no prefetch at all).
    for (int t = 8;; t += 6) {
        step(integral_constant<int, 2>(), integral_constant<int, 4>(), none(), t);
        if (t + 1 >= steps) return;
        step(integral_constant<int, 3>(), integral_constant<int, 4>(), none(), t + 1);
        if (t + 2 >= steps) return;
        step(integral_constant<int, 4>(), integral_constant<int, 4>(), none(), t + 2);
        if (t + 3 >= steps) return;
        step(integral_constant<int, 5>(), integral_constant<int, 4>(), none(), t + 3);
        if (t + 4 >= steps) return;
        step(integral_constant<int, 0>(), integral_constant<int, 4>(), none(), t + 4);
        if (t + 5 >= steps) return;
        step(integral_constant<int, 1>(), integral_constant<int, 4>(), none(), t + 5);
        if (t + 6 >= steps) return;
    }
}

// proxy representation (predict/...:93-100): channel 0 = edge map, channels 1..K = visibility * Gaussian blob
// (label_conversions.py:123: exp(-((row - v) / std)^2 / 2 - ((col - u) / std)^2 / 2)).  A workgroup writes PR_ROWS rows of up to 256
// columns: the row term of a (row, joint) pair is computed once per workgroup (LDS), the column term once per thread and joint --
// the same operations on the same operands as evaluating the formula per pixel (bit-identical), but 2 instead of 34 IEEE
// divisions per pixel: the kernel was bound by them (0.100 ms for 64 x 18 x 256 x 256 = 0.40 of the HBM roofline).
constexpr int PR_ROWS = 8, PR_KMAX = 32;
__global__ __launch_bounds__(256) void proxy_rep_kernel(const float* __restrict__ edge, const float* __restrict__ joints2d,
                                                        const float* __restrict__ visib, float* __restrict__ out, int K,
                                                        int H, int W, float std) {
    __shared__ float s_row[PR_ROWS][PR_KMAX];            // ((y - v) / std)^2 / 2
    __shared__ float s_vis[PR_KMAX];
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y0 = blockIdx.y * PR_ROWS, b = blockIdx.z;
    for (int i = threadIdx.x; i < PR_ROWS * K; i += 256) {
        const int r = i / K, k = i - r * K;
        const float v = joints2d[((size_t)b * K + k) * 2 + 1];
        const float a = ((float)(y0 + r) - v) / std;
        s_row[r][k] = (a * a) / 2.0f;
    }
    if (threadIdx.x < K) s_vis[threadIdx.x] = visib ? visib[(size_t)b * K + threadIdx.x] : 1.0f;
    __syncthreads();
    if (x >= W) return;
    const size_t plane = (size_t)H * W;
    const int rows = min(PR_ROWS, H - y0);
    float* o = out + (size_t)b * (K + 1) * plane + (size_t)y0 * W + x;
    const float* e = edge + (size_t)b * plane + (size_t)y0 * W + x;
    if (edge)
        for (int r = 0; r < rows; ++r) o[(size_t)r * W] = e[(size_t)r * W];
    for (int k = 0; k < K; ++k) {
        const float u = joints2d[((size_t)b * K + k) * 2 + 0];
        const float c = ((float)x - u) / std;
        const float c2 = (c * c) / 2.0f;
        const float vis = s_vis[k];
        float* ok = o + (size_t)(k + 1) * plane;
        for (int r = 0; r < rows; ++r) {
            const float h = expf(-s_row[r][k] - c2);
            ok[(size_t)r * W] = visib ? h * vis : h;
        }
    }
}

// utils/label_conversions.py:127-155: arg-max of every (b,k) heat-map -> (x, y) pixel coordinates and a visibility flag
// (max > eps); invisible joints get (-1,-1).  One workgroup per heat-map; ties resolve to the first (lowest) index
// like torch.max.
__global__ __launch_bounds__(256) void heatmap_argmax_kernel(const float* __restrict__ heat, float* __restrict__ joints2d,
                                                             float* __restrict__ visib, int HW, int W, float eps) {
    const float* h = heat + (size_t)blockIdx.x * HW;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    auto take = [&](float v, int idx) {                       // the larger value; among equal values the lower index (torch.max's first maximum)
        if (v > best || (v == best && idx < bi)) { best = v; bi = idx; }
    };
    // 16-byte loads, four per lane in flight (a 64 x 64 map is four of them per lane), then the tail one float at a time
    const int n4 = ((HW & 3) == 0 && ((size_t)h & 15) == 0) ? HW >> 2 : 0;
    const float4* h4 = reinterpret_cast<const float4*>(h);
    int i = threadIdx.x;
    for (; i + 3 * 256 < n4; i += 4 * 256) {
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = h4[i + q * 256];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * (i + q * 256);
            take(v[q].x, e); take(v[q].y, e + 1); take(v[q].z, e + 2); take(v[q].w, e + 3);
        }
    }
    for (; i < n4; i += 256) {
        const float4 v = h4[i];
        take(v.x, 4 * i); take(v.y, 4 * i + 1); take(v.z, 4 * i + 2); take(v.w, 4 * i + 3);
    }
    for (int e = 4 * n4 + threadIdx.x; e < HW; e += 256) take(h[e], e);
    // wavefront first (shuffles), then the four wavefronts through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(best, off);
        const int oi = __shfl_down(bi, off);
        take(ov, oi);
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) take(sv[w], si[w]);
        sv[0] = best; si[0] = bi;
    }
    if (threadIdx.x == 0) {
        const bool vis = sv[0] > eps;
        joints2d[blockIdx.x * 2 + 0] = vis ? (float)(si[0] % W) : -1.0f;
        joints2d[blockIdx.x * 2 + 1] = vis ? floorf((float)si[0] / (float)W) : -1.0f;
        visib[blockIdx.x] = vis ? 1.0f : 0.0f;
    }
}

// utils/sampling_utils.py:210-229: per sample, the largest image-plane distance between its projected COCO joints
// (flipped 180 degrees about x, weak-perspective projection utils/cam_utils.py:9-16, de-normalised
// utils/joints2d_utils.py:5-10) and the visible input joints.
__global__ void sample_j2d_error_kernel(const float* __restrict__ joints, const int32_t* __restrict__ coco_map, int n_joints_all,
                                        const float* __restrict__ in_j2d, const float* __restrict__ in_vis,
                                        const float* __restrict__ cam, float img_wh, float* __restrict__ err, int N, int K) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const float sc = cam[0], tx = cam[1], ty = cam[2];
    float worst = -INFINITY;
    for (int k = 0; k < K; ++k) {
        if (in_vis[k] == 0.0f) continue;
        const float* j = joints + ((size_t)s * n_joints_all + coco_map[k]) * 3;
        const float u = (sc * (j[0] + tx) + 1.0f) * (img_wh / 2.0f);
        const float v = (sc * (-j[1] + ty) + 1.0f) * (img_wh / 2.0f);
        const float du = u - in_j2d[k * 2], dv = v - in_j2d[k * 2 + 1];
        worst = fmaxf(worst, sqrtf(du * du + dv * dv));
    }
    err[s] = worst;
}

}  // namespace hps

using namespace hps;

extern "C" int hps_heatmaps_to_joints2d(const float* heatmaps, float* joints2d, float* visib, int BK, int H, int W, float eps,
                                        hps_stream_t stream) {
    if (!heatmaps || !joints2d || !visib) return bad_arg("hps_heatmaps_to_joints2d: null pointer");
    if (BK <= 0) return HPS_OK;
    hipLaunchKernelGGL(heatmap_argmax_kernel, dim3(BK), dim3(256), 0, (hipStream_t)stream, heatmaps, joints2d, visib, H * W, W, eps);
    return check_launch("hps_heatmaps_to_joints2d");
}

extern "C" int hps_sample_joints2d_error(const float* joints, const int32_t* coco_map, int n_joints_all, const float* in_j2d,
                                         const float* in_vis, const float* cam, float img_wh, float* err, int N, int K,
                                         hps_stream_t stream) {
    if (!joints || !coco_map || !in_j2d || !in_vis || !cam || !err) return bad_arg("hps_sample_joints2d_error: null pointer");
    if (N <= 0) return HPS_OK;
    hipLaunchKernelGGL(sample_j2d_error_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, joints, coco_map,
                       n_joints_all, in_j2d, in_vis, cam, img_wh, err, N, K);
    return check_launch("hps_sample_joints2d_error");
}

static int launch_canny(const char* who, const float* img, const float* gauss_taps_host, int gauss_size, float* blurred, float* grad_mag,
                        float* grad_ori, float* thr_mag, float* thin, float* thr_thin, float* edge_out, size_t edge_batch_stride,
                        int B, int C, int H, int W, float threshold, int nms, hps_stream_t stream) {
    if (gauss_size != G) { set_error("%s: gaussian size %d unsupported (5)", who, gauss_size); return HPS_E_UNSUPPORTED; }
    if (B <= 0) return HPS_OK;
    if (C == 1 || C == 3) {
        // one wave per (image, column block, strip of rows); strips as tall as they can be while the chip still gets about one wave
        // per SIMD (each strip re-reads eight rows of halo).  Results do not depend on the strip height.
        const int col_blocks = W <= 256 ? 1 : ceil_div(W, 248);
        int rows = 64;
        while (rows > 8 && (long)B * col_blocks * ceil_div(H, rows) < 1024) rows /= 2;
        const int strips = ceil_div(H, rows);
        const long n_items = (long)B * col_blocks * strips;
        if (n_items > 0x7fffffffL / 4) return bad_arg("hps_canny_edges: too many strips");
        const bool full = blurred || grad_mag || grad_ori || thr_mag || thin || thr_thin;
        CannyOut o{blurred, grad_mag, grad_ori, thr_mag, thin, thr_thin, edge_out, edge_batch_stride};
        const dim3 grid((unsigned)ceil_div((int)n_items, 4)), block(256);
        const float* t = gauss_taps_host;
        hipStream_t st = (hipStream_t)stream;
#define HPS_CANNY_ROWS(CC, FF, VV, NN) hipLaunchKernelGGL((canny_rows_kernel<CC, FF, VV, NN>), grid, block, 0, st, img, t[0], t[1], t[2], t[3], t[4], o, H, W, threshold, rows, strips, col_blocks, (int)n_items)
#define HPS_CANNY_ROWS_N(CC, FF, VV) do { if (nms) HPS_CANNY_ROWS(CC, FF, VV, true); else HPS_CANNY_ROWS(CC, FF, VV, false); } while (0)
#define HPS_CANNY_ROWS_F(CC, VV) do { if (full) HPS_CANNY_ROWS_N(CC, true, VV); else HPS_CANNY_ROWS_N(CC, false, VV); } while (0)
        const bool vec = (W & 3) == 0;
        if (C == 3) { if (vec) HPS_CANNY_ROWS_F(3, true); else HPS_CANNY_ROWS_F(3, false); }
        else { if (vec) HPS_CANNY_ROWS_F(1, true); else HPS_CANNY_ROWS_F(1, false); }
#undef HPS_CANNY_ROWS_F
#undef HPS_CANNY_ROWS_N
#undef HPS_CANNY_ROWS
        return check_launch(who);
    }
    dim3 grid(ceil_div(W, CT), ceil_div(H, CT), B);
    hipLaunchKernelGGL(canny_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, gauss_taps_host[0], gauss_taps_host[1],
                       gauss_taps_host[2], gauss_taps_host[3], gauss_taps_host[4], blurred, grad_mag, grad_ori, thr_mag,
                       thin, thr_thin, edge_out, edge_batch_stride, C, H, W, threshold, nms);
    return check_launch(who);
}

extern "C" int hps_canny_edges(const float* img, const float* gauss_taps_host, int gauss_size, float* blurred,
                               float* grad_mag, float* grad_ori, float* thr_mag, float* thin, float* thr_thin, int B,
                               int C, int H, int W, float threshold, int nms, hps_stream_t stream) {
    if (!img || !gauss_taps_host || !blurred || !grad_mag || !grad_ori || !thr_mag) return bad_arg("hps_canny_edges: null pointer");
    if (nms && (!thin || !thr_thin)) return bad_arg("hps_canny_edges: thin / thr_thin needed with nms");
    return launch_canny("hps_canny_edges", img, gauss_taps_host, gauss_size, blurred, grad_mag, grad_ori, thr_mag, thin, thr_thin,
                        nullptr, 0, B, C, H, W, threshold, nms, stream);
}

extern "C" int hps_canny_edge_map(const float* img, const float* gauss_taps_host, int gauss_size, float* edge_out,
                                  int64_t edge_batch_stride, int B, int C, int H, int W, float threshold, int nms,
                                  hps_stream_t stream) {
    if (!img || !gauss_taps_host || !edge_out) return bad_arg("hps_canny_edge_map: null pointer");
    if (edge_batch_stride < (int64_t)H * W) return bad_arg("hps_canny_edge_map: edge_batch_stride smaller than one plane");
    return launch_canny("hps_canny_edge_map", img, gauss_taps_host, gauss_size, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                        edge_out, (size_t)edge_batch_stride, B, C, H, W, threshold, nms, stream);
}

extern "C" int hps_proxy_rep(const float* edge, const float* joints2d, const float* visib, float* out, int B, int K, int H,
                             int W, float std, hps_stream_t stream) {
    if (!joints2d || !out) return bad_arg("hps_proxy_rep: null pointer");
    if (K < 0 || K > PR_KMAX) return bad_arg("hps_proxy_rep: at most 32 joints");
    if (B <= 0 || H <= 0 || W <= 0) return HPS_OK;
    hipLaunchKernelGGL(proxy_rep_kernel, dim3(ceil_div(W, 256), ceil_div(H, PR_ROWS), B), dim3(256), 0, (hipStream_t)stream, edge,
                       joints2d, visib, out, K, H, W, std);
    return check_launch("hps_proxy_rep");
}

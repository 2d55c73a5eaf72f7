// Proxy-representation front end: Canny edge detection
// (models/canny_edge_detector.py:104-166) and 2D-joint Gaussian heat-maps
// (utils/label_conversions.py:105-124), i.e. what turns an RGB crop + 17 keypoints into the 18-channel network
// input (predict/predict_poseMF_shapeGaussian_net.py:88-100).  Pure stencil / element-wise work: HBM bound.
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
__device__ __forceinline__ int orientation_bin(float gx, float gy) {
    const float ax = fabsf(gx), ay = fabsf(gy);
    const float T1 = 0.41421356237309503f, T2 = 2.4142135623730951f;      // tan(22.5 deg), tan(67.5 deg)
    const float b1 = T1 * ax, b2 = T2 * ax;
    if (__builtin_expect(fabsf(ay - b1) <= 1e-4f * b1 || fabsf(ay - b2) <= 1e-4f * b2, 0)) {
        if (!(ax == 0.0f && ay == 0.0f)) {
            const float ori = atan2f(gy, gx) * (180.0f / 3.14159265358979323846f) + 180.0f;
            return (int)rintf(ori / 45.0f);
        }
    }
    const bool sx = __builtin_signbit(gx), sy = __builtin_signbit(gy);
    if (ay <= b1) return sx ? (sy ? 0 : 8) : 4;
    if (ay >= b2) return sy ? 2 : 6;
    return sy ? (sx ? 1 : 3) : (sx ? 7 : 5);
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
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float v = h[i];
        if (v > best) { best = v; bi = i; }                  // strided scan: each lane keeps its first maximum
    }
    __shared__ float sv[256];
    __shared__ int si[256];
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const float ov = sv[threadIdx.x + off];
            const int oi = si[threadIdx.x + off];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
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

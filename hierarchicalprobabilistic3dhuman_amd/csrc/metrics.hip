// Evaluation metrics on the device: per-point L2 errors between predicted and target
// point sets (6890 vertices or 14 joints), raw, after scale-and-translation correction
// (utils/eval_utils.py:70-89) or after Procrustes alignment (utils/eval_utils.py:11-59), as used by
// metrics/eval_metrics_tracker.py:89-269.  Streaming reductions: HBM bound.
//
// A "set" is one (P,3) point cloud.  pred holds S sets; set s is compared with target set s / group
// (group = 1: one target per prediction; group = N: N samples of one frame share its target).
#include "hps_common.h"

namespace hps {

constexpr int NSTAT = 17;  // sum p (3), sum t (3), sum |p|^2, sum |t|^2, sum p_a t_b (9)

__global__ __launch_bounds__(256) void pointset_stats_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                             int group, int P, double* __restrict__ stats) {
    const int s = blockIdx.x;
    const float* p = pred + (size_t)s * P * 3;
    const float* t = target + (size_t)(s / group) * P * 3;
    double acc[NSTAT];
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) acc[i] = 0.0;
    // four of the lane's points (and their targets) requested before the first is used -- 12-byte records, one dwordx3 load each;
    // the sums keep their order (same bits).  As a plain loop every point was a dependent round trip: 27 per lane for 6 890 points.
    auto add = [&](const f3 pp, const f3 tt) {
        const double px = pp.x, py = pp.y, pz = pp.z;
        const double tx = tt.x, ty = tt.y, tz = tt.z;
        acc[0] += px; acc[1] += py; acc[2] += pz;
        acc[3] += tx; acc[4] += ty; acc[5] += tz;
        acc[6] += px * px + py * py + pz * pz;
        acc[7] += tx * tx + ty * ty + tz * tz;
        acc[8] += px * tx; acc[9] += px * ty; acc[10] += px * tz;
        acc[11] += py * tx; acc[12] += py * ty; acc[13] += py * tz;
        acc[14] += pz * tx; acc[15] += pz * ty; acc[16] += pz * tz;
    };
    const f3* p3 = reinterpret_cast<const f3*>(p);
    const f3* t3 = reinterpret_cast<const f3*>(t);
    int i = threadIdx.x;
    for (; i + 3 * 256 < P; i += 4 * 256) {
        f3 pp[4], tt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { pp[q] = p3[i + q * 256]; tt[q] = t3[i + q * 256]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) add(pp[q], tt[q]);
    }
    for (; i < P; i += 256) add(p3[i], t3[i]);
    __shared__ double red[4][NSTAT];
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) {
        double v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NSTAT)
        stats[(size_t)s * NSTAT + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// symmetric 3x3 eigen-decomposition by cyclic Jacobi (double): A = V diag(w) V^T, columns of V
__device__ void jacobi_eig3(double A[3][3], double V[3][3], double w[3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 24; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
                for (int k = 0; k < 3; ++k) {           // A <- A J
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {           // A <- J^T A
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

// per set: similarity transform q = M p + t as 12 floats (row-major 3x4).  mode 0 identity, 1 scale+translation,
// 2 Procrustes (rotation from the SVD of K = X1 X2^T with the det fix, scale = tr(R K) / var1, t = mu2 - s R mu1).
__global__ void pointset_transform_kernel(const double* __restrict__ stats, int S, int P, int mode, float* __restrict__ xf) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double* st = stats + (size_t)s * NSTAT;
    double M[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, tr[3] = {0, 0, 0};
    if (mode != 0) {
        const double n = (double)P;
        const double mp[3] = {st[0] / n, st[1] / n, st[2] / n}, mt[3] = {st[3] / n, st[4] / n, st[5] / n};
        const double var_p = st[6] - n * (mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2]);
        const double var_t = st[7] - n * (mt[0] * mt[0] + mt[1] * mt[1] + mt[2] * mt[2]);
        if (mode == 1) {
            const double sc = sqrt(var_t / n) / sqrt(var_p / n);                 // eval_utils.py:80-87
            for (int i = 0; i < 3; ++i) { M[i][i] = sc; tr[i] = mt[i] - sc * mp[i]; }
        } else {
            double K[3][3];                                                     // K = X1 X2^T  (:33)
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) K[a][b] = st[8 + a * 3 + b] - n * mp[a] * mt[b];
            // SVD K = U S V^T from the eigen-decomposition of K^T K; R = V Z U^T (:37-44)
            double A[3][3], V[3][3], w[3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) A[a][b] = K[0][a] * K[0][b] + K[1][a] * K[1][b] + K[2][a] * K[2][b];
            jacobi_eig3(A, V, w);
            int o[3] = {0, 1, 2};                                               // sort eigenvalues descending
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2 - i; ++j)
                    if (w[o[j]] < w[o[j + 1]]) { const int tmp = o[j]; o[j] = o[j + 1]; o[j + 1] = tmp; }
            double v0[3], v1[3], v2[3], u0[3], u1[3], u2[3];
            for (int k = 0; k < 3; ++k) { v0[k] = V[k][o[0]]; v1[k] = V[k][o[1]]; }
            v2[0] = v0[1] * v1[2] - v0[2] * v1[1]; v2[1] = v0[2] * v1[0] - v0[0] * v1[2]; v2[2] = v0[0] * v1[1] - v0[1] * v1[0];
            for (int k = 0; k < 3; ++k) u0[k] = K[k][0] * v0[0] + K[k][1] * v0[1] + K[k][2] * v0[2];
            double nrm = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
            for (int k = 0; k < 3; ++k) u0[k] /= (nrm > 0 ? nrm : 1.0);
            for (int k = 0; k < 3; ++k) u1[k] = K[k][0] * v1[0] + K[k][1] * v1[1] + K[k][2] * v1[2];
            const double d01 = u0[0] * u1[0] + u0[1] * u1[1] + u0[2] * u1[2];
            for (int k = 0; k < 3; ++k) u1[k] -= d01 * u0[k];
            nrm = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
            for (int k = 0; k < 3; ++k) u1[k] /= (nrm > 0 ? nrm : 1.0);
            u2[0] = u0[1] * u1[2] - u0[2] * u1[1]; u2[1] = u0[2] * u1[0] - u0[0] * u1[2]; u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
            // with det U = det V = +1 the third "singular value" u2^T K v2 carries the sign of det K, and the
            // reference's Z = diag(1,1,sign(det(U V^T))) on a positive-singular-value SVD gives the same R: V U^T
            double R[3][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) R[a][b] = v0[a] * u0[b] + v1[a] * u1[b] + v2[a] * u2[b];
            double trRK = 0.0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) trRK += R[a][b] * K[b][a];
            const double sc = trRK / var_p;                                     // :47
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) M[a][b] = sc * R[a][b];
                tr[a] = mt[a] - (M[a][0] * mp[0] + M[a][1] * mp[1] + M[a][2] * mp[2]);   // :50
            }
        }
    }
    float* o = xf + (size_t)s * 12;
    for (int a = 0; a < 3; ++a) { o[a * 4] = (float)M[a][0]; o[a * 4 + 1] = (float)M[a][1]; o[a * 4 + 2] = (float)M[a][2]; o[a * 4 + 3] = (float)tr[a]; }
}

// per set: sum over points of || M p + t - target ||; optional transformed points out
__global__ __launch_bounds__(256) void pointset_error_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                             const float* __restrict__ xf, int group, int P,
                                                             double* __restrict__ err_sum, float* __restrict__ transformed) {
    const int s = blockIdx.x;
    const float* p = pred + (size_t)s * P * 3;
    const float* t = target + (size_t)(s / group) * P * 3;
    float M[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) M[i] = xf[(size_t)s * 12 + i];
    double acc = 0.0;
    const f3* p3 = reinterpret_cast<const f3*>(p);
    const f3* t3 = reinterpret_cast<const f3*>(t);
    auto one = [&](int i, const f3 pp, const f3 tt) {
        const float px = pp.x, py = pp.y, pz = pp.z;
        const float qx = M[0] * px + M[1] * py + M[2] * pz + M[3];
        const float qy = M[4] * px + M[5] * py + M[6] * pz + M[7];
        const float qz = M[8] * px + M[9] * py + M[10] * pz + M[11];
        if (transformed) {
            float* o = transformed + ((size_t)s * P + i) * 3;
            o[0] = qx; o[1] = qy; o[2] = qz;
        }
        const float dx = qx - tt.x, dy = qy - tt.y, dz = qz - tt.z;
        acc += (double)sqrtf(dx * dx + dy * dy + dz * dz);
    };
    // four points in flight per lane (see pointset_stats_kernel); same order of additions
    int i = threadIdx.x;
    for (; i + 3 * 256 < P; i += 4 * 256) {
        f3 pp[4], tt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { pp[q] = p3[i + q * 256]; tt[q] = t3[i + q * 256]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) one(i + q * 256, pp[q], tt[q]);
    }
    for (; i < P; i += 256) one(i, p3[i], t3[i]);
    __shared__ double red[4];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) err_sum[s] = red[0] + red[1] + red[2] + red[3];
}

// ---- float64 sums of float tensors (result checksums): two launches for up to four tensors, fixed summation order ----
constexpr int SUM_BLOCKS = 128;
struct SumJobs {
    const float* x[4];
    long n[4];
    int take_abs[4];
};

// block (b, j): elements b * 256 + t, + SUM_BLOCKS * 256, ... of tensor j, each lane in index order, then a fixed tree
__global__ __launch_bounds__(256) void sums_partial_kernel(const SumJobs jobs, double* __restrict__ partial) {
    const int j = blockIdx.y;
    const float* x = jobs.x[j];
    const long n = jobs.n[j];
    const bool a = jobs.take_abs[j] != 0;
    double acc = 0.0;
    // eight of the lane's elements requested before the first is added (the additions stay in index order: the same bits).  As a
    // plain loop every element was a dependent L2 round trip: 53 of them per lane for bench.py's largest tensor, 23-64 us per step
    // inside the timed loop for 14 MB of reads.
    constexpr long STRIDE = (long)SUM_BLOCKS * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * STRIDE < n; i += 8 * STRIDE) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[i + q * STRIDE];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += (double)(a ? fabsf(v[q]) : v[q]);
    }
    for (; i < n; i += STRIDE) {
        const float v = x[i];
        acc += (double)(a ? fabsf(v) : v);
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[j * SUM_BLOCKS + blockIdx.x] = red[0];
}

__global__ void sums_final_kernel(const double* __restrict__ partial, int count, double first, double* __restrict__ out,
                                  double* __restrict__ accumulate) {
    const int j = threadIdx.x;
    if (j == 0) {
        out[0] = first;
        if (accumulate) accumulate[0] += first;
    }
    if (j < count) {
        double t = 0.0;
        for (int b = 0; b < SUM_BLOCKS; ++b) t += partial[j * SUM_BLOCKS + b];
        out[1 + j] = t;
        if (accumulate) accumulate[1 + j] += t;       // one lane per entry: running total = total + this call's sum, as a caller's add would form it
    }
}

}  // namespace hps

using namespace hps;

extern "C" int hps_sums_f64(const float* const* xs, const int64_t* ns, const int32_t* take_abs, int count, double first,
                            double* partial_ws, double* out, double* accumulate, hps_stream_t stream) {
    if (!xs || !ns || !take_abs || !partial_ws || !out) return bad_arg("hps_sums_f64: null pointer");
    if (count < 1 || count > 4) return bad_arg("hps_sums_f64: 1..4 tensors");
    SumJobs jobs;
    for (int j = 0; j < 4; ++j) {
        jobs.x[j] = j < count ? xs[j] : nullptr;
        jobs.n[j] = j < count ? (long)ns[j] : 0;
        jobs.take_abs[j] = j < count ? take_abs[j] : 0;
        if (j < count && (!xs[j] || ns[j] < 0)) return bad_arg("hps_sums_f64: tensor");
    }
    hipLaunchKernelGGL(sums_partial_kernel, dim3(SUM_BLOCKS, count), dim3(256), 0, (hipStream_t)stream, jobs, partial_ws);
    hipLaunchKernelGGL(sums_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_ws, count, first, out, accumulate);
    return check_launch("hps_sums_f64");
}

extern "C" int hps_pointset_errors(const float* pred, const float* target, int S, int group, int P, int mode,
                                   double* stats_ws, float* xf_ws, double* err_sum, float* transformed,
                                   hps_stream_t stream) {
    if (!pred || !target || !stats_ws || !xf_ws || !err_sum) return bad_arg("hps_pointset_errors: null pointer");
    if (group < 1 || P < 1 || mode < 0 || mode > 2) return bad_arg("hps_pointset_errors: group / P / mode");
    if (S <= 0) return HPS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (mode != 0) hipLaunchKernelGGL(pointset_stats_kernel, dim3(S), dim3(256), 0, s, pred, target, group, P, stats_ws);
    hipLaunchKernelGGL(pointset_transform_kernel, dim3(ceil_div(S, 64)), dim3(64), 0, s, stats_ws, S, P, mode, xf_ws);
    hipLaunchKernelGGL(pointset_error_kernel, dim3(S), dim3(256), 0, s, pred, target, xf_ws, group, P, err_sum, transformed);
    return check_launch("hps_pointset_errors");
}

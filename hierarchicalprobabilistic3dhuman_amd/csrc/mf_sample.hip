// Matrix-Fisher rotation sampling by Bingham / angular-central-Gaussian rejection sampling, one
// workgroup of 1-8 wavefronts per (image, joint) call.  Replaces utils/sampling_utils.py:10-143:
// the per-call Python loop (:128-137), the boolean-mask compaction (:64-65) and the host
// synchronisation per round (:62) become ballot + prefix-popcount inside one wave.
#include "hps_common.h"

namespace hps {

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based: no state to carry between launches -------
struct u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 ctr, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr.z;
        u4 n;
        n.x = (uint32_t)(p1 >> 32) ^ ctr.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ ctr.w ^ k1;
        n.w = (uint32_t)p0;
        ctr = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return ctr;
}

// (0,1] uniform from 32 bits (never 0, so log() is finite); [0,1) uniform from the top 24 bits
__device__ __forceinline__ float u01_open0(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float r = sqrtf(-2.0f * logf(u01_open0(a)));
    const float t = 6.28318530717958647692f * u01(b);
    n0 = r * cosf(t);
    n1 = r * sinf(t);
}

// Workgroup = one (image, joint) call = W wavefronts (W = blockDim.x / 64, chosen by the launcher from num_samples only).
// Proposals are taken in order in super-blocks of 64 W: wave w evaluates proposals [64 (it W + w), +64) of iteration it, the
// waves exchange their accept counts through LDS (one barrier per iteration, two alternating slots), and every accepted proposal
// gets its rank in PROPOSAL ORDER (accepts of earlier iterations + of lower waves + of lower lanes: ballot and popcount) -- the
// ordered compaction of the reference's boolean mask (:64-65), identical for every W.
// The round is decided as soon as N proposals have been accepted (the reference evaluates all 8N and then keeps the first N
// accepted, :61-66 -- the same N proposals), so with the usual acceptance of 0.4-0.7 about 2N of the 8N proposals are ever
// evaluated and nothing is evaluated twice.  A round that ends below N accepts (all n_prop evaluated) is discarded like the
// reference's (:68-69): whatever it wrote is overwritten by the round that succeeds.  count_all (the Bingham entry point, whose
// accept_ratio :67 needs the total): keeps counting after the N-th accept without emitting.
__global__ __launch_bounds__(512) void mf_sample_kernel(
    const float* __restrict__ pose_u, const float* __restrict__ pose_s, const float* __restrict__ pose_v,
    const float* __restrict__ bingham_a, const float* __restrict__ acg_override, int nj, int N, int n_prop, float b, float m_star, const float* __restrict__ eps, const float* __restrict__ wun,
    const int32_t* __restrict__ draw_idx, uint64_t seed, int64_t call_offset, const uint64_t* __restrict__ seed_dev, int max_rounds,
    int count_all, float* __restrict__ r_out, float* __restrict__ quat_out, int32_t* __restrict__ accepted) {
    __shared__ int sCnt[2][8];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int img = c / nj, joint = c % nj;

    // ---- per-call parameters (uniform over the workgroup; every lane computes them redundantly) ----
    float U[9], V[9], S[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) { U[e] = pose_u[(size_t)c * 9 + e]; V[e] = pose_v[(size_t)c * 9 + e]; }
#pragma unroll
    for (int e = 0; e < 3; ++e) S[e] = pose_s[(size_t)c * 3 + e];
    const float detU = det3(U), detV = det3(V);                 // sampling_utils.py:105
    S[2] *= detU * detV;                                        // :109
    U[2] *= detU; U[5] *= detU; U[8] *= detU;                   // :110  (third column)
    V[2] *= detV; V[5] *= detV; V[8] *= detV;                   // :111
    float A[4], Om[4], sd[4];
    A[0] = 0.0f;
    A[1] = 2.0f * (S[1] + S[2]);                                // :119-121
    A[2] = 2.0f * (S[0] + S[2]);
    A[3] = 2.0f * (S[0] + S[1]);
    if (bingham_a) {                                            // bingham_sampling_for_matrix_fisher_torch(A=...)
#pragma unroll
        for (int e = 0; e < 4; ++e) A[e] = bingham_a[(size_t)c * 4 + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        Om[e] = 1.0f + 2.0f * A[e] / b;                         // :123
        sd[e] = 1.0f / sqrtf(Om[e]);                            // :124  Omega ** -0.5
    }
    if (acg_override) {                                         // caller-supplied Omega / Gaussian_std (:42-45 are only defaults)
#pragma unroll
        for (int e = 0; e < 4; ++e) { Om[e] = acg_override[(size_t)c * 8 + e]; sd[e] = acg_override[(size_t)c * 8 + 4 + e]; }
    }

    const bool host_noise = (eps != nullptr);
    const float* eps_c = nullptr;
    const float* w_c = nullptr;
    if (host_noise) {
        const size_t d = (size_t)draw_idx[c];
        eps_c = eps + d * (size_t)n_prop * 4;
        w_c = wun + d * (size_t)n_prop;
    }
    if (seed_dev) {             // the Philox key from device memory (a launch captured in a hipGraph must not bake the seed in)
        seed = seed_dev[0];
        call_offset = (int64_t)seed_dev[1];
    }
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint64_t gcall = (uint64_t)(call_offset + c);

    int total = 0;
    int slot = 0;
    for (int round = 0; round < max_rounds; ++round) {
        int base = 0;                                            // accepts of the proposals before this iteration's super-block
        for (int p0 = 0; p0 < n_prop; p0 += 64 * W) {
            const int p = p0 + 64 * wave + lane;
            const bool in_range = p < n_prop;
            float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 1.f, wu = 2.0f;
            if (in_range) {
                if (host_noise) {
                    const float4 e4 = *reinterpret_cast<const float4*>(eps_c + (size_t)p * 4);
                    e0 = e4.x; e1 = e4.y; e2 = e4.z; e3 = e4.w;
                    wu = w_c[p];
                } else {
                    u4 ctr;
                    ctr.x = (uint32_t)p; ctr.y = (uint32_t)round;
                    ctr.z = (uint32_t)gcall; ctr.w = (uint32_t)(gcall >> 32) & 0x7fffffffu;
                    const u4 r0 = philox4x32_10(ctr, k0, k1);
                    ctr.w |= 0x80000000u;
                    const u4 r1 = philox4x32_10(ctr, k0, k1);
                    box_muller(r0.x, r0.y, e0, e1);
                    box_muller(r0.z, r0.w, e2, e3);
                    wu = u01(r1.x);
                }
            }
            // y = std * eps ; x = y / ||y||                       (:52-53)
            const float y0 = sd[0] * e0, y1 = sd[1] * e1, y2 = sd[2] * e2, y3 = sd[3] * e3;
            const float nrm = sqrtf(y0 * y0 + y1 * y1 + y2 * y2 + y3 * y3);
            const float x0 = y0 / nrm, x1 = y1 / nrm, x2 = y2 / nrm, x3 = y3 / nrm;
            // p_Bing* = exp(-x^T A x); p_ACG* = (x^T Omega x)^-2   (:56-57)
            const float qa = x0 * A[0] * x0 + x1 * A[1] * x1 + x2 * A[2] * x2 + x3 * A[3] * x3;
            const float qo = x0 * Om[0] * x0 + x1 * Om[1] * x1 + x2 * Om[2] * x2 + x3 * Om[3] * x3;
            const float p_bing = expf(-qa);
            const float p_acg = 1.0f / (qo * qo);
            const bool acc = in_range && (wu < p_bing / (m_star * p_acg));          // :61
            const unsigned long long mask = __ballot(acc);
            const int mine = __popcll(mask);
            int before = 0, all = mine;                                              // accepts of the lower waves / of the super-block
            if (W > 1) {
                if (lane == 0) sCnt[slot][wave] = mine;
                __syncthreads();
                all = 0;
                for (int j = 0; j < W; ++j) {
                    const int cj = sCnt[slot][j];
                    before += j < wave ? cj : 0;
                    all += cj;
                }
                slot ^= 1;
            }
            const int rank = base + before + __popcll(mask & ((1ull << lane) - 1ull));
            if (acc && rank < N) {
                float Rq[9], T[9], Ro[9];
                quat_to_rotmat_dev(x0, x1, x2, x3, Rq);                          // :139
                mat3_mul_bt(Rq, V, T);                                           // R V_p^T
                mat3_mul(U, T, Ro);                                              // U_p (R V_p^T)   :140-141
                const size_t o = ((size_t)img * N + rank) * nj + joint;
#pragma unroll
                for (int e = 0; e < 9; ++e) r_out[o * 9 + e] = Ro[e];
                if (quat_out) {
                    quat_out[o * 4 + 0] = x0; quat_out[o * 4 + 1] = x1;
                    quat_out[o * 4 + 2] = x2; quat_out[o * 4 + 3] = x3;
                }
            }
            base += all;
            if (base >= N && !count_all) break;      // decided (uniform over the workgroup: every wave holds the same base)
        }
        total = base;
        if (total >= N || host_noise) break;         // host noise: the caller supplies the next draw
    }
    if (threadIdx.x == 0) accepted[c] = total;
    if (total < N && !host_noise) {
        // max_rounds exhausted (NaN / Inf concentrations make every accept test false): the reference would loop for ever
        // printing 'Failed sampling' (:68-69).  Make the failure loud instead of leaving the outputs uninitialised.
        const float nan = __builtin_nanf("");
        for (int i = threadIdx.x; i < N * 9; i += blockDim.x) r_out[(((size_t)img * N + i / 9) * nj + joint) * 9 + i % 9] = nan;
        if (quat_out)
            for (int i = threadIdx.x; i < N * 4; i += blockDim.x) quat_out[(((size_t)img * N + i / 4) * nj + joint) * 4 + i % 4] = nan;
    }
}

__global__ void quat_to_rotmat_kernel(const float* __restrict__ q, float* __restrict__ r, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float R[9];
    quat_to_rotmat_dev(q[i * 4 + 0], q[i * 4 + 1], q[i * 4 + 2], q[i * 4 + 3], R);
#pragma unroll
    for (int e = 0; e < 9; ++e) r[(size_t)i * 9 + e] = R[e];
}

// utils/rigid_transform_utils.py:80-94.  F.normalize: v / max(||v||, 1e-12).
__global__ void rot6d_to_rotmat_kernel(const float* __restrict__ x, float* __restrict__ r, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* s = x + (size_t)i * 6;                 // view(-1,3,2): a1 = s[0],s[2],s[4]; a2 = s[1],s[3],s[5]
    float a1[3] = {s[0], s[2], s[4]}, a2[3] = {s[1], s[3], s[5]};
    float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    float* o = r + (size_t)i * 9;                       // stack((b1,b2,b3), dim=-1): columns
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k * 3 + 0] = b1[k]; o[k * 3 + 1] = b2[k]; o[k * 3 + 2] = b3[k]; }
}

__global__ void rodrigues_kernel(const float* __restrict__ aa, float* __restrict__ r, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float R[9];
    rodrigues_dev(aa[i * 3 + 0], aa[i * 3 + 1], aa[i * 3 + 2], R);
#pragma unroll
    for (int e = 0; e < 9; ++e) r[(size_t)i * 9 + e] = R[e];
}

// Inputs of the one flattened SMPL call of the inference core, meshes ordered [mode (B) | T-pose (B) | samples (B N)]
// (predict/predict_poseMF_shapeGaussian_net.py:112-115, :136, utils/sampling_utils.py:178-185): thread per output element.
//   body (M, nj, 9): rows [0, B) = pose_rotmats_mode, rows [B, 2B) = identity; rows [2B, M) are the sampler's output (untouched)
//   glob_all (M, 9): glob_rotmats[b] for the mode and the samples of image b, identity for the T-pose meshes
//   betas_all (M, nb): shape mean of image b, or betas_samples (B, N, nb) for the sample meshes when given
__global__ void infer_assemble_kernel(const float* __restrict__ mode, const float* __restrict__ glob_rotmats,
                                      const float* __restrict__ loc, const float* __restrict__ betas_samples,
                                      float* __restrict__ body, float* __restrict__ glob_all, float* __restrict__ betas_all,
                                      int B, int N, int nj, int nb) {
    const long M = (long)B * (N + 2);
    const long n_body = 2L * B * nj * 9, n_glob = M * 9, n_beta = M * nb;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_body) {
        const long m = i / (nj * 9);
        const int e = (int)(i % 9);
        body[i] = m < B ? mode[i] : ((e % 4 == 0) ? 1.0f : 0.0f);
        return;
    }
    i -= n_body;
    if (i < n_glob) {
        const long m = i / 9;
        const int e = (int)(i % 9);
        const long b = m < B ? m : (m < 2L * B ? -1 : (m - 2L * B) / N);
        glob_all[i] = b < 0 ? ((e % 4 == 0) ? 1.0f : 0.0f) : glob_rotmats[b * 9 + e];
        return;
    }
    i -= n_glob;
    if (i < n_beta) {
        const long m = i / nb;
        const int e = (int)(i % nb);
        if (m >= 2L * B && betas_samples) betas_all[i] = betas_samples[(m - 2L * B) * nb + e];
        else betas_all[i] = loc[(m < B ? m : (m < 2L * B ? m - B : (m - 2L * B) / N)) * nb + e];
    }
}

}  // namespace hps

using namespace hps;

extern "C" int hps_infer_assemble(const float* mode, const float* glob_rotmats, const float* loc, const float* betas_samples,
                                  float* body, float* glob_all, float* betas_all, int B, int N, int num_body_joints,
                                  int num_betas, hps_stream_t stream) {
    if (!mode || !glob_rotmats || !loc || !body || !glob_all || !betas_all) return bad_arg("hps_infer_assemble: null pointer");
    if (B <= 0 || N < 0 || num_body_joints <= 0 || num_betas <= 0) return bad_arg("hps_infer_assemble: dims");
    const long M = (long)B * (N + 2);
    const long total = 2L * B * num_body_joints * 9 + M * 9 + M * num_betas;
    hipLaunchKernelGGL(infer_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mode,
                       glob_rotmats, loc, betas_samples, body, glob_all, betas_all, B, N, num_body_joints, num_betas);
    return check_launch("hps_infer_assemble");
}

extern "C" int hps_mf_sample(const float* pose_u, const float* pose_s, const float* pose_v, const float* bingham_a,
                             const float* acg_override, int C, int num_joints, int num_samples, int n_prop, float b, float m_star, const float* eps, const float* w,
                             const int32_t* draw_idx, uint64_t seed, int64_t call_offset, const uint64_t* seed_dev, int max_rounds,
                             float* r_out, float* quat_out, int32_t* accepted, hps_stream_t stream) {
    if (!pose_u || !pose_s || !pose_v || !r_out || !accepted) return bad_arg("hps_mf_sample: null pointer");
    if ((eps != nullptr) != (w != nullptr) || (eps && !draw_idx)) return bad_arg("hps_mf_sample: eps, w and draw_idx go together");
    if (num_joints <= 0 || C % num_joints != 0) return bad_arg("hps_mf_sample: C must be a multiple of num_joints");
    if (num_samples <= 0 || n_prop < num_samples || !(b > 0.f)) return bad_arg("hps_mf_sample: num_samples / n_prop / b");
    if (max_rounds < 1) max_rounds = 1;
    if (C == 0) return HPS_OK;
    // wavefronts per call: a function of num_samples alone (results do not depend on it; this only keeps each wave at about
    // four 64-proposal blocks until the N-th accept at the usual acceptance of one in two)
    int waves = (2 * num_samples + 255) / 256;
    waves = waves < 1 ? 1 : (waves > 8 ? 8 : waves);
    const int count_all = quat_out != nullptr;      // the Bingham entry point reports the round's total (accept_ratio, :67)
    hipLaunchKernelGGL(mf_sample_kernel, dim3(C), dim3(64 * waves), 0, (hipStream_t)stream, pose_u, pose_s, pose_v,
                       bingham_a, acg_override, num_joints, num_samples, n_prop, b, m_star, eps, w, draw_idx, seed, call_offset, seed_dev,
                       max_rounds, count_all, r_out, quat_out, accepted);
    return check_launch("hps_mf_sample");
}

extern "C" int hps_quat_to_rotmat(const float* quat, float* rotmat, int n, hps_stream_t stream) {
    if (!quat || !rotmat) return bad_arg("hps_quat_to_rotmat: null pointer");
    if (n <= 0) return HPS_OK;
    hipLaunchKernelGGL(quat_to_rotmat_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, quat, rotmat, n);
    return check_launch("hps_quat_to_rotmat");
}

extern "C" int hps_rot6d_to_rotmat(const float* x6, float* rotmat, int n, hps_stream_t stream) {
    if (!x6 || !rotmat) return bad_arg("hps_rot6d_to_rotmat: null pointer");
    if (n <= 0) return HPS_OK;
    hipLaunchKernelGGL(rot6d_to_rotmat_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x6, rotmat, n);
    return check_launch("hps_rot6d_to_rotmat");
}

extern "C" int hps_batch_rodrigues(const float* aa, float* rotmat, int n, hps_stream_t stream) {
    if (!aa || !rotmat) return bad_arg("hps_batch_rodrigues: null pointer");
    if (n <= 0) return HPS_OK;
    hipLaunchKernelGGL(rodrigues_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, aa, rotmat, n);
    return check_launch("hps_batch_rodrigues");
}

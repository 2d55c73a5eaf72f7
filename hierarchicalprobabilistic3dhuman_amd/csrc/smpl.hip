// SMPL forward on gfx950: pose prep (Rodrigues + rest joints + forward kinematics), linear blend
// skinning, joint regression, per-vertex sample uncertainty.
// Replaces models/smpl_official.py:27-41 -> smplx 0.1.26 SMPL.forward / lbs
// and utils/sampling_utils.py:189-190.  The blend-shape GEMM lives in blend_gemm.hip.
#include "hps_common.h"

namespace hps {

constexpr int MAXJ = 32;  // joints per mesh handled by one 32-lane group (SMPL: 24)

// ---------------------------------------------------------------------------------------------
// pose prep: 32 lanes per mesh (lane = joint), 8 meshes per 256-thread workgroup.
// Local transforms live in LDS; the kinematic chain is evaluated level by level so that every
// G_i = G_parent(i) * L_i is the same product the reference's index-ordered loop forms.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pose_prep_kernel(
    const float* __restrict__ glob, const float* __restrict__ body, int is_rotmat,
    const float* __restrict__ betas, int nb, const float* __restrict__ j_template,
    const float* __restrict__ j_shapedirs, const int32_t* __restrict__ parents,
    const int32_t* __restrict__ depth, int J, float* __restrict__ xt, int kp, int mp,
    float* __restrict__ a_out, float* __restrict__ j_posed, float* __restrict__ rot_out, int M) {
    __shared__ float sG[8][MAXJ][12];  // world transform (3x4 row-major) per joint
    __shared__ float sJ[8][MAXJ][3];   // rest joints
    __shared__ float sBeta[8][16];

    const int g = threadIdx.x >> 5;     // mesh slot in the workgroup
    const int j = threadIdx.x & 31;     // joint
    const int m = blockIdx.x * 8 + g;
    const bool live = (m < M) && (j < J);

    if (m < M && j < nb && j < 16) sBeta[g][j] = betas[(size_t)m * nb + j];
    __syncthreads();

    float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    float Jr[3] = {0.f, 0.f, 0.f};
    int par = -1, dep = 0;
    if (live) {
        par = parents[j];
        dep = depth[j];
        if (is_rotmat) {
            const float* src = (j == 0) ? glob + (size_t)m * 9 : body + ((size_t)m * (J - 1) + (j - 1)) * 9;
#pragma unroll
            for (int e = 0; e < 9; ++e) R[e] = src[e];
        } else {
            const float* src = (j == 0) ? glob + (size_t)m * 3 : body + ((size_t)m * (J - 1) + (j - 1)) * 3;
            rodrigues_dev(src[0], src[1], src[2], R);
        }
        // rest joint: J = J_regressor (v_template + shapedirs beta) = j_template + j_shapedirs beta
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = j_template[j * 3 + c];
            for (int l = 0; l < nb; ++l) acc += j_shapedirs[(j * 3 + c) * nb + l] * sBeta[g][l];
            Jr[c] = acc;
            sJ[g][j][c] = acc;
        }
        if (rot_out) {
#pragma unroll
            for (int e = 0; e < 9; ++e) rot_out[((size_t)m * J + j) * 9 + e] = R[e];
        }
    }
    __syncthreads();

    // local transform L = [R | J - J_parent]; root: [R | J]
    float T[12];
    if (live) {
        float rel[3] = {Jr[0], Jr[1], Jr[2]};
        if (par >= 0) {
            rel[0] -= sJ[g][par][0]; rel[1] -= sJ[g][par][1]; rel[2] -= sJ[g][par][2];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            T[r * 4 + 0] = R[r * 3 + 0]; T[r * 4 + 1] = R[r * 3 + 1]; T[r * 4 + 2] = R[r * 3 + 2];
            T[r * 4 + 3] = rel[r];
        }
        if (dep == 0) {
#pragma unroll
            for (int e = 0; e < 12; ++e) sG[g][j][e] = T[e];
        }
    }
    int max_depth = 0;
    for (int q = 0; q < J; ++q) max_depth = max(max_depth, depth[q]);
    for (int lvl = 1; lvl <= max_depth; ++lvl) {
        __syncthreads();
        if (live && dep == lvl) {
            float P[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) P[e] = sG[g][par][e];
            float Gn[12];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    Gn[r * 4 + c] = P[r * 4 + 0] * T[0 * 4 + c] + P[r * 4 + 1] * T[1 * 4 + c] + P[r * 4 + 2] * T[2 * 4 + c];
                Gn[r * 4 + 3] = P[r * 4 + 0] * T[3] + P[r * 4 + 1] * T[7] + P[r * 4 + 2] * T[11] + P[r * 4 + 3];
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) { T[e] = Gn[e]; sG[g][j][e] = Gn[e]; }
        }
    }
    // T now holds the world transform G_j (roots kept their local transform).
    if (live) {
        float* ao = a_out + ((size_t)m * J + j) * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            ao[r * 4 + 0] = T[r * 4 + 0]; ao[r * 4 + 1] = T[r * 4 + 1]; ao[r * 4 + 2] = T[r * 4 + 2];
            // A = G - pad(G [J;0]) : translation minus rotated rest joint
            ao[r * 4 + 3] = T[r * 4 + 3] - (T[r * 4 + 0] * Jr[0] + T[r * 4 + 1] * Jr[1] + T[r * 4 + 2] * Jr[2]);
            j_posed[((size_t)m * J + j) * 3 + r] = T[r * 4 + 3];
        }
        // blend operand, k-major: pose feature rows nb + 9 (j-1) + e = (R_j - I)
        if (j >= 1) {
#pragma unroll
            for (int e = 0; e < 9; ++e)
                xt[(size_t)(nb + 9 * (j - 1) + e) * mp + m] = R[e] - ((e % 4 == 0) ? 1.0f : 0.0f);
        }
    }
    if (m < M) {
        // betas rows and zero padding rows (lanes stride over them)
        for (int k = j; k < nb; k += 32) xt[(size_t)k * mp + m] = sBeta[g][k];
        for (int k = nb + 9 * (J - 1) + j; k < kp; k += 32) xt[(size_t)k * mp + m] = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// LBS: one lane per vertex; the lane keeps its K (joint, weight) pairs in registers and walks the
// meshes of its chunk.  A[m] (J x 12 floats) is staged through LDS, G meshes at a time, double
// buffered so there is one barrier per group.  v_posed / verts move as lane-contiguous 12-byte
// records (768 contiguous bytes per wave instruction).
// ---------------------------------------------------------------------------------------------
template <int K, int G>
__global__ __launch_bounds__(256) void lbs_kernel(const f3* __restrict__ v_posed, const float* __restrict__ a,
                                                  const int32_t* __restrict__ w_idx, const float* __restrict__ w_val,
                                                  int J, const float* __restrict__ transl, f3* __restrict__ verts,
                                                  int M, int V, int meshes_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sA[];  // [2][G][J*12]
    const int v = blockIdx.x * 256 + threadIdx.x;
    const bool live = v < V;
    const int m_begin = blockIdx.y * meshes_per_block;
    const int m_end = min(M, m_begin + meshes_per_block);
    const int a_stride = J * 12;

    int idx[K];
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        idx[k] = live ? w_idx[(size_t)v * K + k] * 12 : 0;
        w[k] = live ? w_val[(size_t)v * K + k] : 0.0f;
    }

    auto stage = [&](int buf, int m0) {
        const int n4 = (G * a_stride) >> 2;  // a_stride multiple of 4
        const int avail = (min(M, m0 + G) - m0) * a_stride >> 2;
        const float4* src = reinterpret_cast<const float4*>(a + (size_t)m0 * a_stride);
        float4* dst = reinterpret_cast<float4*>(sA + buf * G * a_stride);
        for (int i = threadIdx.x; i < n4; i += 256)
            if (i < avail) dst[i] = src[i];
    };

    int buf = 0;
    stage(0, m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += G) {
        __syncthreads();
        if (m0 + G < m_end) stage(buf ^ 1, m0 + G);
        f3 p[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int m = m0 + g;
            if (live && m < m_end) p[g] = v_posed[(size_t)m * V + v];
            else p[g] = f3{0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int m = m0 + g;
            const float* Am = sA + (buf * G + g) * a_stride;
            float T[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float4* t4 = reinterpret_cast<const float4*>(Am + idx[k]);
                float4 r0 = t4[0], r1 = t4[1], r2 = t4[2];
                T[0] += w[k] * r0.x; T[1] += w[k] * r0.y; T[2] += w[k] * r0.z; T[3] += w[k] * r0.w;
                T[4] += w[k] * r1.x; T[5] += w[k] * r1.y; T[6] += w[k] * r1.z; T[7] += w[k] * r1.w;
                T[8] += w[k] * r2.x; T[9] += w[k] * r2.y; T[10] += w[k] * r2.z; T[11] += w[k] * r2.w;
            }
            f3 o;
            o.x = T[0] * p[g].x + T[1] * p[g].y + T[2] * p[g].z + T[3];
            o.y = T[4] * p[g].x + T[5] * p[g].y + T[6] * p[g].z + T[7];
            o.z = T[8] * p[g].x + T[9] * p[g].y + T[10] * p[g].z + T[11];
            if (live && m < m_end) {
                if (transl) { o.x += transl[m * 3 + 0]; o.y += transl[m * 3 + 1]; o.z += transl[m * 3 + 2]; }
                verts[(size_t)m * V + v] = o;
            }
        }
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------
// joints: one 128-thread workgroup per mesh; thread r evaluates CSR row r on the mesh's vertices.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void joints_kernel(const float* __restrict__ verts, const float* __restrict__ j_posed,
                                                     const int32_t* __restrict__ csr_ptr, const int32_t* __restrict__ csr_col,
                                                     const float* __restrict__ csr_val, int n_rows, int J,
                                                     const float* __restrict__ transl, float* __restrict__ joints, int V) {
    const int m = blockIdx.x;
    const int n_out = J + n_rows;
    float tx = 0.f, ty = 0.f, tz = 0.f;
    if (transl) { tx = transl[m * 3 + 0]; ty = transl[m * 3 + 1]; tz = transl[m * 3 + 2]; }
    const float* vm = verts + (size_t)m * V * 3;  // verts already include transl
    for (int r = threadIdx.x; r < n_out; r += blockDim.x) {
        float x, y, z;
        if (r < J) {
            const float* s = j_posed + ((size_t)m * J + r) * 3;
            x = s[0] + tx; y = s[1] + ty; z = s[2] + tz;
        } else {
            x = y = z = 0.f;
            for (int e = csr_ptr[r - J]; e < csr_ptr[r - J + 1]; ++e) {
                const float wv = csr_val[e];
                const float* s = vm + (size_t)csr_col[e] * 3;
                x += wv * s[0]; y += wv * s[1]; z += wv * s[2];
            }
        }
        float* d = joints + ((size_t)m * n_out + r) * 3;
        d[0] = x; d[1] = y; d[2] = z;
    }
}

// ---------------------------------------------------------------------------------------------
// vertex uncertainty: lane per vertex, two sweeps over the image's N samples (the second sweep
// re-reads lines the first one left in L2 / Infinity Cache).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uncertainty_kernel(const f3* __restrict__ verts, float* __restrict__ unc,
                                                          int N, int V) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (v >= V) return;
    const f3* base = verts + (size_t)b * N * V + v;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int s = 0; s < N; ++s) {
        f3 p = base[(size_t)s * V];
        sx += p.x; sy += p.y; sz += p.z;
    }
    const float mx = sx / N, my = sy / N, mz = sz / N;
    float acc = 0.f;
    for (int s = 0; s < N; ++s) {
        f3 p = base[(size_t)s * V];
        const float dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
        acc += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    unc[(size_t)b * V + v] = acc / N;
}

}  // namespace hps

using namespace hps;

extern "C" int hps_smpl_pose_prep(const float* glob, const float* body, int is_rotmat, const float* betas,
                                  int num_betas, const float* j_template, const float* j_shapedirs,
                                  const int32_t* parents, const int32_t* depth, int num_joints, float* xt, int kp,
                                  int mp, float* a, float* j_posed, float* rot_out, int M, hps_stream_t stream) {
    if (!glob || !body || !betas || !j_template || !j_shapedirs || !parents || !depth || !xt || !a || !j_posed)
        return bad_arg("hps_smpl_pose_prep: null pointer");
    if (num_joints < 1 || num_joints > MAXJ || num_betas < 0 || num_betas > 16)
        return bad_arg("hps_smpl_pose_prep: num_joints must be 1..32 and num_betas 0..16");
    if (kp < num_betas + 9 * (num_joints - 1) || mp < M) return bad_arg("hps_smpl_pose_prep: kp/mp too small");
    if (M <= 0) return HPS_OK;
    hipLaunchKernelGGL(pose_prep_kernel, dim3(ceil_div(M, 8)), dim3(256), 0, (hipStream_t)stream, glob, body,
                       is_rotmat, betas, num_betas, j_template, j_shapedirs, parents, depth, num_joints, xt, kp, mp,
                       a, j_posed, rot_out, M);
    return check_launch("hps_smpl_pose_prep");
}

template <int K>
static int launch_lbs(const float* v_posed, const float* a, const int32_t* w_idx, const float* w_val, int J,
                      const float* transl, float* verts, int M, int V, hipStream_t s) {
    constexpr int G = 4;
    // enough workgroups to fill 256 CUs several times over, few enough that the per-lane weights are reused
    int mpb = 32;
    while (mpb > G && (size_t)ceil_div(V, 256) * ceil_div(M, mpb) < 2048) mpb >>= 1;
    dim3 grid(ceil_div(V, 256), ceil_div(M, mpb));
    size_t lds = (size_t)2 * G * J * 12 * sizeof(float);
    hipLaunchKernelGGL((lbs_kernel<K, G>), grid, dim3(256), lds, s, reinterpret_cast<const f3*>(v_posed), a, w_idx,
                       w_val, J, transl, reinterpret_cast<f3*>(verts), M, V, mpb);
    return check_launch("hps_smpl_lbs");
}

extern "C" int hps_smpl_lbs(const float* v_posed, const float* a, const int32_t* w_idx, const float* w_val, int K,
                            int num_joints, const float* transl, float* verts, int M, int V, hps_stream_t stream) {
    if (!v_posed || !a || !w_idx || !w_val || !verts) return bad_arg("hps_smpl_lbs: null pointer");
    if (num_joints < 1 || num_joints > 64) return bad_arg("hps_smpl_lbs: num_joints");
    if (M <= 0 || V <= 0) return HPS_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 4: return launch_lbs<4>(v_posed, a, w_idx, w_val, num_joints, transl, verts, M, V, s);
        case 8: return launch_lbs<8>(v_posed, a, w_idx, w_val, num_joints, transl, verts, M, V, s);
        case 12: return launch_lbs<12>(v_posed, a, w_idx, w_val, num_joints, transl, verts, M, V, s);
        case 24: return launch_lbs<24>(v_posed, a, w_idx, w_val, num_joints, transl, verts, M, V, s);
        default: set_error("hps_smpl_lbs: K=%d unsupported (4, 8, 12, 24)", K); return HPS_E_UNSUPPORTED;
    }
}

extern "C" int hps_smpl_joints(const float* verts, const float* j_posed, const int32_t* csr_ptr,
                               const int32_t* csr_col, const float* csr_val, int n_rows, int num_joints,
                               const float* transl, float* joints, int M, int V, hps_stream_t stream) {
    if (!verts || !j_posed || !csr_ptr || !csr_col || !csr_val || !joints) return bad_arg("hps_smpl_joints: null pointer");
    if (M <= 0) return HPS_OK;
    hipLaunchKernelGGL(joints_kernel, dim3(M), dim3(128), 0, (hipStream_t)stream, verts, j_posed, csr_ptr, csr_col,
                       csr_val, n_rows, num_joints, transl, joints, V);
    return check_launch("hps_smpl_joints");
}

extern "C" int hps_vertex_uncertainty(const float* verts, float* unc, int B, int N, int V, hps_stream_t stream) {
    if (!verts || !unc) return bad_arg("hps_vertex_uncertainty: null pointer");
    if (B <= 0 || N <= 0 || V <= 0) return HPS_OK;
    hipLaunchKernelGGL(uncertainty_kernel, dim3(ceil_div(V, 256), B), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f3*>(verts), unc, N, V);
    return check_launch("hps_vertex_uncertainty");
}

// SMPL forward on gfx950: pose prep (Rodrigues + rest joints + forward kinematics), linear blend
// skinning, joint regression, per-vertex sample uncertainty.
// Replaces models/smpl_official.py:27-41 -> smplx 0.1.26 SMPL.forward / lbs
// and utils/sampling_utils.py:189-190.  The blend-shape GEMM lives in blend_gemm.hip.
#include "hps_common.h"

namespace hps {

constexpr int MAXJ = 32;  // joints per mesh handled by one 32-lane group (SMPL: 24)

#ifdef HPS_DEV_BUILD
// ---------------------------------------------------------------------------------------------
// pose prep, first generation (dev library only: the bit-level cross-check of the kernel below): 32 lanes per mesh (lane = joint), 8 meshes per 256-thread workgroup.
// Local transforms live in LDS; the kinematic chain is evaluated level by level so that every
// G_i = G_parent(i) * L_i is the same product the reference's index-ordered loop forms.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pose_prep_kernel_v1(
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

#endif

// ---------------------------------------------------------------------------------------------
// pose prep: 32 lanes per mesh (lane = joint), 8 meshes per 256-thread workgroup.
// Local transforms live in LDS; the kinematic chain is evaluated level by level so that every
// G_i = G_parent(i) * L_i is the same product the reference's index-ordered loop forms.
//
// Second generation (round 5; the first stays in the dev library as the bit-level cross-check).  (1) `max depth` was a loop of J
// dependent scalar loads in front of the kinematic levels -- now a shuffle reduction of the lanes' own depths; (2) the rest joints
// read 3 x nb values of j_shapedirs per lane one after another from global memory -- the joint regressor (J x 3 x (nb + 1) floats)
// is now staged in LDS by the whole workgroup with coalesced loads (the sums keep their order: identical bits).  Measured alone at
// 6 528 meshes (tests/dev/pair_time.py): 19.5 -> 19.0 us -- neither chain was what the kernel waits for.  What did shorten it was
// 16 meshes per workgroup with the k-major operand transposed through LDS for 64-byte runs (512 threads, 57 KB: 15.1 us alone) --
// and that form waited three times as long for a CU in the pipelined loop (82 us on average, up to 0.39 ms, in front of the
// exclusive mesh kernel: it runs beside the encoder's persistent kernels, which leave a CU little LDS and no registers), so the
// workgroup stays SMALL on purpose (256 threads, 22 KB).  Requesting a lane's 48 coefficients at once from global memory instead
// of staging them was slower than the first generation (25.4 us).
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
    __shared__ float sJT[MAXJ * 3];       // j_template
    __shared__ float sJS[MAXJ * 3 * 16];  // j_shapedirs

    const int g = threadIdx.x >> 5;     // mesh slot in the workgroup
    const int j = threadIdx.x & 31;     // joint
    const int m = blockIdx.x * 8 + g;
    const bool live = (m < M) && (j < J);

    if (m < M && j < nb && j < 16) sBeta[g][j] = betas[(size_t)m * nb + j];
    for (int i = threadIdx.x; i < J * 3; i += 256) sJT[i] = j_template[i];
    for (int i = threadIdx.x; i < J * 3 * nb; i += 256) sJS[i] = j_shapedirs[i];
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
            float acc = sJT[j * 3 + c];
            for (int l = 0; l < nb; ++l) acc += sJS[(j * 3 + c) * nb + l] * sBeta[g][l];
            Jr[c] = acc;
            sJ[g][j][c] = acc;
        }
        if (rot_out) {
#pragma unroll
            for (int e = 0; e < 9; ++e) rot_out[((size_t)m * J + j) * 9 + e] = R[e];
        }
    }
    // deepest kinematic level: a reduction over the 32 lanes of a mesh (every mesh slot computes the same value)
    int max_depth = (j < J) ? depth[j] : 0;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) max_depth = max(max_depth, __shfl_xor(max_depth, d));
    max_depth = __builtin_amdgcn_readfirstlane(max_depth);
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
// LBS.  A workgroup owns VPT*256 consecutive vertices for its whole life (each lane keeps the K
// (joint, weight) pairs of its VPT vertices in registers) and walks a contiguous chunk of meshes, G at a
// time.  A[m] (J x 12 floats) is staged through LDS, double buffered: one barrier per G meshes.
// v_posed / verts move as lane-contiguous 12-byte records (768 contiguous bytes per wave instruction).
// The grid is sized to be resident in one round (no tail) and mapped so that the workgroups sharing a mesh
// chunk -- the ones that read the same A -- run on the same XCD and hit its L2.
// ---------------------------------------------------------------------------------------------
template <int K, int G, int VPT>
__global__ __launch_bounds__(256) void lbs_kernel(const float* __restrict__ v_posed_f, int ld_vposed,
                                                  const float* __restrict__ a,
                                                  const int32_t* __restrict__ w_idx, const float* __restrict__ w_val,
                                                  int J, const float* __restrict__ transl, f3* __restrict__ verts,
                                                  int M, int V, int meshes_per_block, int n_vtiles) {
    extern __shared__ __attribute__((aligned(16))) float sA[];  // [2][G][J*12]
    // block id -> (mesh chunk, vertex tile): ids congruent mod 8 (one XCD) share chunks
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int chunk = (local / n_vtiles) * 8 + xcd, vt = local % n_vtiles;
    const int m_begin = chunk * meshes_per_block;
    if (m_begin >= M) return;
    const int m_end = min(M, m_begin + meshes_per_block);
    const int a_stride = J * 12;

    int vtx[VPT];
    bool live[VPT];
    int idx[VPT][K];
    float w[VPT][K];
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        vtx[q] = vt * (256 * VPT) + q * 256 + threadIdx.x;
        live[q] = vtx[q] < V;
        if (!live[q]) vtx[q] = V - 1;          // clamp: loads stay in range, stores are predicated
#pragma unroll
        for (int k = 0; k < K; ++k) {
            idx[q][k] = w_idx[(size_t)vtx[q] * K + k] * 12;
            w[q][k] = w_val[(size_t)vtx[q] * K + k];
        }
    }

    auto stage = [&](int buf, int m0) {
        const int n4 = (min(m_end, m0 + G) - m0) * a_stride >> 2;  // a_stride is a multiple of 4
        const float4* src = reinterpret_cast<const float4*>(a + (size_t)m0 * a_stride);
        float4* dst = reinterpret_cast<float4*>(sA + buf * G * a_stride);
        for (int i = threadIdx.x; i < n4; i += 256) dst[i] = src[i];
    };

    int buf = 0;
    stage(0, m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += G) {
        __syncthreads();
        if (m0 + G < m_end) stage(buf ^ 1, m0 + G);
        f3 p[G][VPT];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int m = min(m0 + g, m_end - 1);   // clamp: the tail group re-reads the last mesh
#pragma unroll
            for (int q = 0; q < VPT; ++q)
                p[g][q] = reinterpret_cast<const f3*>(v_posed_f + (size_t)m * ld_vposed)[vtx[q]];     // row pitch ld_vposed floats
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int m = m0 + g;
            const float* Am = sA + (buf * G + g) * a_stride;
            float tx = 0.f, ty = 0.f, tz = 0.f;
            if (transl && m < m_end) { tx = transl[m * 3 + 0]; ty = transl[m * 3 + 1]; tz = transl[m * 3 + 2]; }
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                const f3 o = skin_vertex<K>(Am, idx[q], w[q], p[g][q], tx, ty, tz);
                if (live[q] && m < m_end) verts[(size_t)m * V + vtx[q]] = o;
            }
        }
        buf ^= 1;
    }
}

#ifdef HPS_DEV_BUILD
// ---------------------------------------------------------------------------------------------
// joints, first generation (dev library only: the bit-level cross-check of the kernel below): one 128-thread workgroup per mesh; thread r evaluates CSR row r on the mesh's vertices.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void joints_kernel_v1(const float* __restrict__ verts, const float* __restrict__ j_posed,
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
            // four entries of the row in flight (column -> vertex is a dependent load chain); added in row order
            const int e1 = csr_ptr[r - J + 1];
            int e = csr_ptr[r - J];
            for (; e + 4 <= e1; e += 4) {
                float wv[4];
                f3 p[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wv[i] = csr_val[e + i];
                    p[i] = *reinterpret_cast<const f3*>(vm + (size_t)csr_col[e + i] * 3);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { x += wv[i] * p[i].x; y += wv[i] * p[i].y; z += wv[i] * p[i].z; }
            }
            for (; e < e1; ++e) {
                const float wv = csr_val[e];
                const float* s = vm + (size_t)csr_col[e] * 3;
                x += wv * s[0]; y += wv * s[1]; z += wv * s[2];
            }
        }
        float* d = joints + ((size_t)m * n_out + r) * 3;
        d[0] = x; d[1] = y; d[2] = z;
    }
}

#endif

// ---------------------------------------------------------------------------------------------
// joints: one 128-thread workgroup per mesh; thread r evaluates CSR row r (a vertex pick or a regressed joint) on the mesh's
// vertices, four entries in flight, summed as ONE chain of explicit fused multiply-adds in row order.
//
// Round 5 kept the first generation's shape and only pinned its arithmetic: hipcc's SLP vectoriser had compiled the four-entry
// groups `x += w * p` as a mix of v_pk_mul + add and v_pk_fma -- two of every four products rounded separately, an accident of that
// build (the first generation in the dev library still has it: results agree within one unit in the last place).  Two re-designs
// were built and measured against it at 6 528 meshes with the vertices coming from HBM (tests/dev/pair_time.py,
// profiles/r05_experiments.txt): thread = row with the row's entries in registers and four meshes per workgroup -- 74 us against
// 60 (padded 12-slot rows issue three times the loads); one gather per (entry, mesh) into LDS, then the chains -- 59 us.  The kernel
// is bound by the rate of scattered 12-byte requests (276 per mesh), not by its dependency chains, and the small footprint matters
// more: it runs beside the next batch's encoder, whose persistent kernels leave a CU almost no LDS or registers (the LDS form
// averaged 69 us in the loop where this one takes 45).
// ---------------------------------------------------------------------------------------------
// the rows of mesh m, row r = t, t + 128, ... (t < 128: the lane's index in the mesh's group of 128)
__device__ __forceinline__ void joints_of_mesh(int m, int t, const float* __restrict__ verts, const float* __restrict__ j_posed,
                                               const int32_t* __restrict__ csr_ptr, const int32_t* __restrict__ csr_col,
                                               const float* __restrict__ csr_val, int n_rows, int J,
                                               const float* __restrict__ transl, float* __restrict__ joints, int V) {
    const int n_out = J + n_rows;
    float tx = 0.f, ty = 0.f, tz = 0.f;
    if (transl) { tx = transl[m * 3 + 0]; ty = transl[m * 3 + 1]; tz = transl[m * 3 + 2]; }
    const float* vm = verts + (size_t)m * V * 3;  // verts already include transl
    for (int r = t; r < n_out; r += 128) {
        float x, y, z;
        if (r < J) {
            const float* s = j_posed + ((size_t)m * J + r) * 3;
            x = s[0] + tx; y = s[1] + ty; z = s[2] + tz;
        } else {
            x = y = z = 0.f;
            // four entries of the row in flight (column -> vertex is a dependent load chain); added in row order
            const int e1 = csr_ptr[r - J + 1];
            int e = csr_ptr[r - J];
            for (; e + 4 <= e1; e += 4) {
                float wv[4];
                f3 p[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wv[i] = csr_val[e + i];
                    p[i] = *reinterpret_cast<const f3*>(vm + (size_t)csr_col[e + i] * 3);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x = __builtin_fmaf(wv[i], p[i].x, x); y = __builtin_fmaf(wv[i], p[i].y, y); z = __builtin_fmaf(wv[i], p[i].z, z);
                }
            }
            for (; e < e1; ++e) {
                const float wv = csr_val[e];
                const float* s = vm + (size_t)csr_col[e] * 3;
                x = __builtin_fmaf(wv, s[0], x); y = __builtin_fmaf(wv, s[1], y); z = __builtin_fmaf(wv, s[2], z);
            }
        }
        float* d = joints + ((size_t)m * n_out + r) * 3;
        d[0] = x; d[1] = y; d[2] = z;
    }
}

__global__ __launch_bounds__(128) void joints_kernel(const float* __restrict__ verts, const float* __restrict__ j_posed,
                                                     const int32_t* __restrict__ csr_ptr, const int32_t* __restrict__ csr_col,
                                                     const float* __restrict__ csr_val, int n_rows, int J,
                                                     const float* __restrict__ transl, float* __restrict__ joints, int V) {
    joints_of_mesh(blockIdx.x, threadIdx.x, verts, j_posed, csr_ptr, csr_col, csr_val, n_rows, J, transl, joints, V);
}

// ---------------------------------------------------------------------------------------------
// vertex uncertainty: lane per vertex, two sweeps over the image's N samples (the second sweep
// re-reads lines the first one left in L2 / Infinity Cache).
// ---------------------------------------------------------------------------------------------
// one definition with explicit fused multiply-adds: every uncertainty kernel rounds a distance the same way, whatever the
// compiler would contract in its surroundings
__device__ __forceinline__ float dist3(float dx, float dy, float dz) {
    return sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
}

__global__ __launch_bounds__(256) void uncertainty_kernel(const f3* __restrict__ verts, float* __restrict__ unc,
                                                          int N, int V) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (v >= V) return;
    const f3* base = verts + (size_t)b * N * V + v;
    // eight samples requested before the first is added (the sums stay in sample order: the same bits as one by one)
    constexpr int UB = 8;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int s = 0;
    for (; s + UB <= N; s += UB) {
        f3 p[UB];
#pragma unroll
        for (int i = 0; i < UB; ++i) p[i] = base[(size_t)(s + i) * V];
#pragma unroll
        for (int i = 0; i < UB; ++i) { sx += p[i].x; sy += p[i].y; sz += p[i].z; }
    }
    for (; s < N; ++s) {
        const f3 p = base[(size_t)s * V];
        sx += p.x; sy += p.y; sz += p.z;
    }
    const float mx = sx / N, my = sy / N, mz = sz / N;
    float acc = 0.f;
    for (s = 0; s + UB <= N; s += UB) {
        f3 p[UB];
#pragma unroll
        for (int i = 0; i < UB; ++i) p[i] = base[(size_t)(s + i) * V];
#pragma unroll
        for (int i = 0; i < UB; ++i) acc += dist3(p[i].x - mx, p[i].y - my, p[i].z - mz);
    }
    for (; s < N; ++s) {
        const f3 p = base[(size_t)s * V];
        acc += dist3(p.x - mx, p.y - my, p.z - mz);
    }
    unc[(size_t)b * V + v] = acc / N;
}

constexpr int UG = 8;             // sample groups of the single-pass uncertainty kernels
#ifdef HPS_DEV_BUILD
static int g_unc_mode = 0;        // hps_dev_unc_mode: 0 = automatic, 1 = two-sweep, 2 = LDS with 128 vertices, 3 = LDS with 64, 4 = registers, 5 = one sweep with 32 vertices, 7 = registers in the earlier block order
#else
constexpr int g_unc_mode = 0;     // product library: no process-global switches
#endif

#ifdef HPS_DEV_BUILD              // the LDS-resident forms the register-resident kernel replaced (same bits): A/B runs only
// Single-pass form: one workgroup = one image x 128 vertices, 1024 lanes = 8 sample groups x 128 vertices.  The
// image's N x 128 vertex positions are read from HBM once into LDS (N * 1536 bytes: N <= 100 fits the 160 KiB of a
// CU), the mean and the mean distance are then formed from LDS -- half the HBM traffic of the two-sweep kernel.
// UV = 64 halves the footprint (N = 100: 80 KiB): a second workgroup -- e.g. of the next batch's convolution, beside
// which this kernel runs in the pipelined loop -- can share the CU.

template <int UV>
__global__ __launch_bounds__(UV * UG) void uncertainty_lds_kernel(const f3* __restrict__ verts, float* __restrict__ unc,
                                                                  int N, int V) {
    extern __shared__ __attribute__((aligned(16))) float sU[];   // [N][3][UV] samples, then [UG/2][3][UV] reduction slots
    float* sRed = sU + (size_t)N * 3 * UV;
    const int v = threadIdx.x & (UV - 1), g = threadIdx.x / UV;
    const int vg = blockIdx.x * UV + v, b = blockIdx.y;
    const bool live = vg < V;
    const f3* base = verts + (size_t)b * N * V + (live ? vg : V - 1);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int s = g; s < N; s += UG) {
        const f3 p = base[(size_t)s * V];
        sU[(s * 3 + 0) * UV + v] = p.x; sU[(s * 3 + 1) * UV + v] = p.y; sU[(s * 3 + 2) * UV + v] = p.z;
        sx += p.x; sy += p.y; sz += p.z;
    }
    // two-step reduction over the 8 sample groups through a 4-slot buffer (keeps N = 100 within 160 KiB)
    constexpr int H = UG / 2;
    if (g >= H) { sRed[((g - H) * 3 + 0) * UV + v] = sx; sRed[((g - H) * 3 + 1) * UV + v] = sy; sRed[((g - H) * 3 + 2) * UV + v] = sz; }
    __syncthreads();
    if (g < H) {
        sx += sRed[(g * 3 + 0) * UV + v]; sy += sRed[(g * 3 + 1) * UV + v]; sz += sRed[(g * 3 + 2) * UV + v];
    }
    __syncthreads();
    if (g < H) { sRed[(g * 3 + 0) * UV + v] = sx; sRed[(g * 3 + 1) * UV + v] = sy; sRed[(g * 3 + 2) * UV + v] = sz; }
    __syncthreads();
    float mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll
    for (int q = 0; q < H; ++q) { mx += sRed[(q * 3 + 0) * UV + v]; my += sRed[(q * 3 + 1) * UV + v]; mz += sRed[(q * 3 + 2) * UV + v]; }
    mx /= N; my /= N; mz /= N;
    __syncthreads();                                    // everyone has read the partial sums before they are reused
    float acc = 0.f;
    for (int s = g; s < N; s += UG) {
        const float dx = sU[(s * 3 + 0) * UV + v] - mx, dy = sU[(s * 3 + 1) * UV + v] - my, dz = sU[(s * 3 + 2) * UV + v] - mz;
        acc += dist3(dx, dy, dz);
    }
    if (g >= H) sRed[(g - H) * UV + v] = acc;
    __syncthreads();
    if (g < H) acc += sRed[g * UV + v];
    __syncthreads();
    if (g < H) sRed[g * UV + v] = acc;
    __syncthreads();
    if (g == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < UG / 2; ++q) t += sRed[q * UV + v];
        unc[(size_t)b * V + vg] = t / N;
    }
}

#endif  // HPS_DEV_BUILD

// Register-resident single pass: a workgroup is 64 vertices x 8 sample groups (one wave per group); lane (v, g) keeps
// its samples s = g, g + 8, ... (SPT of them, 3 floats each) in registers, so the samples are read from HBM once, the
// only LDS is the 3 KB reduction buffer, and many workgroups share a CU (the LDS-resident form allows one).
// Same sample-to-group assignment, same summation order and reduction tree as uncertainty_lds_kernel: identical bits.
// Block order (round 6): images fastest, vertex panels DESCENDING.  The pass follows the mesh kernel, which walks the panels in ascending
// order: the panels written last -- the ones still in the memory-side cache -- are read first, and reading them does not push the older,
// still dirty lines out through HBM.  Measured in the pipelined loop against panels-fastest / images-ascending (hps_dev_unc_mode 7, same
// kernel): 22.23 k against 22.10-22.16 k images/s, three interleaved pairs (profiles/r06_ab.txt).  The order of the blocks changes no sum.
template <int SPT>
__device__ __forceinline__ void uncertainty_reg_body(const f3* __restrict__ verts, float* __restrict__ unc, int N, int V, int panel,
                                                     int b) {
    constexpr int RV = 64, H = UG / 2;
    __shared__ float sRed[H * 3 * RV];
    const int v = threadIdx.x & (RV - 1), g = threadIdx.x / RV;
    const int vg = panel * RV + v;
    const bool live = vg < V;
    const f3* base = verts + (size_t)b * N * V + (live ? vg : V - 1);
    // Guard-free: a sample beyond N re-reads sample N - 1 and contributes +0 (x + 0 = x bit for bit; the sums start at +0 and
    // can never be -0).  With `if (s < N) { load; add; }` hipcc put every load into a block of its own that ends in
    // s_waitcnt vmcnt(0): thirteen serialised HBM round trips per lane.
    f3 p[SPT];
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int i = 0; i < SPT; ++i) p[i] = base[(size_t)min(g + UG * i, N - 1) * V];
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        const bool ok = g + UG * i < N;
        sx += ok ? p[i].x : 0.0f; sy += ok ? p[i].y : 0.0f; sz += ok ? p[i].z : 0.0f;
    }
    if (g >= H) { sRed[((g - H) * 3 + 0) * RV + v] = sx; sRed[((g - H) * 3 + 1) * RV + v] = sy; sRed[((g - H) * 3 + 2) * RV + v] = sz; }
    __syncthreads();
    if (g < H) {
        sx += sRed[(g * 3 + 0) * RV + v]; sy += sRed[(g * 3 + 1) * RV + v]; sz += sRed[(g * 3 + 2) * RV + v];
    }
    __syncthreads();
    if (g < H) { sRed[(g * 3 + 0) * RV + v] = sx; sRed[(g * 3 + 1) * RV + v] = sy; sRed[(g * 3 + 2) * RV + v] = sz; }
    __syncthreads();
    float mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll
    for (int q = 0; q < H; ++q) { mx += sRed[(q * 3 + 0) * RV + v]; my += sRed[(q * 3 + 1) * RV + v]; mz += sRed[(q * 3 + 2) * RV + v]; }
    mx /= N; my /= N; mz /= N;
    __syncthreads();                                    // everyone has read the partial sums before they are reused
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        const float dx = p[i].x - mx, dy = p[i].y - my, dz = p[i].z - mz;
        const float dist = dist3(dx, dy, dz);
        acc += (g + UG * i < N) ? dist : 0.0f;
    }
    if (g >= H) sRed[(g - H) * RV + v] = acc;
    __syncthreads();
    if (g < H) acc += sRed[g * RV + v];
    __syncthreads();
    if (g < H) sRed[g * RV + v] = acc;
    __syncthreads();
    if (g == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < H; ++q) t += sRed[q * RV + v];
        unc[(size_t)b * V + vg] = t / N;
    }
}

template <int SPT, bool FWD = false>
__global__ __launch_bounds__(512) void uncertainty_reg_kernel(const f3* __restrict__ verts, float* __restrict__ unc,
                                                              int N, int V) {
    if (FWD) uncertainty_reg_body<SPT>(verts, unc, N, V, (int)blockIdx.x, (int)blockIdx.y);
    else uncertainty_reg_body<SPT>(verts, unc, N, V, (int)(gridDim.y - 1 - blockIdx.y), (int)blockIdx.x);
}

// The joint regression of ALL meshes of a call and the uncertainty pass over its sample meshes in ONE launch (round 6): both read what
// the mesh kernel has just written and neither depends on the other, but as two launches on one stream they ran one after the other --
// 18 us of joint regression (6 528 x 128 threads: a fraction of the chip) plus a launch boundary in front of the HBM-bound pass.
// Grid (B, joint rows + panels): rows y < jrows are the joint regression's, four meshes per 512-thread workgroup (each 128 lanes are
// one joints_kernel workgroup: the same sums), the rest is uncertainty_reg_kernel's grid.  Identical bits to the two launches.
struct JointArgs {
    const float* picked; const float* j_posed; const int32_t* csr_ptr; const int32_t* csr_slot; const float* csr_val;
    const float* transl; float* joints;
    int n_rows, J, M, n_picked, jrows;
};
template <int SPT>
__global__ __launch_bounds__(512) void uncertainty_joints_kernel(const f3* __restrict__ verts, float* __restrict__ unc, int N, int V,
                                                                 const JointArgs ja) {
    if ((int)blockIdx.y < ja.jrows) {                                  // (workgroup-uniform)
        const int m = 4 * ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) + (int)(threadIdx.x >> 7);
        if (m < ja.M)
            joints_of_mesh(m, threadIdx.x & 127, ja.picked, ja.j_posed, ja.csr_ptr, ja.csr_slot, ja.csr_val, ja.n_rows, ja.J, ja.transl,
                           ja.joints, ja.n_picked);
        return;
    }
    uncertainty_reg_body<SPT>(verts, unc, N, V, (int)(gridDim.y - 1 - blockIdx.y), (int)blockIdx.x);
}

template <int SPT>
static int launch_unc_joints(const float* verts, float* unc, int B, int N, int V, JointArgs ja, hipStream_t s) {
    ja.jrows = ceil_div(ceil_div(ja.M, 4), B);
    hipLaunchKernelGGL(uncertainty_joints_kernel<SPT>, dim3(B, ja.jrows + ceil_div(V, 64)), dim3(512), 0, s,
                       reinterpret_cast<const f3*>(verts), unc, N, V, ja);
    return check_launch("hps_joints_and_uncertainty");
}

template <int SPT>
static int launch_unc_reg(const float* verts, float* unc, int B, int N, int V, hipStream_t s) {
#ifdef HPS_DEV_BUILD
    if (g_unc_mode == 7) {      // A/B: the earlier block order (panels fastest, ascending)
        hipLaunchKernelGGL((uncertainty_reg_kernel<SPT, true>), dim3(ceil_div(V, 64), B), dim3(512), 0, s, reinterpret_cast<const f3*>(verts),
                           unc, N, V);
        return check_launch("hps_vertex_uncertainty");
    }
#endif
    hipLaunchKernelGGL(uncertainty_reg_kernel<SPT>, dim3(B, ceil_div(V, 64)), dim3(512), 0, s, reinterpret_cast<const f3*>(verts),
                       unc, N, V);
    return check_launch("hps_vertex_uncertainty");
}


// One-sweep form for 128 < N <= 1024 samples (BASELINE configs[4]: N = 1000): a workgroup is VW = 16 vertices x G = 32 sample
// groups (512 lanes, two workgroups per CU so that one's arithmetic runs under the other's loads); lane (v, g) keeps rows
// s = i G + g, i < SPT <= 32 (96 registers), in registers: the image's samples are read from HBM ONCE -- the two-sweep kernel
// above reads them twice (measured 2.0 x the algorithmic traffic, 0.31 of the HBM roofline at N = 1000).
// Addressing: row i of a lane is (uniform base of row block i) + one per-lane 32-bit offset, i.e. the scalar-base form of
// global_load -- per-lane 64-bit addresses for every load would not fit beside the data registers.  A row block that would reach
// beyond row N - 1 is shifted back to end there (uniformly), and the rows it then repeats are masked by ONE compare of the
// lane's group against a uniform threshold: every address is in bounds, no load is guarded (a guarded load is a basic block of
// its own ending in s_waitcnt vmcnt(0)).
// A sample row of a workgroup is VW * 12 contiguous bytes; the neighbouring vertex chunks, which touch the other parts of the
// same 128-byte lines, are mapped onto the SAME XCD (blockIdx -> chunk below) so that the shared lines are L2 hits.
// Summation order (fixed, independent of B and of the launch geometry): per lane i ascending; then the 64 / VW groups of a
// wave by xor butterflies (commutative: every lane holds the same bits); then the waves in wave order.
// Register budget: two 512-lane workgroups per CU leave 128 registers per lane; 32 rows are 96 of them and hipcc needs ~34
// beside the data, so with SPT > 24 the first LR = SPT - 23 row blocks of a lane do not pass through registers at all: they are
// DMA'd straight into LDS (global_load_lds_dwordx3: 12 bytes per lane, which the hardware places at a 16-BYTE lane pitch --
// tools/ldsdma_probe.hip; 1 KiB per wave instruction, 72 KiB per workgroup at LR = 9, so two workgroups still share the
// 160 KiB) and read back by the lane that requested them -- no barrier, only the wave's own vmcnt.
template <int VW, int G, int SPT>
__global__ __launch_bounds__(VW * G, (VW * G / 256) * (VW * G <= 512 ? 2 : 1)) void uncertainty_sweep1_kernel(
    const f3* __restrict__ verts, float* __restrict__ unc, int N, int V, int chunks_per_xcd) {
    constexpr int GW = 64 / VW, NW = VW * G / 64;              // groups per wave, waves
    constexpr int LR = SPT > 24 ? SPT - 23 : 0;                // row blocks parked in LDS
    constexpr int RR = SPT - LR;                               // row blocks in registers
    __shared__ float sRed[NW * 3 * VW];
    __shared__ __attribute__((aligned(16))) float sRows[LR > 0 ? LR * NW * 64 * 4 : 4];     // [row block][wave][lane][xyz_]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = lane & (VW - 1);
    int g = wave * GW + lane / VW;
    const int chunk = (blockIdx.x & 7) * chunks_per_xcd + (blockIdx.x >> 3);     // blocks of one XCD own adjacent vertex chunks
    const int vg = chunk * VW + v, b = blockIdx.y;
    if (chunk * VW >= V) return;                                                  // padding block of the last XCD (whole workgroup)
    const bool live = vg < V;
    typedef const char __attribute__((address_space(1))) * gbytes;                // GLOBAL pointers (generic ones become flat_load)
    typedef const float __attribute__((address_space(1))) * gfloats;
    auto ld3 = [](gbytes at) { const gfloats q = (gfloats)at; f3 r; r.x = q[0]; r.y = q[1]; r.z = q[2]; return r; };   // one global_load_dwordx3
    const gbytes img = (gbytes)(verts + (size_t)b * N * V);
    const size_t pitch = (size_t)V * 12;
    const unsigned lane_off = ((unsigned)g * (unsigned)V + (unsigned)(live ? vg : V - 1)) * 12u;   // < 32 rows * 82,680 bytes
    f3 p[RR];
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        const int row0 = min(i * G, N - G);                                      // uniform (N > G): scalar registers
        if (i < LR)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + (size_t)row0 * pitch + lane_off),
                                             (__attribute__((address_space(3))) void*)(sRows + (size_t)(i * NW + wave) * 256), 12, 0, 0);
        else
            p[i - LR] = ld3(img + (size_t)row0 * pitch + lane_off);
    }
    // row block i holds this lane's row i G + g iff g >= lo_i, lo_i = (i + 1) G - N (uniform; <= 0 for the full blocks)
    float sx = 0.f, sy = 0.f, sz = 0.f;
    const float* mine = sRows + (size_t)wave * 256 + lane * 4;                    // + i * NW * 256: what this lane's DMAs wrote
#pragma unroll
    for (int i = 0; i < LR; ++i) {                                                // (hipcc waits vmcnt(0) before the first read)
        const bool ok = g >= (i + 1) * G - N;
        const float* q = mine + (size_t)i * NW * 256;
        sx += ok ? q[0] : 0.0f; sy += ok ? q[1] : 0.0f; sz += ok ? q[2] : 0.0f;
    }
#pragma unroll
    for (int i = LR; i < SPT; ++i) {
        const bool ok = g >= (i + 1) * G - N;
        sx += ok ? p[i - LR].x : 0.0f; sy += ok ? p[i - LR].y : 0.0f; sz += ok ? p[i - LR].z : 0.0f;
    }
    asm volatile("" : "+v"(g));                        // the second pass recomputes its masks (32 kept lane masks = 64 SGPRs)
#pragma unroll
    for (int d = VW; d < 64; d <<= 1) { sx += __shfl_xor(sx, d); sy += __shfl_xor(sy, d); sz += __shfl_xor(sz, d); }
    if (lane < VW) { sRed[(wave * 3 + 0) * VW + v] = sx; sRed[(wave * 3 + 1) * VW + v] = sy; sRed[(wave * 3 + 2) * VW + v] = sz; }
    __syncthreads();
    float mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) { mx += sRed[(q * 3 + 0) * VW + v]; my += sRed[(q * 3 + 1) * VW + v]; mz += sRed[(q * 3 + 2) * VW + v]; }
    mx /= N; my /= N; mz /= N;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < LR; ++i) {
        const float* q = mine + (size_t)i * NW * 256;
        const float dist = dist3(q[0] - mx, q[1] - my, q[2] - mz);
        acc += (g >= (i + 1) * G - N) ? dist : 0.0f;
    }
#pragma unroll
    for (int i = LR; i < SPT; ++i) {
        const float dist = dist3(p[i - LR].x - mx, p[i - LR].y - my, p[i - LR].z - mz);
        acc += (g >= (i + 1) * G - N) ? dist : 0.0f;
    }
#pragma unroll
    for (int d = VW; d < 64; d <<= 1) acc += __shfl_xor(acc, d);
    __syncthreads();                                    // every lane has read the partial sums before the buffer is reused
    if (lane < VW) sRed[wave * VW + v] = acc;
    __syncthreads();
    if (wave == 0 && lane < VW && live) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < NW; ++q) t += sRed[q * VW + v];
        unc[(size_t)b * V + vg] = t / N;
    }
}

template <int VW, int G, int SPT>
static int launch_unc_sweep1(const float* verts, float* unc, int B, int N, int V, hipStream_t s) {
    const int chunks = ceil_div(V, VW), cpx = ceil_div(chunks, 8);
    hipLaunchKernelGGL((uncertainty_sweep1_kernel<VW, G, SPT>), dim3(cpx * 8, B), dim3(VW * G), 0, s, reinterpret_cast<const f3*>(verts),
                       unc, N, V, cpx);
    return check_launch("hps_vertex_uncertainty");
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

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_smpl_pose_prep_v1(const float* glob, const float* body, int is_rotmat, const float* betas,
                                         int num_betas, const float* j_template, const float* j_shapedirs,
                                         const int32_t* parents, const int32_t* depth, int num_joints, float* xt, int kp,
                                         int mp, float* a, float* j_posed, float* rot_out, int M, hps_stream_t stream) {
    if (!glob || !body || !betas || !j_template || !j_shapedirs || !parents || !depth || !xt || !a || !j_posed)
        return bad_arg("hps_dev_smpl_pose_prep_v1: null pointer");
    if (num_joints < 1 || num_joints > MAXJ || num_betas < 0 || num_betas > 16) return bad_arg("hps_dev_smpl_pose_prep_v1: dims");
    if (M <= 0) return HPS_OK;
    hipLaunchKernelGGL(pose_prep_kernel_v1, dim3(ceil_div(M, 8)), dim3(256), 0, (hipStream_t)stream, glob, body,
                       is_rotmat, betas, num_betas, j_template, j_shapedirs, parents, depth, num_joints, xt, kp, mp,
                       a, j_posed, rot_out, M);
    return check_launch("hps_dev_smpl_pose_prep_v1");
}
#endif

// LBS launch geometry.  variant 0 = the default chosen for the shipped path; the others exist for tuning
// (hps_dev_lbs_variant).  target_blocks ~ how many workgroups are resident at once on 256 CUs.
template <int K, int G, int VPT>
static int launch_lbs_cfg(const float* v_posed, int ldv, const float* a, const int32_t* w_idx, const float* w_val, int J,
                          const float* transl, float* verts, int M, int V, int target_blocks, hipStream_t s) {
    const int n_vtiles = ceil_div(V, 256 * VPT);
    int n_chunks = max(1, target_blocks / n_vtiles);
    int mpb = ceil_div(ceil_div(M, n_chunks), G) * G;          // meshes per workgroup, a multiple of G
    n_chunks = ceil_div(M, mpb);
    const int chunks_padded = ceil_div(n_chunks, 8) * 8;        // chunk = (local / n_vtiles) * 8 + xcd
    const size_t lds = (size_t)2 * G * J * 12 * sizeof(float);
    hipLaunchKernelGGL((lbs_kernel<K, G, VPT>), dim3(chunks_padded * n_vtiles), dim3(256), lds, s, v_posed, ldv, a, w_idx,
                       w_val, J, transl, reinterpret_cast<f3*>(verts), M, V, mpb, n_vtiles);
    return check_launch("hps_smpl_lbs");
}

template <int K>
static int launch_lbs(const float* v_posed, int ldv, const float* a, const int32_t* w_idx, const float* w_val, int J,
                      const float* transl, float* verts, int M, int V, int variant, int target_blocks, hipStream_t s) {
    if (target_blocks <= 0) target_blocks = 1536;
    switch (variant) {
        case 1: return launch_lbs_cfg<K, 8, 1>(v_posed, ldv, a, w_idx, w_val, J, transl, verts, M, V, target_blocks, s);
#ifdef HPS_DEV_BUILD
        case 0: return launch_lbs_cfg<K, 4, 1>(v_posed, ldv, a, w_idx, w_val, J, transl, verts, M, V, target_blocks, s);
        case 2: return launch_lbs_cfg<K, 4, 2>(v_posed, ldv, a, w_idx, w_val, J, transl, verts, M, V, target_blocks, s);
        case 3: return launch_lbs_cfg<K, 2, 2>(v_posed, ldv, a, w_idx, w_val, J, transl, verts, M, V, target_blocks, s);
        case 4: return launch_lbs_cfg<K, 2, 1>(v_posed, ldv, a, w_idx, w_val, J, transl, verts, M, V, target_blocks, s);
        case 5: return launch_lbs_cfg<K, 8, 2>(v_posed, ldv, a, w_idx, w_val, J, transl, verts, M, V, target_blocks, s);
        case 6: return launch_lbs_cfg<K, 16, 1>(v_posed, ldv, a, w_idx, w_val, J, transl, verts, M, V, target_blocks, s);
#endif
        default: set_error("hps_smpl_lbs: unknown variant %d", variant); return HPS_E_BADARG;
    }
}

static int lbs_dispatch(const float* v_posed, int ldv, const float* a, const int32_t* w_idx, const float* w_val, int K,
                        int num_joints, const float* transl, float* verts, int M, int V, int variant, int target_blocks,
                        hipStream_t s) {
    if (!v_posed || !a || !w_idx || !w_val || !verts) return bad_arg("hps_smpl_lbs: null pointer");
    if (num_joints < 1 || num_joints > 64) return bad_arg("hps_smpl_lbs: num_joints");
    if (ldv < 3 * V) return bad_arg("hps_smpl_lbs: ld_vposed < 3 V");
    if (M <= 0 || V <= 0) return HPS_OK;
    switch (K) {
        case 4: return launch_lbs<4>(v_posed, ldv, a, w_idx, w_val, num_joints, transl, verts, M, V, variant, target_blocks, s);
        case 8: return launch_lbs_cfg<8, 4, 1>(v_posed, ldv, a, w_idx, w_val, num_joints, transl, verts, M, V, 1024, s);
        case 12: return launch_lbs_cfg<12, 2, 1>(v_posed, ldv, a, w_idx, w_val, num_joints, transl, verts, M, V, 768, s);
        case 24: return launch_lbs_cfg<24, 2, 1>(v_posed, ldv, a, w_idx, w_val, num_joints, transl, verts, M, V, 512, s);
        default: set_error("hps_smpl_lbs: K=%d unsupported (4, 8, 12, 24)", K); return HPS_E_UNSUPPORTED;
    }
}

extern "C" int hps_smpl_lbs(const float* v_posed, int ld_vposed, const float* a, const int32_t* w_idx, const float* w_val,
                            int K, int num_joints, const float* transl, float* verts, int M, int V, hps_stream_t stream) {
    // measured on MI355X at 6528 meshes (tests/dev/gpu_bringup.py lbs_tune): many small workgroups of
    // 256 vertices x 8 meshes beat a resident persistent grid: 194 us (5.6 TB/s algorithmic) vs 227-280 us
    return lbs_dispatch(v_posed, ld_vposed, a, w_idx, w_val, K, num_joints, transl, verts, M, V, 1, 24576, (hipStream_t)stream);
}

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_lbs_variant(const float* v_posed, int ld_vposed, const float* a, const int32_t* w_idx,
                                   const float* w_val, int K, int num_joints, const float* transl, float* verts, int M,
                                   int V, int variant, int target_blocks, hps_stream_t stream) {
    return lbs_dispatch(v_posed, ld_vposed, a, w_idx, w_val, K, num_joints, transl, verts, M, V, variant, target_blocks,
                        (hipStream_t)stream);
}
#endif

extern "C" int hps_smpl_joints(const float* verts, const float* j_posed, const int32_t* csr_ptr,
                               const int32_t* csr_col, const float* csr_val, int n_rows, int num_joints,
                               const float* transl, float* joints, int M, int V, hps_stream_t stream) {
    if (!verts || !j_posed || !csr_ptr || !csr_col || !csr_val || !joints) return bad_arg("hps_smpl_joints: null pointer");
    if (M <= 0) return HPS_OK;
    if (num_joints < 0 || n_rows < 0) return bad_arg("hps_smpl_joints: num_joints / n_rows");
    hipLaunchKernelGGL(joints_kernel, dim3(M), dim3(128), 0, (hipStream_t)stream, verts, j_posed, csr_ptr, csr_col,
                       csr_val, n_rows, num_joints, transl, joints, V);
    return check_launch("hps_smpl_joints");
}

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_smpl_joints_v1(const float* verts, const float* j_posed, const int32_t* csr_ptr,
                                      const int32_t* csr_col, const float* csr_val, int n_rows, int num_joints,
                                      const float* transl, float* joints, int M, int V, hps_stream_t stream) {
    if (!verts || !j_posed || !csr_ptr || !csr_col || !csr_val || !joints) return bad_arg("hps_dev_smpl_joints_v1: null pointer");
    if (M <= 0) return HPS_OK;
    hipLaunchKernelGGL(joints_kernel_v1, dim3(M), dim3(128), 0, (hipStream_t)stream, verts, j_posed, csr_ptr, csr_col,
                       csr_val, n_rows, num_joints, transl, joints, V);
    return check_launch("hps_dev_smpl_joints_v1");
}
#endif

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_unc_mode(int mode) {
    g_unc_mode = mode;
    return HPS_OK;
}
#endif

extern "C" int hps_joints_and_uncertainty(const float* picked, const float* j_posed, const int32_t* csr_ptr, const int32_t* csr_slot,
                                          const float* csr_val, int n_rows, int num_joints, const float* transl, float* joints, int M,
                                          int n_picked, const float* verts_samples, float* unc, int B, int N, int V,
                                          hps_stream_t stream) {
    if (!picked || !j_posed || !csr_ptr || !csr_slot || !csr_val || !joints || !verts_samples || !unc)
        return bad_arg("hps_joints_and_uncertainty: null pointer");
    if (num_joints < 0 || n_rows < 0 || M <= 0 || B <= 0 || V <= 0) return bad_arg("hps_joints_and_uncertainty: sizes");
    if (N < 8 || N > 16 * UG) {
        set_error("hps_joints_and_uncertainty: exists for 8 <= num_samples <= %d (call hps_smpl_joints and hps_vertex_uncertainty)", 16 * UG);
        return HPS_E_UNSUPPORTED;
    }
    JointArgs ja;
    ja.picked = picked; ja.j_posed = j_posed; ja.csr_ptr = csr_ptr; ja.csr_slot = csr_slot; ja.csr_val = csr_val; ja.transl = transl;
    ja.joints = joints; ja.n_rows = n_rows; ja.J = num_joints; ja.M = M; ja.n_picked = n_picked; ja.jrows = 0;
    const int spt = ceil_div(N, UG);
    hipStream_t st = (hipStream_t)stream;
    if (spt <= 2) return launch_unc_joints<2>(verts_samples, unc, B, N, V, ja, st);
    if (spt <= 4) return launch_unc_joints<4>(verts_samples, unc, B, N, V, ja, st);
    if (spt <= 8) return launch_unc_joints<8>(verts_samples, unc, B, N, V, ja, st);
    if (spt <= 13) return launch_unc_joints<13>(verts_samples, unc, B, N, V, ja, st);
    return launch_unc_joints<16>(verts_samples, unc, B, N, V, ja, st);
}

extern "C" int hps_vertex_uncertainty(const float* verts, float* unc, int B, int N, int V, hps_stream_t stream) {
    if (!verts || !unc) return bad_arg("hps_vertex_uncertainty: null pointer");
    if (B <= 0 || N <= 0 || V <= 0) return HPS_OK;
#ifdef HPS_DEV_BUILD
    auto lds_bytes = [&](int uv) { return ((size_t)N * 3 * uv + (size_t)(UG / 2) * 3 * uv) * sizeof(float); };
#endif
    // N <= 128 samples: register-resident single pass (mode 4 forces it, modes 1-3 select the older kernels)
    if ((g_unc_mode == 0 || g_unc_mode == 4 || g_unc_mode == 7) && N >= 8 && N <= 16 * UG) {
        const int spt = ceil_div(N, UG);
        hipStream_t st = (hipStream_t)stream;
        if (spt <= 2) return launch_unc_reg<2>(verts, unc, B, N, V, st);
        if (spt <= 4) return launch_unc_reg<4>(verts, unc, B, N, V, st);
        if (spt <= 8) return launch_unc_reg<8>(verts, unc, B, N, V, st);
        if (spt <= 13) return launch_unc_reg<13>(verts, unc, B, N, V, st);
        return launch_unc_reg<16>(verts, unc, B, N, V, st);
    }
    // 128 < N <= 1024: one sweep, 16 vertices x 32 sample groups per workgroup (mode 1: the two-sweep kernel; mode 5: 32 vertices)
    if ((g_unc_mode == 0 || g_unc_mode == 5) && N > 16 * UG && N <= 32 * 32) {
        hipStream_t st = (hipStream_t)stream;
#ifdef HPS_DEV_BUILD
        if (g_unc_mode == 5) return launch_unc_sweep1<32, 32, 32>(verts, unc, B, N, V, st);
#endif
        if (N <= 32 * 8) return launch_unc_sweep1<16, 32, 8>(verts, unc, B, N, V, st);
        if (N <= 32 * 16) return launch_unc_sweep1<16, 32, 16>(verts, unc, B, N, V, st);
        if (N <= 32 * 24) return launch_unc_sweep1<16, 32, 24>(verts, unc, B, N, V, st);
        return launch_unc_sweep1<16, 32, 32>(verts, unc, B, N, V, st);
    }
#ifdef HPS_DEV_BUILD
    int uv = 0;
    if (g_unc_mode == 0) uv = (N >= 8 && lds_bytes(128) <= 160 * 1024) ? 128 : 0;
    else if (g_unc_mode == 2) uv = 128;
    else if (g_unc_mode == 3) uv = 64;
    if (uv && (N < 8 || lds_bytes(uv) > 160 * 1024)) uv = 0;
    if (uv == 128) {       // the image's samples of 128 vertices fit in LDS: read HBM once
        if (int rc = grant_lds<&uncertainty_lds_kernel<128>>(160 * 1024, "hps_vertex_uncertainty")) return rc;
        hipLaunchKernelGGL(uncertainty_lds_kernel<128>, dim3(ceil_div(V, 128), B), dim3(1024), lds_bytes(128), (hipStream_t)stream,
                           reinterpret_cast<const f3*>(verts), unc, N, V);
        return check_launch("hps_vertex_uncertainty");
    }
    if (uv == 64) {
        if (int rc = grant_lds<&uncertainty_lds_kernel<64>>(160 * 1024, "hps_vertex_uncertainty")) return rc;
        hipLaunchKernelGGL(uncertainty_lds_kernel<64>, dim3(ceil_div(V, 64), B), dim3(512), lds_bytes(64), (hipStream_t)stream,
                           reinterpret_cast<const f3*>(verts), unc, N, V);
        return check_launch("hps_vertex_uncertainty");
    }
#endif
    hipLaunchKernelGGL(uncertainty_kernel, dim3(ceil_div(V, 256), B), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f3*>(verts), unc, N, V);
    return check_launch("hps_vertex_uncertainty");
}

// Distribution-prediction head: FC trunk and the hierarchical per-joint MLPs
// (models/poseMF_shapeGaussian_net.py:95-162).
// The 23 sequential joints of the reference loop (:121-160) are grouped by kinematic depth; one
// launch evaluates every joint of a level for the whole batch.
#include "hps_common.h"
#include "svd3_gesdd.h"

namespace hps {

__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }

// These layers are tiny (<= 17 MFLOP per launch at B = 64) and sit on the critical path of the 8-level
// kinematic chain, so the kernels are organised for latency, not throughput: the reduction dimension is
// split over many lanes (KS slices), every lane keeps 8 independent weight loads in flight, and the
// partial sums meet in LDS.

constexpr int TB = 8;      // batch rows per workgroup
constexpr int LCOLS = 16;  // output columns per workgroup of linear_kernel
constexpr int LKS = 16;    // K slices of linear_kernel (LCOLS * LKS = 256 threads)

// acc[r] += sum_{k in [k_lo, k_hi)} xs[k][r] * w[k * ldw]   with 8 weight loads in flight
template <int ROWS>
__device__ __forceinline__ void dot_slice(const float* __restrict__ w, size_t ldw, const float* xs, int k_lo, int k_hi,
                                          float (&acc)[ROWS]) {
    int k = k_lo;
    for (; k + 8 <= k_hi; k += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(k + u) * ldw];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4* xr = reinterpret_cast<const float4*>(xs + (k + u) * ROWS);
#pragma unroll
            for (int q = 0; q < ROWS / 4; ++q) {
                const float4 xv = xr[q];
                acc[q * 4 + 0] += xv.x * wv[u]; acc[q * 4 + 1] += xv.y * wv[u];
                acc[q * 4 + 2] += xv.z * wv[u]; acc[q * 4 + 3] += xv.w * wv[u];
            }
        }
    }
    for (; k < k_hi; ++k) {
        const float wv = w[(size_t)k * ldw];
        const float4* xr = reinterpret_cast<const float4*>(xs + k * ROWS);
#pragma unroll
        for (int q = 0; q < ROWS / 4; ++q) {
            const float4 xv = xr[q];
            acc[q * 4 + 0] += xv.x * wv; acc[q * 4 + 1] += xv.y * wv;
            acc[q * 4 + 2] += xv.z * wv; acc[q * 4 + 3] += xv.w * wv;
        }
    }
}

// Where the fused fc_shape | fc_glob | fc_cam layer (:98-107) delivers its 2 nsh + ng + nc outputs besides the fc_embed input
// buffer: the Gaussian's mean and exp(log std) (:100-101), glob (+ init_glob), cam (+ init_cam) -- each contiguous.
struct TrunkSplit {
    float* loc; float* scale; float* glob; float* cam;
    int nsh, ng, nc;
};

// out[b, n] = act(x[b, :] . wt[:, n] + bias[n] + addend[n]);  256 threads = LKS K-slices x LCOLS columns.
// x2: optional second source -- input column k >= K1 comes from x2[b, k - K1] (fc_embed's input cat[feats, shape, glob, cam], :108,
// without materialising the concatenation); split.loc != nullptr: the outputs are also scattered as TrunkSplit says.
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ x2, int ldx2, int K1,
                                                     const float* __restrict__ wt,
                                                     const float* __restrict__ bias, const float* __restrict__ addend,
                                                     float* __restrict__ out, int ldo, int B, int K, int N, int act, TrunkSplit split) {
    // The head is a short latency chain that shares SIMDs with MFMA-bound kernels of other streams in the pipelined loop:
    // its few instructions go first in the SIMD's issue arbitration (they cost the neighbours next to nothing).
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) float smem[];   // xs[K][TB] then red[LKS][TB][LCOLS]
    float* xs = smem;
    float* red = smem + (size_t)((K * TB + 3) & ~3);
    const int b0 = blockIdx.y * TB;
    const int col = threadIdx.x % LCOLS, ks = threadIdx.x / LCOLS;
    const int n = blockIdx.x * LCOLS + col;

    for (int i = threadIdx.x; i < K * TB; i += 256) {
        const int r = i / K, k = i % K;                      // coalesced along k
        xs[k * TB + r] = (b0 + r < B) ? (k < K1 ? x[(size_t)(b0 + r) * ldx + k] : x2[(size_t)(b0 + r) * ldx2 + (k - K1)]) : 0.0f;
    }
    __syncthreads();

    float acc[TB];
#pragma unroll
    for (int r = 0; r < TB; ++r) acc[r] = 0.0f;
    const int kchunk = ceil_div(K, LKS);
    const int k_lo = min(K, ks * kchunk), k_hi = min(K, k_lo + kchunk);
    if (n < N) dot_slice<TB>(wt + n, (size_t)N, xs, k_lo, k_hi, acc);
#pragma unroll
    for (int r = 0; r < TB; ++r) red[(ks * TB + r) * LCOLS + col] = acc[r];
    __syncthreads();
    if (threadIdx.x < TB * LCOLS) {
        const int r = threadIdx.x / LCOLS, c = threadIdx.x % LCOLS;
        const int nn = blockIdx.x * LCOLS + c;
        if (nn < N && b0 + r < B) {
            float v = 0.0f;
#pragma unroll
            for (int q = 0; q < LKS; ++q) v += red[(q * TB + r) * LCOLS + c];
            v += bias[nn] + (addend ? addend[nn] : 0.0f);
            if (act == HPS_ACT_ELU) v = elu1(v);
            else if (act == HPS_ACT_RELU) v = fmaxf(v, 0.0f);
            out[(size_t)(b0 + r) * ldo + nn] = v;
            if (split.loc) {
                const size_t b = (size_t)(b0 + r);
                const int h = split.nsh;                         // shape_params = [mean (nsh) | log std (nsh)]  (:99-100)
                if (nn < h) split.loc[b * h + nn] = v;
                else if (nn < 2 * h) split.scale[b * h + (nn - h)] = expf(v);                      // Normal(loc, exp(log_std)) :101
                else if (nn < 2 * h + split.ng) split.glob[b * split.ng + (nn - 2 * h)] = v;
                else split.cam[b * split.nc + (nn - 2 * h - split.ng)] = v;
            }
        }
    }
}

// One kinematic level: grid = (n_level joints, batch tiles of TBL images).
// proper SVD + mode (models/poseMF_shapeGaussian_net.py:139-152) of one joint of one image from its raw factors.  Contraction is
// off for this function: it is inlined into two kernels (the level kernel with the in-kernel SVD and svd_finish_kernel of the
// host-LAPACK mode), and u_proper / mode feed the descendants' MLPs -- left to the compiler, the two copies fused different
// multiply-adds and the two SVD modes differed in the last bits of every later F although their SVDs are bit-identical.
#pragma clang fp contract(off)
__device__ __forceinline__ float det3_unfused(const float* m) {          // the expressions of det3 / mat3_mul_bt, written here so
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +      // that the pragma applies to them
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}
__device__ __forceinline__ void mat3_mul_bt_unfused(const float* a, const float* b, float* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[i * 3 + j] = a[i * 3 + 0] * b[j * 3 + 0] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
}
__device__ __forceinline__ void proper_svd_store(float* U, const float* S, float* V, size_t o, float* __restrict__ pose_u,
                                                 float* __restrict__ pose_s, float* __restrict__ pose_v, float* u_proper,
                                                 float* s_proper, float* mode) {
#pragma unroll
    for (int e = 0; e < 9; ++e) { pose_u[o * 9 + e] = U[e]; pose_v[o * 9 + e] = V[e]; }
#pragma unroll
    for (int e = 0; e < 3; ++e) pose_s[o * 3 + e] = S[e];
    const float dU = det3_unfused(U), dV = det3_unfused(V);
    U[2] *= dU; U[5] *= dU; U[8] *= dU;
    V[2] *= dV; V[5] *= dV; V[8] *= dV;
    float Mo[9];
    mat3_mul_bt_unfused(U, V, Mo);
#pragma unroll
    for (int e = 0; e < 9; ++e) { u_proper[o * 9 + e] = U[e]; mode[o * 9 + e] = Mo[e]; }
    s_proper[o * 3 + 0] = S[0];
    s_proper[o * 3 + 1] = S[1];
    s_proper[o * 3 + 2] = S[2] * (dU * dV);
}
#pragma clang fp contract(fast)

// DEVSVD: the level's 3x3 SVDs run inside the kernel (svd3_gesdd.h: LAPACK's sgesdd followed step by step, so that the
// singular vectors carry the signs the reference's torch.svd would give them), followed by the proper-SVD fix -- the level
// needs no host round trip.  u_proper / s_proper / mode are read for ancestor joints (earlier levels) and written for this
// level's joints: no element is both read and written by one launch.
// NT threads = (NT / HID) K-slices x HID columns, TBL images per workgroup.  The product uses NT = 256, TBL = 4: a workgroup
// of four waves, ~110 registers and 18 KB of LDS has the footprint of one workgroup of the kernels it runs beside in the
// pipelined loop (fused mesh kernel, stem convolution), so it is placed as soon as one of those retires; the former
// 1024-thread / 60 KB workgroup needed a nearly empty CU and starved behind them (2.2 ms per head instead of 0.5).
// joint_level_body: one joint of one level for the TBL images starting at b0 (the whole workgroup); slot / n_slots: this
// joint's position in the level (only the host-LAPACK mode's f_level uses them).
template <int HID, bool DEVSVD, int NT, int TBL>
__device__ __forceinline__ void joint_level_body(
    float* smem, const int joint, const int slot, const int n_slots, const int b0,
    const float* __restrict__ embed, int embed_dim,
    const int32_t* __restrict__ anc_ptr, const int32_t* __restrict__ anc_idx, const float* const* __restrict__ w1t_ptrs,
    const float* const* __restrict__ b1_ptrs, const float* const* __restrict__ w2_ptrs,
    const float* const* __restrict__ b2_ptrs, float* u_proper, float* s_proper,
    float* mode, float delta_i_weight, float* __restrict__ pose_f, float* __restrict__ f_level,
    float* __restrict__ pose_u, float* __restrict__ pose_s, float* __restrict__ pose_v, int B, int NJ, int svd_flavor,
    int* published = nullptr) {
    constexpr int KS = NT / HID;
    constexpr int PARTS = NT >= 9 * TBL * 8 ? 8 : 4;       // lanes per output-layer dot product
    static_assert(NT % HID == 0 && TBL % 4 == 0 && NT >= 9 * TBL * PARTS && KS * HID >= 9, "joint_level_kernel: shape");
    const int a_lo = anc_ptr[joint], P = anc_ptr[joint + 1] - a_lo;
    const int in_dim = embed_dim + 21 * P;
    float* xs = smem;                                        // [in_dim][TBL]
    float* hs = smem + (size_t)((in_dim * TBL + 3) & ~3);    // [HID][TBL]
    float* red = hs + HID * TBL;                             // [KS][TBL][HID] partial sums

    // gather: cat[embed, U_proper[anc] (9P), S_proper[anc] (3P), mode[anc] (9P)]   (:126-132).  Guard-free and four elements
    // per lane in flight: the ancestor list goes to LDS first, every source address is then plain arithmetic, rows beyond B read
    // row B - 1 and store 0.  (As `if (b < B) { if (k < embed_dim) ... else { idx = anc_idx[..]; v = table[idx] } }` every element
    // was two dependent global loads, each behind s_waitcnt vmcnt(0): ~13 serialised round trips per lane, 9 of the level's 13 us.)
    int* s_anc = reinterpret_cast<int*>(red);                // the partial-sum buffer is free until the hidden layer
    if ((int)threadIdx.x < P) s_anc[threadIdx.x] = anc_idx[a_lo + threadIdx.x];
    __syncthreads();
    auto source = [&](int i, bool& live) -> const float* {
        const int r = i / in_dim, k = i - r * in_dim;
        const int b = min(b0 + r, B - 1);
        live = b0 + r < B && i < in_dim * TBL;
        if (k < embed_dim) return embed + (size_t)b * embed_dim + k;
        int t = k - embed_dim;
        if (t < 9 * P) return u_proper + ((size_t)b * NJ + s_anc[t / 9]) * 9 + t % 9;
        t -= 9 * P;
        if (t < 3 * P) return s_proper + ((size_t)b * NJ + s_anc[t / 3]) * 3 + t % 3;
        t -= 3 * P;
        return mode + ((size_t)b * NJ + s_anc[max(0, min(t / 9, P - 1))]) * 9 + t % 9;
    };
    for (int i0 = threadIdx.x; i0 < in_dim * TBL; i0 += 4 * NT) {
        float v[4];
        bool live[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *source(min(i0 + q * NT, in_dim * TBL - 1), live[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * NT;
            if (i < in_dim * TBL) {
                const int r = i / in_dim, k = i - r * in_dim;
                xs[k * TBL + r] = (b0 + r < B) ? v[q] : 0.0f;
            }
        }
    }
    __syncthreads();

    // hidden layer
    const int n = threadIdx.x % HID, ks = threadIdx.x / HID;
    float acc[TBL];
#pragma unroll
    for (int r = 0; r < TBL; ++r) acc[r] = 0.0f;
    const int kchunk = ceil_div(in_dim, KS);
    const int k_lo = min(in_dim, ks * kchunk), k_hi = min(in_dim, k_lo + kchunk);
    dot_slice<TBL>(w1t_ptrs[joint] + n, (size_t)HID, xs, k_lo, k_hi, acc);
#pragma unroll
    for (int r = 0; r < TBL; ++r) red[(ks * TBL + r) * HID + n] = acc[r];
    __syncthreads();
    for (int i = threadIdx.x; i < HID * TBL; i += NT) {
        const int r = i / HID, c = i % HID;
        float v = b1_ptrs[joint][c];
#pragma unroll
        for (int q = 0; q < KS; ++q) v += red[(q * TBL + r) * HID + c];
        hs[c * TBL + r] = elu1(v);
    }
    __syncthreads();

    // output layer: 9 x TBL dot products of length HID split over PARTS lanes each, + bias + delta_i_weight * I (:134-135)
    if (threadIdx.x < 9 * TBL * PARTS) {
        const int part = threadIdx.x % PARTS, o = threadIdx.x / PARTS;
        const int e = o / TBL, r = o % TBL;
        const float* w2 = w2_ptrs[joint] + (size_t)e * HID;
        float v = 0.0f;
        for (int k = part; k < HID; k += PARTS) v += w2[k] * hs[k * TBL + r];
#pragma unroll
        for (int m = 1; m < PARTS; m <<= 1) v += __shfl_xor(v, m);
        if (part == 0) {
            v += b2_ptrs[joint][e];
            if (e % 4 == 0) v += delta_i_weight;
            if (b0 + r < B) {
                pose_f[((size_t)(b0 + r) * NJ + joint) * 9 + e] = v;
                if (f_level) f_level[((size_t)(b0 + r) * n_slots + slot) * 9 + e] = v;
            }
            if (DEVSVD) red[r * 9 + e] = v;                 // the partial-sum buffer is free by now
        }
    }
    if (DEVSVD) {
        // One matrix per WAVE (lane 0 of wave r takes image r), not per lane: the LAPACK sequence is all data-dependent branches
        // (sweep direction, zero / non-zero shift, 2 x 2 blocks, sweep counts), and lanes of one wave that take different paths
        // run them one after the other -- 64 different matrices on the lanes of a wave take 45 us, identical ones 9 us
        // (tests/dev/svd_time.py).  A wave of its own follows one path: the level kernel went from 57 to under 30 us.
        static_assert(NT / 64 >= TBL, "one wave per image of the tile");
        __syncthreads();
        const int r = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0 && r < TBL && b0 + r < B) {
            float F[9], U[9], S[3], V[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) F[e] = red[r * 9 + e];
            gesdd3::svd3(svd_flavor, F, U, S, V);
            proper_svd_store(U, S, V, (size_t)(b0 + r) * NJ + joint, pose_u, pose_s, pose_v, u_proper, s_proper, mode);
            // joint_levels_fused_kernel: this lane's stores are what the next level's workgroups wait for -- released by the lane itself
            if (published) __hip_atomic_fetch_add(published, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int HID, bool DEVSVD, int NT, int TBL>
__global__ __launch_bounds__(NT) void joint_level_kernel(
    const float* __restrict__ embed, int embed_dim, const int32_t* __restrict__ joint_ids,
    const int32_t* __restrict__ anc_ptr, const int32_t* __restrict__ anc_idx, const float* const* __restrict__ w1t_ptrs,
    const float* const* __restrict__ b1_ptrs, const float* const* __restrict__ w2_ptrs,
    const float* const* __restrict__ b2_ptrs, float* u_proper, float* s_proper,
    float* mode, float delta_i_weight, float* __restrict__ pose_f, float* __restrict__ f_level,
    float* __restrict__ pose_u, float* __restrict__ pose_s, float* __restrict__ pose_v, int B, int NJ, int svd_flavor) {
    __builtin_amdgcn_s_setprio(3);                          // see linear_kernel
    extern __shared__ __attribute__((aligned(16))) float smem[];
    joint_level_body<HID, DEVSVD, NT, TBL>(smem, joint_ids[blockIdx.x], (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y * TBL, embed,
                                           embed_dim, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs, u_proper, s_proper, mode,
                                           delta_i_weight, pose_f, f_level, pose_u, pose_s, pose_v, B, NJ, svd_flavor);
}

#ifdef HPS_DEV_BUILD
// EXPERIMENT, dev library only (measured and NOT adopted): ALL kinematic levels in ONE launch.  The hypothesis
// was that one image per call spends 8 x ~4.5 us of dispatch floor on the level launches and runs the LAPACK-faithful SVD -- a
// long, branchy routine executed by one lane per matrix -- from a cold instruction cache in every one of them.  Measured on one
// box, interleaved (tools/latency_b1.py --latency [--per-level], profiles/r05_experiments.txt): batch-1 infer() 0.716-0.718 ms
// with this kernel (202 us for the eight levels) against 0.708-0.709 ms with eight launches (8 x 22.6 us) -- stream-ordered
// launches already start back to back (the next dispatch is prepared while its predecessor runs), so there is no floor to
// remove, the SVD's ~15 us per level is a genuine dependent-instruction chain (warm or cold), and a cross-workgroup hand-over
// through L2 costs about what a dispatch does.  Kept as the bit-level cross-check of "images are independent through the head".  grid = (widest level, batch tiles): workgroup (slot, tile) evaluates, level after
// level, the slot-th joint of the level for its tile's images with the very code of joint_level_kernel (same bits), and the
// workgroups of a tile meet between levels at a counter in global memory: level l + 1 starts when all joints of level l have
// published their U_proper / S_proper / mode (release: the lane that ran a matrix's SVD and stored its results adds 1 to the
// level's counter with release semantics; acquire: one lane polls the counter, then every wave executes an acquire fence, which
// also drops the CU's L1 lines -- a line that holds a level-l entry may have been cached with an earlier level's neighbour).
// Images are independent through the head, so no grid-wide synchronisation is needed -- only the <= 5 workgroups of a tile wait
// for each other.  The counters reset themselves: the last workgroup of a tile to leave zeroes them, so the workspace is zeroed
// once when it is allocated.  A waiting workgroup only needs its tile's other workgroups to be scheduled eventually; workgroups
// are dispatched in order (slot fastest), so a tile is never split across "resident" and "never started" for long -- the host
// side still uses this form only for grids that fit the chip at once (hps_dev_head_pose_levels_fused).
struct LevelTable {
    int n_levels;
    int first[HPS_HEAD_MAX_LEVELS];     // index of the level's first joint in level_joints
    int size[HPS_HEAD_MAX_LEVELS];
};
template <int HID, int NT, int TBL>
__global__ __launch_bounds__(NT) void joint_levels_fused_kernel(
    const float* __restrict__ embed, int embed_dim, const int32_t* __restrict__ level_joints, const LevelTable lv,
    const int32_t* __restrict__ anc_ptr, const int32_t* __restrict__ anc_idx, const float* const* __restrict__ w1t_ptrs,
    const float* const* __restrict__ b1_ptrs, const float* const* __restrict__ w2_ptrs,
    const float* const* __restrict__ b2_ptrs, float* u_proper, float* s_proper,
    float* mode, float delta_i_weight, float* __restrict__ pose_f,
    float* __restrict__ pose_u, float* __restrict__ pose_s, float* __restrict__ pose_v, int B, int NJ, int svd_flavor, int* sync) {
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int slot = blockIdx.x, tile = blockIdx.y;
    int* cnt = sync + (size_t)tile * (HPS_HEAD_MAX_LEVELS + 1);
    const int n_live = min(TBL, B - tile * TBL);            // images of this tile: one published (joint, image) result each
    for (int l = 0; l < lv.n_levels; ++l) {
        if (l > 0) {
            if (threadIdx.x == 0) {
                while (__hip_atomic_load(cnt + (l - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lv.size[l - 1] * n_live) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();                                          // (also: the previous level's body is done with the LDS buffers)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // per wave: drop the CU's L1 lines, see the other workgroups' results
        }
        if (slot < lv.size[l])
            joint_level_body<HID, true, NT, TBL>(smem, level_joints[lv.first[l] + slot], slot, lv.size[l], tile * TBL, embed, embed_dim,
                                                 anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs, u_proper, s_proper, mode,
                                                 delta_i_weight, pose_f, nullptr, pose_u, pose_s, pose_v, B, NJ, svd_flavor, cnt + l);
    }
    // the last workgroup of the tile to get here resets the tile's counters (nobody waits on them any more: a workgroup arrives
    // here only after its last wait)
    __syncthreads();
    if (threadIdx.x == 0) {
        const int arrived = __hip_atomic_fetch_add(cnt + HPS_HEAD_MAX_LEVELS, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (int)gridDim.x - 1) {
            for (int l = 0; l <= HPS_HEAD_MAX_LEVELS; ++l) __hip_atomic_store(cnt + l, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#endif

// proper SVD + mode (:139-152): thread per (image, joint of the level); input packed [U | S | V] per matrix
__global__ void svd_finish_kernel(const float* __restrict__ usv, const int32_t* __restrict__ joint_ids, int n_level,
                                  float* __restrict__ pose_u, float* __restrict__ pose_s, float* __restrict__ pose_v,
                                  float* __restrict__ u_proper, float* __restrict__ s_proper, float* __restrict__ mode,
                                  int B, int NJ) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n_level) return;
    const int b = i / n_level, joint = joint_ids[i % n_level];
    const size_t o = (size_t)b * NJ + joint;
    const float* src = usv + (size_t)i * 21;
    float U[9], V[9], S[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) { U[e] = src[e]; V[e] = src[12 + e]; }
#pragma unroll
    for (int e = 0; e < 3; ++e) S[e] = src[9 + e];
    proper_svd_store(U, S, V, o, pose_u, pose_s, pose_v, u_proper, s_proper, mode);
}

// n row-major 3x3 matrices -> packed [U | S | V] (21 floats each), the layout of hps_host_svd3_packed
__global__ void svd3_packed_kernel(const float* __restrict__ f, float* __restrict__ usv, int n, int svd_flavor) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float F[9], U[9], S[3], V[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) F[e] = f[(size_t)i * 9 + e];
    gesdd3::svd3(svd_flavor, F, U, S, V);
    float* o = usv + (size_t)i * 21;
#pragma unroll
    for (int e = 0; e < 9; ++e) { o[e] = U[e]; o[12 + e] = V[e]; }
    o[9] = S[0]; o[10] = S[1]; o[11] = S[2];
}

}  // namespace hps

using namespace hps;

static int launch_linear(const char* who, const float* x, int ldx, const float* x2, int ldx2, int K1, const float* wt, const float* bias,
                         const float* addend, float* out, int ldo, int B, int K, int N, int act, TrunkSplit split, hps_stream_t stream) {
    size_t lds = ((size_t)((K * TB + 3) & ~3) + LKS * TB * LCOLS) * sizeof(float);
    if (lds > 64 * 1024) { set_error("%s: K=%d too large for the LDS tile", who, K); return HPS_E_UNSUPPORTED; }
    hipLaunchKernelGGL(linear_kernel, dim3(ceil_div(N, LCOLS), ceil_div(B, TB)), dim3(256), lds, (hipStream_t)stream, x,
                       ldx, x2, ldx2, K1, wt, bias, addend, out, ldo, B, K, N, act, split);
    return check_launch(who);
}

extern "C" int hps_linear(const float* x, int ldx, const float* wt, const float* bias, const float* addend, float* out,
                          int ldo, int B, int K, int N, int act, hps_stream_t stream) {
    if (!x || !wt || !bias || !out) return bad_arg("hps_linear: null pointer");
    if (K <= 0 || N <= 0 || ldx < K || ldo < N) return bad_arg("hps_linear: dims");
    if (B <= 0) return HPS_OK;
    return launch_linear("hps_linear", x, ldx, nullptr, 0, K, wt, bias, addend, out, ldo, B, K, N, act, TrunkSplit{}, stream);
}

extern "C" int hps_head_trunk(const float* feats, int ldf, const float* fc1_wt, const float* fc1_b, const float* sgc_wt,
                              const float* sgc_b, const float* sgc_add, const float* embed_wt, const float* embed_b, float* x_ws,
                              float* sgc_out, float* embed, float* shape_loc, float* shape_scale, float* glob, float* cam, int B,
                              int num_feats, int hidden, int num_shape, int num_glob, int num_cam, int embed_dim,
                              hps_stream_t stream) {
    if (!feats || !fc1_wt || !fc1_b || !sgc_wt || !sgc_b || !embed_wt || !embed_b || !x_ws || !sgc_out || !embed || !shape_loc ||
        !shape_scale || !glob || !cam)
        return bad_arg("hps_head_trunk: null pointer");
    if (num_feats <= 0 || hidden <= 0 || num_shape <= 0 || num_glob <= 0 || num_cam <= 0 || embed_dim <= 0 || ldf < num_feats)
        return bad_arg("hps_head_trunk: dims");
    if (B <= 0) return HPS_OK;
    const int nt = 2 * num_shape + num_glob + num_cam;
    // x = ELU(fc1(feats))                                                                              (:95-96)
    int rc = launch_linear("hps_head_trunk", feats, ldf, nullptr, 0, num_feats, fc1_wt, fc1_b, nullptr, x_ws, hidden, B, num_feats,
                           hidden, HPS_ACT_ELU, TrunkSplit{}, stream);
    if (rc != HPS_OK) return rc;
    // [shape_params | glob + init_glob | cam + init_cam] = fc_shape | fc_glob | fc_cam (x)             (:98-107)
    TrunkSplit sp{shape_loc, shape_scale, glob, cam, num_shape, num_glob, num_cam};
    rc = launch_linear("hps_head_trunk", x_ws, hidden, nullptr, 0, hidden, sgc_wt, sgc_b, sgc_add, sgc_out, nt, B, hidden, nt,
                       HPS_ACT_NONE, sp, stream);
    if (rc != HPS_OK) return rc;
    // embed = ELU(fc_embed(cat[feats, shape_params, glob, cam]))                                       (:108-110)
    return launch_linear("hps_head_trunk", feats, ldf, sgc_out, nt, num_feats, embed_wt, embed_b, nullptr, embed, embed_dim, B,
                         num_feats + nt, embed_dim, HPS_ACT_ELU, TrunkSplit{}, stream);
}

static int joint_level_launch(const float* embed, int embed_dim, int hidden, const int32_t* joint_ids, int n_level,
                              const int32_t* anc_ptr, const int32_t* anc_idx, const float* const* w1t_ptrs,
                              const float* const* b1_ptrs, const float* const* w2_ptrs, const float* const* b2_ptrs,
                              float* u_proper, float* s_proper, float* mode, float delta_i_weight, float* pose_f,
                              float* f_level, float* pose_u, float* pose_s, float* pose_v, int B, int num_body_joints,
                              int svd_flavor, hps_stream_t stream) {
    const bool devsvd = pose_u != nullptr;
    const bool wide = devsvd && (svd_flavor & HPS_HEAD_WIDE_WORKGROUPS) != 0;
    svd_flavor &= ~HPS_HEAD_WIDE_WORKGROUPS;
    if (devsvd && svd_flavor != HPS_SVD_ROUNDING_REFERENCE && svd_flavor != HPS_SVD_ROUNDING_FMA)
        return bad_arg("hps_head_joint_level_svd: svd_flavor");
    if (!embed || !joint_ids || !anc_ptr || !anc_idx || !w1t_ptrs || !b1_ptrs || !w2_ptrs || !b2_ptrs || !u_proper ||
        !s_proper || !mode || !pose_f || (devsvd && (!pose_s || !pose_v)))
        return bad_arg("hps_head_joint_level: null pointer");
    if (hidden != 128) { set_error("hps_head_joint_level: hidden=%d unsupported (128 = EMBED_DIM/2)", hidden); return HPS_E_UNSUPPORTED; }
    if (B <= 0 || n_level <= 0) return HPS_OK;
    constexpr int NT = 256, TBL = 4;
    const int max_in = embed_dim + 21 * num_body_joints;
    size_t lds = ((size_t)((max_in * TBL + 3) & ~3) + 128 * TBL + (NT / 128) * TBL * 128) * sizeof(float);
    if (lds > 64 * 1024) { set_error("hps_head_joint_level: embed_dim=%d too large", embed_dim); return HPS_E_UNSUPPORTED; }
    constexpr int NTW = 1024;
    const size_t ldsw = ((size_t)((max_in * TBL + 3) & ~3) + 128 * TBL + (NTW / 128) * TBL * 128) * sizeof(float);
    // the wide form needs 12 KiB more than the default; an embed_dim whose default form still fits but whose wide form does
    // not takes the default (same results up to the K-slice summation order the latency mode already implies)
    if (wide && ldsw <= 64 * 1024) {
        hipLaunchKernelGGL((joint_level_kernel<128, true, NTW, TBL>), dim3(n_level, ceil_div(B, TBL)), dim3(NTW), ldsw, (hipStream_t)stream,
                           embed, embed_dim, joint_ids, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs, u_proper,
                           s_proper, mode, delta_i_weight, pose_f, f_level, pose_u, pose_s, pose_v, B, num_body_joints, svd_flavor);
    } else if (devsvd)
        hipLaunchKernelGGL((joint_level_kernel<128, true, NT, TBL>), dim3(n_level, ceil_div(B, TBL)), dim3(NT), lds, (hipStream_t)stream,
                           embed, embed_dim, joint_ids, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs, u_proper,
                           s_proper, mode, delta_i_weight, pose_f, f_level, pose_u, pose_s, pose_v, B, num_body_joints, svd_flavor);
    else
        hipLaunchKernelGGL((joint_level_kernel<128, false, NT, TBL>), dim3(n_level, ceil_div(B, TBL)), dim3(NT), lds, (hipStream_t)stream,
                           embed, embed_dim, joint_ids, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs, u_proper,
                           s_proper, mode, delta_i_weight, pose_f, f_level, pose_u, pose_s, pose_v, B, num_body_joints, svd_flavor);
    return check_launch("hps_head_joint_level");
}

extern "C" int hps_head_joint_level(const float* embed, int embed_dim, int hidden, const int32_t* joint_ids,
                                    int n_level, const int32_t* anc_ptr, const int32_t* anc_idx,
                                    const float* const* w1t_ptrs, const float* const* b1_ptrs,
                                    const float* const* w2_ptrs, const float* const* b2_ptrs, const float* u_proper,
                                    const float* s_proper, const float* mode, float delta_i_weight, float* pose_f,
                                    float* f_level, int B, int num_body_joints, hps_stream_t stream) {
    return joint_level_launch(embed, embed_dim, hidden, joint_ids, n_level, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs,
                              const_cast<float*>(u_proper), const_cast<float*>(s_proper), const_cast<float*>(mode),
                              delta_i_weight, pose_f, f_level, nullptr, nullptr, nullptr, B, num_body_joints, 0, stream);
}

extern "C" int hps_head_joint_level_svd(const float* embed, int embed_dim, int hidden, const int32_t* joint_ids,
                                        int n_level, const int32_t* anc_ptr, const int32_t* anc_idx,
                                        const float* const* w1t_ptrs, const float* const* b1_ptrs,
                                        const float* const* w2_ptrs, const float* const* b2_ptrs, float* u_proper,
                                        float* s_proper, float* mode, float delta_i_weight, float* pose_f, float* pose_u,
                                        float* pose_s, float* pose_v, int B, int num_body_joints, int svd_flavor,
                                        hps_stream_t stream) {
    if (!pose_u) return bad_arg("hps_head_joint_level_svd: null pointer");
    return joint_level_launch(embed, embed_dim, hidden, joint_ids, n_level, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs,
                              u_proper, s_proper, mode, delta_i_weight, pose_f, nullptr, pose_u, pose_s, pose_v, B,
                              num_body_joints, svd_flavor, stream);
}

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_head_pose_levels_fused(const float* embed, int embed_dim, int hidden, const int32_t* level_joints,
                                          const int32_t* level_sizes_host, int n_levels, const int32_t* anc_ptr,
                                          const int32_t* anc_idx, const float* const* w1t_ptrs, const float* const* b1_ptrs,
                                          const float* const* w2_ptrs, const float* const* b2_ptrs, float* u_proper,
                                          float* s_proper, float* mode, float delta_i_weight, float* pose_f, float* pose_u,
                                          float* pose_s, float* pose_v, int B, int num_body_joints, int svd_flavor,
                                          int32_t* sync_ws, hps_stream_t stream) {
    const bool wide = (svd_flavor & HPS_HEAD_WIDE_WORKGROUPS) != 0;
    svd_flavor &= ~HPS_HEAD_WIDE_WORKGROUPS;
    if (svd_flavor != HPS_SVD_ROUNDING_REFERENCE && svd_flavor != HPS_SVD_ROUNDING_FMA) return bad_arg("hps_dev_head_pose_levels_fused: svd_flavor");
    if (!embed || !level_joints || !level_sizes_host || !anc_ptr || !anc_idx || !w1t_ptrs || !b1_ptrs || !w2_ptrs || !b2_ptrs ||
        !u_proper || !s_proper || !mode || !pose_f || !pose_u || !pose_s || !pose_v || !sync_ws)
        return bad_arg("hps_dev_head_pose_levels_fused: null pointer");
    if (hidden != 128) { set_error("hps_dev_head_pose_levels_fused: hidden=%d unsupported (128 = EMBED_DIM/2)", hidden); return HPS_E_UNSUPPORTED; }
    if (n_levels < 1 || n_levels > HPS_HEAD_MAX_LEVELS) return bad_arg("hps_dev_head_pose_levels_fused: n_levels");
    if (B <= 0) return HPS_OK;
    LevelTable lv;
    lv.n_levels = n_levels;
    int widest = 0, first = 0;
    for (int l = 0; l < HPS_HEAD_MAX_LEVELS; ++l) {
        lv.first[l] = first;
        lv.size[l] = l < n_levels ? level_sizes_host[l] : 0;
        if (lv.size[l] < 0) return bad_arg("hps_dev_head_pose_levels_fused: level size");
        first += lv.size[l];
        widest = lv.size[l] > widest ? lv.size[l] : widest;
    }
    if (widest == 0) return HPS_OK;
    constexpr int NT = 256, NTW = 1024, TBL = 4;
    const int tiles = ceil_div(B, TBL);
    // the workgroups of a tile wait for each other: only for grids the chip holds at once (two 1024-thread or eight 256-thread
    // workgroups per CU) -- beyond that the caller uses the per-level launches (hps_head_pose_levels), same bits
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    if ((long)widest * tiles > (long)cus) {
        set_error("hps_dev_head_pose_levels_fused: %d x %d workgroups do not fit the device's %d CUs at once; use hps_head_pose_levels", widest, tiles, cus);
        return HPS_E_UNSUPPORTED;
    }
    const int max_in = embed_dim + 21 * num_body_joints;
    const size_t lds = ((size_t)((max_in * TBL + 3) & ~3) + 128 * TBL + (NT / 128) * TBL * 128) * sizeof(float);
    const size_t ldsw = ((size_t)((max_in * TBL + 3) & ~3) + 128 * TBL + (NTW / 128) * TBL * 128) * sizeof(float);
    if (lds > 64 * 1024) { set_error("hps_dev_head_pose_levels_fused: embed_dim=%d too large", embed_dim); return HPS_E_UNSUPPORTED; }
    if (wide && ldsw <= 64 * 1024)
        hipLaunchKernelGGL((joint_levels_fused_kernel<128, NTW, TBL>), dim3(widest, tiles), dim3(NTW), ldsw, (hipStream_t)stream, embed, embed_dim,
                           level_joints, lv, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs, u_proper, s_proper, mode, delta_i_weight,
                           pose_f, pose_u, pose_s, pose_v, B, num_body_joints, svd_flavor, sync_ws);
    else
        hipLaunchKernelGGL((joint_levels_fused_kernel<128, NT, TBL>), dim3(widest, tiles), dim3(NT), lds, (hipStream_t)stream, embed, embed_dim,
                           level_joints, lv, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs, u_proper, s_proper, mode, delta_i_weight,
                           pose_f, pose_u, pose_s, pose_v, B, num_body_joints, svd_flavor, sync_ws);
    return check_launch("hps_dev_head_pose_levels_fused");
}

#endif

extern "C" int hps_svd3_packed(const float* f, float* usv, int n, int svd_flavor, hps_stream_t stream) {
    if (!f || !usv) return bad_arg("hps_svd3_packed: null pointer");
    if (svd_flavor != HPS_SVD_ROUNDING_REFERENCE && svd_flavor != HPS_SVD_ROUNDING_FMA) return bad_arg("hps_svd3_packed: svd_flavor");
    if (n <= 0) return HPS_OK;
    hipLaunchKernelGGL(svd3_packed_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, (hipStream_t)stream, f, usv, n, svd_flavor);
    return check_launch("hps_svd3_packed");
}

extern "C" int hps_head_svd_finish(const float* usv_level, const int32_t* joint_ids, int n_level, float* pose_u,
                                   float* pose_s, float* pose_v, float* u_proper, float* s_proper, float* mode, int B,
                                   int num_body_joints, hps_stream_t stream) {
    if (!usv_level || !pose_u || !pose_s || !pose_v || !joint_ids || !u_proper || !s_proper || !mode)
        return bad_arg("hps_head_svd_finish: null pointer");
    if (B <= 0 || n_level <= 0) return HPS_OK;
    hipLaunchKernelGGL(svd_finish_kernel, dim3(ceil_div(B * n_level, 128)), dim3(128), 0, (hipStream_t)stream, usv_level,
                       joint_ids, n_level, pose_u, pose_s, pose_v, u_proper, s_proper, mode, B, num_body_joints);
    return check_launch("hps_head_svd_finish");
}

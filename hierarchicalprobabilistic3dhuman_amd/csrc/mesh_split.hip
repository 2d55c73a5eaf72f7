// Fused mesh kernel, shared-shape form, with the pose blend GEMM on the bf16 matrix pipe at fp32 accuracy ("bf16x3"):
// every fp32 operand x is carried as THREE bf16 pieces x = x1 + x2 + x3 (x1 = RN_bf16(x), x2 = RN_bf16(x - x1), x3 = RN_bf16(x - x1 - x2):
// 3 x 8 significand bits = the 24 of an fp32 number, so the sum is exact for every x with 2^-110 <= |x| < 3.39e38 -- the pieces stay normal
// and finite --, tests/test_host_logic.py), and a product a b is formed as the six
// piece products of weight >= 2^-16 -- a1 b3, a3 b1, a2 b2, a1 b2, a2 b1, a1 b1 -- each of them EXACT in the fp32 accumulator's input
// (8 x 8 bits); what is dropped (a2 b3 + a3 b2 + a3 b3) is below 2^-23 |a b|, the size of the rounding of ONE fp32 multiply-add, and the
// accumulator is rounded six times per 16 k instead of sixteen times (v_mfma_f32_32x32x2_f32 = a chain of fmaf).  Measured against the
// float64 twin the vertices are as close as the fp32-MFMA kernel's (tests/test_gpu_smpl.py::test_split_bf16_*).
//
// Why: on gfx950 v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD -- it shares the fp32 datapath, which is why
// fp32 VALU work never overlaps it: tools/mfma_valu_overlap.hip) while v_mfma_f32_32x32x16_bf16 runs sixteen times faster on its own
// pipe: six bf16 MFMAs per 16 k cost 192 cycles where eight fp32 MFMAs cost 512, and the epilogue's fp32 VALU work of the other
// workgroups on the CU can run beside them.  Opt-in (SMPL.mesh_arith = "bf16x3"): the default path keeps the reference's own
// arithmetic type in every instruction.
//
// Data.  Both operands are pre-split into MFMA fragment order by hps_smpl_split_bf16x3 (the blend matrix once per model, the pose
// features once per call: 5 MB):  dst[col tile][16-row chunk][piece s][k half kl][column w][8 bf16]  -- a (tile, chunk) block is one
// contiguous run (6 KiB per 64 meshes, 18 KiB per 192-column panel), so a K-loop stage is 24 linear 1 KiB LDS-DMA pieces, and a
// fragment is one conflict-free ds_read_b128 per lane (32 consecutive lanes read 512 consecutive bytes).
// Tile, block mapping and the skinning epilogue are mesh_fused_kernel's (csrc/mesh_fused.hip), the VS / PICK form.
//
// Replaces smplx 0.1.26 lbs steps pose_feature @ posedirs, W @ A and T @ v_posed (reached from models/smpl_official.py:29).

#include <type_traits>

#include "hps_common.h"

namespace hps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SV = 64;                         // vertices per panel
constexpr int SN = 3 * SV;                     // blend-matrix columns per panel
constexpr int SBK = 16;                        // K rows per chunk = one bf16 MFMA
constexpr int S_BB = 3 * 2 * SN * 16;          // bytes of a chunk's blend-matrix panel: [3 pieces][2 k halves][192 columns][8 bf16] = 18 432
// MG = mesh groups of 32 per workgroup tile (2 vertex groups of 32 each): the tile is 32 MG meshes x 64 vertices, 2 MG waves.
// Operand bytes per chunk = 6 144 per 64 meshes + 18 432: the L2 -> LDS stream per unit of work falls with MG (24 / 15 KiB per 64 x 64 at
// MG = 2 / 4), and at the bf16 rate that stream, not the MFMA pipe, is what the K loop waits for (MG = 2: 3.5 GB per 6 528 meshes).
template <int MG> struct SplitCfg {
    static constexpr int SM = 32 * MG;                 // meshes per tile
    static constexpr int THREADS = 128 * MG;
    static constexpr int WAVES = 2 * MG;
    static constexpr int XB = 3 * 2 * SM * 16;         // bytes of a chunk's mesh operand: [3 pieces][2 k halves][SM meshes][8 bf16]
    static constexpr int STAGE = XB + S_BB;
    static constexpr int NPIECES = STAGE / 1024;       // 1 KiB DMA pieces per stage
    static constexpr int PER_WAVE = (NPIECES + WAVES - 1) / WAVES;
    static constexpr int A_BYTES = 16 * MG * 24 * 12 * 4;      // skinning transforms of 16 meshes per mesh group (one epilogue pass)
    static constexpr int LDS = 2 * STAGE > A_BYTES ? 2 * STAGE : A_BYTES;
    static constexpr int WAVES_PER_SIMD = MG == 2 ? 3 : 4;     // MG 2: 48 KiB -> three workgroups per CU; MG 4: 72 KiB -> two (16 waves)
    static_assert(STAGE % 1024 == 0 && PER_WAVE <= 6 /* (ABL 13: <= 4) */, "whole pieces; at most one per product group of three MFMAs");
};

// RN-even fp32 -> bf16 (the bits torch's .bfloat16() produces); no NaN handling: operands are finite
__device__ __forceinline__ unsigned bf16_rn(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// dst[(tile, chunk)][s][kl][w][j] = piece s of src[(16 chunk + 8 kl + j) * ld + tile * W + w]; rows >= rows are zero.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, int rows, int ld, int cols, int W,
                                                           uint4* __restrict__ dst, int nchunks) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    const int c = blockIdx.y >> 1, kl = blockIdx.y & 1;
    unsigned p[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = c * SBK + kl * 8 + j;
        const float x = k < rows ? src[(size_t)k * ld + col] : 0.0f;
        const unsigned h1 = bf16_rn(x);
        const float r1 = x - __uint_as_float(h1 << 16);               // exact
        const unsigned h2 = bf16_rn(r1);
        const float r2 = r1 - __uint_as_float(h2 << 16);              // exact
        p[0][j] = h1; p[1][j] = h2; p[2][j] = bf16_rn(r2);
    }
    const int tile = col / W, w = col - tile * W;
    uint4* base = dst + ((size_t)(tile * nchunks + c) * 3) * (2 * W) + (size_t)kl * W + w;       // in 16-byte units
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        uint4 o;
        o.x = p[s][0] | (p[s][1] << 16); o.y = p[s][2] | (p[s][3] << 16);
        o.z = p[s][4] | (p[s][5] << 16); o.w = p[s][6] | (p[s][7] << 16);
        base[(size_t)s * 2 * W] = o;
    }
}

// ABL (dev library only): 1 = K loop only (no skinning), 2 = skinning only (no K loop), 3 = K loop without its MFMAs (operand stream only),
// 4 = DMA pieces in a burst, 5 = operands one chunk ahead (the fp32 kernel's scheme); 4 and 5 give valid results; 6 = K loop only WITHOUT the operand stream after the first two chunks
// (MFMAs, fragment reads, one barrier per chunk), 7 = 6 with the second barrier, 8 = 6 without barriers, 9 = 8 on one stage;
// skinning only (like 2): 10 = without the DMA of the transforms, 11 = without the stores, 12 = without the skinning arithmetic (v_posed stored)
template <int K, int JC, int MG, int ABL = 0>
__global__ __launch_bounds__(SplitCfg<MG>::THREADS, SplitCfg<MG>::WAVES_PER_SIMD) void mesh_split_kernel(
    const char* __restrict__ xsplit, const char* __restrict__ bsplit, const float* __restrict__ v_shaped, const float* __restrict__ a,
    const int32_t* __restrict__ w_idx, const float* __restrict__ w_val, f3* __restrict__ verts, int M, int V, int nchunks, int tiles_m,
    int tiles_m_per_xcd, const int32_t* __restrict__ pick_slot, f3* __restrict__ picked, int n_picked,
    const int32_t* __restrict__ mesh_row, const int32_t* __restrict__ group_rows, int stagger) {
    typedef SplitCfg<MG> C;
    constexpr int SM = C::SM, SW = C::WAVES, S_XB = C::XB, S_STAGE = C::STAGE;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // union: two operand stages | A of 16 MG of the tile's meshes

    // block -> (mesh tile, panel): mesh_fused_kernel's mapping (blocks of one XCD own the same mesh tiles and walk the panels in order)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile_m = tiles_m_per_xcd ? (local % tiles_m_per_xcd) * 8 + xcd : (int)blockIdx.x % tiles_m;
    const int panel = tiles_m_per_xcd ? local / tiles_m_per_xcd : (int)blockIdx.x / tiles_m;
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * SM;
    // De-phasing: the workgroups that share a CU start together and, all tiles costing the same, stay in step -- K loops (matrix pipe, L2
    // stream) together, then epilogues (VALU, LDS, stores) together.  The second resident workgroup of every CU (blocks 256 .. 511 of
    // the launch: the dispatcher fills one slot per CU first) starts `stagger` x ~1.5 us late, once; its successors inherit the offset.
    if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(56);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, il = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    const int v = panel * SV + wn * 32 + il;
    const bool live_v = v < V;
    const int vc = live_v ? v : V - 1;
    int idx[K];
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        idx[k] = w_idx[(size_t)vc * K + k] * 12;
        w[k] = w_val[(size_t)vc * K + k];
    }
    const int32_t* gr = group_rows + 3 * __builtin_amdgcn_readfirstlane((m0 + wm * 32) >> 5);      // wave-uniform: scalar loads
    const int split = gr[2];
    const f3 vt = reinterpret_cast<const f3*>(v_shaped)[(size_t)gr[0] * V + vc];
    const f3 vtb = reinterpret_cast<const f3*>(v_shaped)[(size_t)gr[1] * V + vc];
    const int split_lane = split - 4 * kl;                    // local mesh 4 kl + dr < split  <=>  dr < split_lane
    const int pick = live_v ? pick_slot[vc] : -1;

    const unsigned lds0 = (unsigned)(size_t)(lptr_t)(smem);
    const unsigned lane16 = (unsigned)lane * 16u;
    const char* x_src = xsplit + (size_t)tile_m * nchunks * S_XB;
    const char* b_src = bsplit + (size_t)panel * nchunks * S_BB;
    // piece j of the wave's pieces of a stage: stage piece q = wave + WAVES j; the first XB / 1024 pieces are the mesh operand, the rest
    // the blend matrix (both sources are linear runs); with MG = 4 a stage has 30 pieces: waves 6, 7 have three
    auto dma_piece = [&](int j, int buf, const char* xs, const char* bs) {
        const int q = wave + SW * j;
        if (C::NPIECES % SW != 0 && q >= C::NPIECES) return;                                      // wave-uniform
        const char* src = q < S_XB / 1024 ? xs + q * 1024 : bs + (q - S_XB / 1024) * 1024;       // wave-uniform select
        lds_dma16(lane16, reinterpret_cast<const float*>(src), lds0 + (unsigned)(buf * S_STAGE + q * 1024));
    };

    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

#pragma unroll
    for (int j = 0; j < C::PER_WAVE; ++j) dma_piece(j, 0, x_src, b_src);

    const char* const sbytes = reinterpret_cast<const char*>(smem);
    const int a_frag = kl * (SM * 16) + (wm * 32 + il) * 16;                     // + s * 2 SM 16
    const int b_frag = S_XB + kl * (SN * 16) + (wn * 32 + il) * 16;              // + s * 2 SN 16 + t * 64 * 16
    // K loop, operands fetched TWO chunks ahead through two stages: a chunk's fragments are complete in registers before its first MFMA,
    // so its stage is free as soon as every wave has read -- a second barrier behind the reads, and the DMA of chunk c + 2 goes into the
    // stage chunk c was just read from, with the MFMAs of chunks c and c + 1 to land under.  (One chunk ahead -- the fp32 kernel's scheme,
    // ABL = 5 -- left the MFMA pipe idle half the time: at the bf16 rate a chunk's MFMAs are 0.24 us per wave, shorter than a loaded
    // L2 -> LDS round trip.)  YOUNGER: chunk c + 1's pieces are in flight at the top of the iteration; ISSUE: chunk c + 2 exists.
    // The body is instantiated for the three cases: a run-time test inside the MFMA run would be a branch per piece.
    constexpr bool AHEAD2 = ABL != 5;
    const bool full_wave = C::NPIECES % SW == 0 || wave + SW * (C::PER_WAVE - 1) < C::NPIECES;      // this wave has PER_WAVE pieces per chunk
    auto do_chunk = [&](auto younger_c, auto issue_c, int c) __attribute__((always_inline)) {
        constexpr bool nostream = ABL == 6 || ABL == 7 || ABL == 8 || ABL == 9;
        constexpr bool younger = decltype(younger_c)::value && !nostream, issue = decltype(issue_c)::value && !nostream;
        if (AHEAD2 && younger) {                           // chunk c has landed; chunk c + 1 (issued later) may still be in flight
            if (full_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PER_WAVE) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PER_WAVE - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (ABL != 8 && ABL != 9) __syncthreads();         // ... for everyone
        const int ahead = AHEAD2 ? 2 : 1;
        const char* nx = x_src + (size_t)(c + ahead) * S_XB;
        const char* nb = b_src + (size_t)(c + ahead) * S_BB;
        const int nbuf = (c + ahead) & 1;
        const char* stage = sbytes + (ABL == 9 ? 0 : (c & 1)) * S_STAGE;
        bf16x8 af[3], bf[3][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) af[s] = *reinterpret_cast<const bf16x8*>(stage + a_frag + s * (2 * SM * 16));
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                bf[t][s] = *reinterpret_cast<const bf16x8*>(stage + b_frag + s * (2 * SN * 16) + t * (SV * 16));
        constexpr bool LATE = ABL == 13;                   // dev: the stage-free barrier behind the first two product groups instead of in front
        if (AHEAD2 && (issue || ABL == 7) && !LATE) {      // every wave holds its fragments: the stage may be refilled
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (issue && ABL == 4) {                           // dev: the pieces in one burst in front of the MFMAs
#pragma unroll
            for (int j = 0; j < C::PER_WAVE; ++j) dma_piece(j, nbuf, nx, nb);
        }
        __builtin_amdgcn_sched_barrier(0);                 // all fragment reads in flight before the first MFMA
        // piece products smallest first; the three tiles (x, y, z of the wave's 32 vertices) take turns, so that an MFMA never waits
        // for the accumulator of the one before it
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (ABL == 3) { acc[t][p] += (float)af[PA[p]][0] * (float)bf[t][PB[p]][0]; continue; }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[p]], bf[t][PB[p]], acc[t], 0, 0, 0);
            }
            if (LATE && AHEAD2 && issue && p == 1) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (issue && ABL != 4 && (LATE ? (p >= 2 && p - 2 < C::PER_WAVE) : p < C::PER_WAVE)) {    // the pieces go out between the MFMAs, not in a burst
                __builtin_amdgcn_sched_barrier(0);
                dma_piece(LATE ? p - 2 : p, nbuf, nx, nb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (ABL == 2 || (ABL >= 10 && ABL <= 12)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (AHEAD2) {
        if (nchunks > 1) {
#pragma unroll
            for (int j = 0; j < C::PER_WAVE; ++j) dma_piece(j, 1, x_src + S_XB, b_src + S_BB);
        }
        int c = 0;
        for (; c + 2 < nchunks; ++c) do_chunk(std::true_type(), std::true_type(), c);
        if (nchunks > 1) do_chunk(std::true_type(), std::false_type(), c++);
        do_chunk(std::false_type(), std::false_type(), c);
    } else {
        for (int c = 0; c + 1 < nchunks; ++c) do_chunk(std::false_type(), std::true_type(), c);
        do_chunk(std::false_type(), std::false_type(), nchunks - 1);
    }
    if (ABL == 1 || ABL == 3 || (ABL >= 6 && ABL <= 9)) {                            // K loop only: one never-taken store keeps the accumulators alive
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r] + acc[2][r];
        if (t == 12345.678f) verts[0].x = t;
        return;
    }

    // Skinning: mesh_fused_kernel's epilogue (VS, PICK, no translation), two passes: pass p stages A of meshes [32 h + 16 p, + 16) of the
    // tile for every mesh group h -- the meshes of accumulator registers r = 8 p .. 8 p + 7 -- as LDS slots 16 h .. 16 h + 15.
    const int a_stride = JC * 12;
    const int half_bytes = 16 * a_stride * 4;
    const int slot0 = wm * 16 + 4 * kl;
    int aoff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) aoff[k] = slot0 * a_stride + idx[k];
    char* const vbase = reinterpret_cast<char*>(verts) + (size_t)(m0 + wm * 32) * V * 12;      // wave-uniform
    const unsigned voff = ((unsigned)(4 * kl) * (unsigned)V + (unsigned)v) * 12u;              // per lane
    char* const pbase = reinterpret_cast<char*>(picked) + (size_t)(m0 + wm * 32) * n_picked * 12;
    const unsigned poff = ((unsigned)(4 * kl) * (unsigned)n_picked + (unsigned)max(pick, 0)) * 12u;
    auto epilogue = [&](auto many_c) __attribute__((always_inline)) {
        constexpr bool MANY = decltype(many_c)::value;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();                               // operand stages / the previous pass's transforms are dead
#pragma unroll
            for (int h = 0; h < MG; ++h) {
                const int mh = m0 + 32 * h + 16 * pass;
                const int valid = max(0, min(16, M - mh)) * a_stride * 4;
                const float* a_src = a + (size_t)mh * a_stride;
                for (int piece = wave; piece * 1024 < half_bytes; piece += SW) {
                    const int off = piece * 1024 + lane * 16;
                    if (off < valid && ABL != 10) lds_dma16((unsigned)off, a_src, lds0 + (unsigned)(h * half_bytes + piece * 1024));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 8 * pass + q;
                const int dr = (r & 3) + 8 * (r >> 2);
                const int ds = (q & 3) + 8 * (q >> 2);
                const int m = m0 + wm * 32 + 4 * kl + dr;
                f3 base;
                if (!MANY) {
                    const bool first = dr < split_lane;
                    base.x = first ? vt.x : vtb.x; base.y = first ? vt.y : vtb.y; base.z = first ? vt.z : vtb.z;
                } else {
                    base = reinterpret_cast<const f3*>(v_shaped)[(size_t)mesh_row[min(m, M - 1)] * V + vc];
                }
                f3 pv;
                pv.x = base.x + acc[0][r]; pv.y = base.y + acc[1][r]; pv.z = base.z + acc[2][r];
                int ao[K];
#pragma unroll
                for (int k = 0; k < K; ++k) ao[k] = aoff[k] + ds * a_stride;
                const f3 o = ABL == 12 ? pv : skin_vertex<K>(smem, ao, w, pv, 0.f, 0.f, 0.f);
                asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z));      // (see mesh_fused_kernel: keeps the skinning out of the guarded block)
                if (ABL == 11) {                           // dev: no stores (a never-true guard keeps the skinning alive)
                    if (o.x == 12345.678f) *reinterpret_cast<f3*>(vbase + (size_t)dr * V * 12 + voff) = o;
                    continue;
                }
                if (live_v && m < M) *reinterpret_cast<f3*>(vbase + (size_t)dr * V * 12 + voff) = o;
                if (pick >= 0 && m < M) *reinterpret_cast<f3*>(pbase + (size_t)dr * n_picked * 12 + poff) = o;
            }
        }
    };
    if (split < 0) epilogue(std::true_type());
    else epilogue(std::false_type());
}

#ifdef HPS_DEV_BUILD
static int g_split_mg = 0;            // hps_dev_mesh_split_groups: 2 = the 64-mesh tile (A/B)
static int g_split_abl = 0;           // hps_dev_mesh_split_ablate
static int g_split_stagger = -1;      // hps_dev_mesh_split_stagger (< 0: the product's value)
#endif
constexpr int SPLIT_STAGGER = 8;            // measured: 0.419 -> 0.402 ms at 6 528 meshes (2 .. 20 tried; tests/dev/mesh_split_time.py)

template <int MG, int ABL = 0>
static int launch_split(const void* xsplit, const void* bsplit, const float* v_shaped, const int32_t* mesh_row, const int32_t* group_rows,
                        const float* a, const int32_t* w_idx, const float* w_val, float* verts, int M, int V, int rows,
                        const int32_t* pick_slot, float* picked, int n_picked, hipStream_t s) {
    typedef SplitCfg<MG> C;
    const int tiles_m = ceil_div(M, C::SM), n_panels = ceil_div(V, SV);
    const int tiles_m_per_xcd = tiles_m >= 8 ? ceil_div(tiles_m, 8) : 0;
    const dim3 grid(tiles_m_per_xcd ? tiles_m_per_xcd * 8 * n_panels : tiles_m * n_panels);
    // (only launches of several resident rounds: a few hundred workgroups have no second round to keep apart, and the delay would be latency)
    int stagger = grid.x >= 2048 ? SPLIT_STAGGER : 0;
#ifdef HPS_DEV_BUILD
    if (g_split_stagger >= 0) stagger = g_split_stagger;
#endif
    if (int rc = grant_lds<&mesh_split_kernel<4, 24, MG, ABL>>(160 * 1024, "hps_smpl_mesh_fused_shared_shape_bf16x3")) return rc;
    hipLaunchKernelGGL((mesh_split_kernel<4, 24, MG, ABL>), grid, dim3(C::THREADS), (size_t)C::LDS, s, reinterpret_cast<const char*>(xsplit),
                       reinterpret_cast<const char*>(bsplit), v_shaped, a, w_idx, w_val, reinterpret_cast<f3*>(verts), M, V,
                       ceil_div(rows, SBK), tiles_m, tiles_m_per_xcd, pick_slot, reinterpret_cast<f3*>(picked), n_picked, mesh_row, group_rows, stagger);
    return check_launch("hps_smpl_mesh_fused_shared_shape_bf16x3");
}

}  // namespace hps

using namespace hps;

extern "C" size_t hps_smpl_split_bf16x3_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)ceil_div(rows, SBK) * SBK * (size_t)cols * 6;
}

extern "C" int hps_smpl_split_bf16x3_mesh_tile(void) {
#ifdef HPS_DEV_BUILD
    if (g_split_mg == 2) return 64;
#endif
    return 128;
}

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_mesh_split_groups(int mg) {
    g_split_mg = mg;
    return HPS_OK;
}
extern "C" int hps_dev_mesh_split_stagger(int units) {
    g_split_stagger = units;
    return HPS_OK;
}
extern "C" int hps_dev_mesh_split_ablate(int ablate) {
    g_split_abl = ablate;
    return HPS_OK;
}
#endif

extern "C" int hps_smpl_split_bf16x3(const float* src, int rows, int ld, int cols, int tile_cols, void* dst, hps_stream_t stream) {
    if (!src || !dst) return bad_arg("hps_smpl_split_bf16x3: null pointer");
    if (rows <= 0 || cols <= 0) return HPS_OK;
    if (tile_cols != 64 && tile_cols != 128 && tile_cols != SN)
        return bad_arg("hps_smpl_split_bf16x3: tile_cols must be hps_smpl_split_bf16x3_mesh_tile() (mesh operand) or 192 (blend matrix panel)");
    if (cols % tile_cols != 0 || ld < cols) return bad_arg("hps_smpl_split_bf16x3: cols must be a multiple of tile_cols and ld >= cols");
    const int nchunks = ceil_div(rows, SBK);
    if (2 * nchunks > 65535) return bad_arg("hps_smpl_split_bf16x3: too many rows");
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(ceil_div(cols, 256), 2 * nchunks), dim3(256), 0, (hipStream_t)stream, src, rows, ld, cols,
                       tile_cols, reinterpret_cast<uint4*>(dst), nchunks);
    return check_launch("hps_smpl_split_bf16x3");
}

extern "C" int hps_smpl_mesh_fused_shared_shape_bf16x3(const void* xsplit, const void* bsplit, const float* v_shaped,
                                                       const int32_t* mesh_row, const int32_t* group_rows, const float* a,
                                                       const int32_t* w_idx, const float* w_val, int K, int num_joints, float* verts,
                                                       int M, int V, int rows, int mp, const int32_t* pick_slot, float* picked,
                                                       int n_picked, hps_stream_t stream) {
    if (!xsplit || !bsplit || !v_shaped || !mesh_row || !group_rows || !a || !w_idx || !w_val || !verts || !pick_slot || !picked)
        return bad_arg("hps_smpl_mesh_fused_shared_shape_bf16x3: null pointer");
    if (rows <= 0 || n_picked <= 0) return bad_arg("hps_smpl_mesh_fused_shared_shape_bf16x3: rows and n_picked must be positive");
    if (M <= 0 || V <= 0) return HPS_OK;
    const int mg = hps_smpl_split_bf16x3_mesh_tile() / 32;
    if (mp % (32 * mg) != 0 || mp < M) return bad_arg("hps_smpl_mesh_fused_shared_shape_bf16x3: mp must be a multiple of the mesh tile covering M");
    if (K != 4 || num_joints != 24) {
        set_error("hps_smpl_mesh_fused_shared_shape_bf16x3: exists for K = 4, 24 joints (SMPL)");
        return HPS_E_UNSUPPORTED;
    }
#ifdef HPS_DEV_BUILD
#define HPS_SPLIT_ARGS xsplit, bsplit, v_shaped, mesh_row, group_rows, a, w_idx, w_val, verts, M, V, rows, pick_slot, picked, n_picked, (hipStream_t)stream
    if (mg == 4 && g_split_abl == 1) return launch_split<4, 1>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 2) return launch_split<4, 2>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 3) return launch_split<4, 3>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 4) return launch_split<4, 4>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 5) return launch_split<4, 5>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 6) return launch_split<4, 6>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 7) return launch_split<4, 7>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 8) return launch_split<4, 8>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 9) return launch_split<4, 9>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 10) return launch_split<4, 10>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 11) return launch_split<4, 11>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 12) return launch_split<4, 12>(HPS_SPLIT_ARGS);
    if (mg == 4 && g_split_abl == 13) return launch_split<4, 13>(HPS_SPLIT_ARGS);
#undef HPS_SPLIT_ARGS
    if (mg == 2) return launch_split<2>(xsplit, bsplit, v_shaped, mesh_row, group_rows, a, w_idx, w_val, verts, M, V, rows, pick_slot, picked, n_picked, (hipStream_t)stream);
#endif
    return launch_split<4>(xsplit, bsplit, v_shaped, mesh_row, group_rows, a, w_idx, w_val, verts, M, V, rows, pick_slot, picked, n_picked, (hipStream_t)stream);
}

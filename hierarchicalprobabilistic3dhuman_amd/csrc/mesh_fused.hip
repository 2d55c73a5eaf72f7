// Fused mesh kernel: blend shapes (fp32 MFMA GEMM) + linear blend skinning in one pass -- smplx lbs steps (1), (3), (5)
//  without the v_posed round trip through HBM.
//
//   v_posed[m, v, c] = v_template[v, c] + sum_k xt[k, m] * bmat[k, 3 v + c]        (K = betas + 207 pose features)
//   verts[m, v]      = (sum_j w[v, j] A[m, j]) [v_posed[m, v]; 1] (+ transl[m])
//
// The unfused pair (blend_gemm_kernel -> 540 MB of v_posed -> lbs_kernel) moved 1.08 GB per 6 528 meshes that only
// existed to hand the GEMM result to the skinning; here the GEMM tile stays in the MFMA accumulators and is skinned in
// the epilogue, so the only HBM stream left is the write of verts.  The kernel is bound by the fp32 MFMA rate
// (2 * 224 * 3 * 6 890 FLOP per mesh), not by HBM.
//
// Workgroup = 256 threads = 4 waves as 2 (mesh groups of 32) x 2 (vertex groups of 32); tile = 64 meshes x 64 vertices.
// The blend matrix is stored panel-permuted (bmat_p): the 192 columns of a 64-vertex panel are [x of the 64 vertices |
// y | z], so the three 32x32 MFMA tiles of a wave are the x, y and z coordinates of the SAME 32 vertices: with the mesh
// fragment as the MFMA's row operand and the blend-matrix fragment as its column operand a lane ends with one vertex
// (column = lane & 31) and 16 meshes (rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) -- x, y, z in registers, no transpose.
// K loop: 16-row chunks of both operands (k-major, exactly as they lie in HBM) land in LDS by LDS-DMA, one chunk ahead;
// fragments are conflict-free 32-lane row reads.  MFMA k order = blend_gemm_kernel's (pairs (2i, 2i+1), ascending):
// identical bits.  Epilogue: the skinning transforms A of the tile's meshes are DMA'd into the LDS that held the operand
// chunks, in two passes of 32 meshes (36 KiB for SMPL: the first / second half of each wave's 32 meshes), and every lane
// skins its vertex for its 16 meshes with skin_vertex<K> (the function lbs_kernel uses) and stores 12-byte records:
// 384 contiguous bytes per mesh per wave instruction.
//
// Why small workgroups: tools/mfma_valu_overlap.hip shows that on gfx950 fp32 VALU work does NOT overlap with
// v_mfma_f32_32x32x2_f32, neither in one wave nor across the waves of a SIMD (times add) -- so the epilogue's VALU
// work is a fixed cost and what can be hidden is only latency (operand / A DMA, barriers, stores).  36 KiB of LDS and
// <= 128 registers put four independent workgroups on a CU (one wave of each per SIMD), each in its own phase.  The first
// version (512 threads, 64 meshes x 128 vertices, all 72 KiB of A at once: two workgroups per CU) ran 0.60 ms at 6 528
// meshes against 0.46 ms for its K loop alone.
// Block -> (mesh tile, panel): block ids congruent mod 8 (one XCD) own the same mesh tiles, and every XCD walks the
// panels in order, so an XCD's L2 holds its own xt / A slices plus the few panels in flight; bmat_p streams from the
// memory-side cache once per XCD.

#include <type_traits>

#include "hps_common.h"

namespace hps {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FM = 64;            // meshes per workgroup tile
constexpr int FV = 64;            // vertices per panel
constexpr int FN = 3 * FV;        // blend-matrix columns per panel
constexpr int FBK = 16;           // K rows per chunk
constexpr int FT = 256;           // threads per workgroup
constexpr int FW = FT / 64;       // waves
constexpr int F_CHUNK_FLOATS = FBK * (FM + FN);          // 4 096 floats = 16 KiB: [FBK][FM] then [FBK][FN]
constexpr int F_XP = FBK * FM / 256;                     // 1 KiB DMA pieces of the mesh operand per chunk (4)
constexpr int F_BP = FBK * FN / 256;                     // ... of the blend matrix per chunk (12)
static_assert(F_XP == FW && F_BP % FW == 0, "DMA pieces must divide evenly over the waves");

// ABL: profiling ablations, dev library only (hps_dev_mesh_fused); the product instantiates ABL = 0.
// JC: joints per mesh as a compile-time constant (24 = SMPL: every LDS offset of the epilogue folds into an immediate), 0 = runtime J;
// HAS_T: a per-mesh translation is added (smplx SMPL.forward step (7)).
// TAIL: k-pairs of the LAST chunk that carry data (compile-time: a run-time bound inside the unrolled MFMA run cost 30 % -- the
// branch per k-step broke the pinned schedule and put an accumulator into scratch); 8 = the whole chunk.
// ST: operand chunks resident in LDS (K-loop stages).  2 = the throughput form (four workgroups per CU cover each other's fetch
// latency; 36 KiB).  4 (round 5) = chunks fetched THREE ahead, for calls whose meshes fill at most two tiles (one image at a time:
// 52 meshes): 108-216 workgroups, at most one per CU, and the 14 chunks of the 18.6 MB blend matrix arrived one round trip after the
// other (55 us for 52 meshes).  Pure pipelining: the same MFMAs in the same order, identical bits.
// PICK (round 5): the vertices the joint regressors read (pick_slot[v] >= 0: 198 of SMPL's 6 890) are ALSO written to a compact
// (M, n_picked, 3) array -- the lane that skins a vertex has it in registers, and hps_smpl_joints then reads 2.4 KB per mesh in one
// place instead of gathering 276 scattered 12-byte records from the 83 KB mesh (60 us per 6 528 meshes, bound by the request rate).
// VS (round 6): meshes SHARE their shape -- every mesh of an image has the image's betas (use_mean_shape, the reference's predict
// default: utils/sampling_utils.py:178-179) -- so the shape blend is not part of the GEMM at all: ``v_template`` is then the (R, V, 3) array
// of the R distinct shaped templates v_template + S beta_r (hps_smpl_v_shaped: smplx lbs step (1), once per image), ``xt`` / ``bmat_p`` hold
// the 207 pose rows only (K = 207 -> kp = 208 = thirteen whole chunks: 312 MFMAs per wave instead of 327) and the lane adds its mesh's
// shaped template where the plain form adds v_template: v_posed = v_shaped + P pf, in smplx's own order of the two additions.  A lane's 16
// meshes lie in ONE group of 32 consecutive meshes; ``group_rows`` describes the group as (row A, row B, split): local mesh < split has
// template row A, the others row B (a tile of sample meshes spans at most two images) -- both rows are fetched before the K loop;
// split < 0 marks a group whose rows change more than once (the mode / T-pose meshes: one image each): its lanes fetch per mesh by ``mesh_row``.
template <int K, int ABL, int JC, bool HAS_T, int TAIL = FBK / 2, int ST = 2, bool PICK = false, bool VS = false>
__global__ __launch_bounds__(FT, 4) void mesh_fused_kernel(
    const float* __restrict__ xt, const float* __restrict__ bmat_p, const float* __restrict__ v_template,
    const float* __restrict__ a, const int32_t* __restrict__ w_idx, const float* __restrict__ w_val, int J,
    const float* __restrict__ transl, f3* __restrict__ verts, int M, int V, int kp, int mp, int np, int tiles_m,
    int tiles_m_per_xcd, const int32_t* __restrict__ pick_slot, f3* __restrict__ picked, int n_picked,
    const int32_t* __restrict__ mesh_row, const int32_t* __restrict__ group_rows) {
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // union: ST operand chunks | A of 32 of the tile's meshes

    // tiles_m_per_xcd == 0: fewer than eight mesh tiles -- plain mapping (block = panel * tiles_m + tile), consecutive panels on
    // consecutive XCDs.  (With the XCD-aware mapping below a call of ONE mesh tile -- one image at a time, 52 meshes -- had all of
    // its 108 working blocks at ids = 0 mod 8: on ONE XCD's 32 CUs, the other seven idle: 55-58 us per call.)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile_m = tiles_m_per_xcd ? (local % tiles_m_per_xcd) * 8 + xcd : (int)blockIdx.x % tiles_m;
    const int panel = tiles_m_per_xcd ? local / tiles_m_per_xcd : (int)blockIdx.x / tiles_m;
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * FM;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kl = lane >> 5, il = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    // this lane's vertex and its skinning weights (kept in registers through the K loop)
    const int v = panel * FV + wn * 32 + il;
    const bool live_v = v < V;
    const int vc = live_v ? v : V - 1;
    int idx[K];
    float w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        idx[k] = w_idx[(size_t)vc * K + k] * 12;
        w[k] = w_val[(size_t)vc * K + k];
    }
    f3 vt, vtb;
    int split = 32;
    if (VS) {
        const int32_t* gr = group_rows + 3 * __builtin_amdgcn_readfirstlane((m0 + wm * 32) >> 5);      // wave-uniform: scalar loads
        const int row_a = gr[0], row_b = gr[1];
        split = gr[2];
        vt = reinterpret_cast<const f3*>(v_template)[(size_t)row_a * V + vc];
        vtb = reinterpret_cast<const f3*>(v_template)[(size_t)row_b * V + vc];
    } else {
        vt = reinterpret_cast<const f3*>(v_template)[vc];
        vtb = vt;
    }
    const int split_lane = split - 4 * kl;                    // local mesh 4 kl + dr < split  <=>  dr < split_lane
    const int pick = PICK && live_v ? pick_slot[vc] : -1;     // this lane's slot in the compact array of regressor vertices, or -1

    // LDS-DMA pieces (1 KiB each): a chunk is F_XP pieces of xt rows ([FBK][FM]) and F_BP of bmat_p rows ([FBK][FN]);
    // wave w moves mesh-operand piece w and blend-matrix pieces w, w + 4, w + 8.
    unsigned b_off[F_BP / FW];
#pragma unroll
    for (int j = 0; j < F_BP / FW; ++j) {
        const int f = 256 * (wave + FW * j) + 4 * lane;
        b_off[j] = (unsigned)(((f / FN) * np + (f % FN)) * 4);
    }
    unsigned x_off;
    {
        const int f = 256 * wave + 4 * lane;
        x_off = (unsigned)(((f / FM) * mp + (f % FM)) * 4);
    }
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)(smem);
    const float* b_src = bmat_p + (size_t)panel * FN;
    const float* x_src = xt + m0;
    auto dma_chunk = [&](int buf) {
        const unsigned base = lds0 + (unsigned)buf * F_CHUNK_FLOATS * 4;
        lds_dma16(x_off, x_src, base + (unsigned)wave * 1024);
#pragma unroll
        for (int j = 0; j < F_BP / FW; ++j) lds_dma16(b_off[j], b_src, base + FBK * FM * 4 + (unsigned)(wave + FW * j) * 1024);
        b_src += (size_t)FBK * np;
        x_src += (size_t)FBK * mp;
    };
    // piece p of the wave's four pieces of a chunk (0: mesh operand, 1..3: blend matrix), for the spread issue in the K loop
    auto dma_piece = [&](int p, int buf, const float* xs, const float* bs) {
        const unsigned base = lds0 + (unsigned)buf * F_CHUNK_FLOATS * 4;
        if (p == 0) lds_dma16(x_off, xs, base + (unsigned)wave * 1024);
        else lds_dma16(b_off[p - 1], bs, base + FBK * FM * 4 + (unsigned)(wave + FW * (p - 1)) * 1024);
    };

    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

    // kp = K rows that carry data (even).  Chunks are 16 rows; the last one runs only its TAIL k-pairs: the rows behind kp are zero
    // padding (SMPL: K = 217 -> kp = 218, TAIL = 5: nine MFMAs fewer of 336 per wave; adding their zero products changes nothing,
    // so the bits are those of the padded sum).
    const int nchunks = (kp + FBK - 1) / FBK;
    if (ABL != 4) {
        dma_chunk(0);
#pragma unroll
        for (int q = 1; q < ST - 1; ++q)
            if (q < nchunks) dma_chunk(q);
    }
    static_assert(1 + F_BP / FW == FBK / 4, "four DMA pieces per wave and chunk, one per two k-steps");
    constexpr int PIECES = 1 + F_BP / FW;                  // DMA pieces per wave and chunk
    auto do_chunk = [&](auto pairs_c, int c, bool more_in) __attribute__((always_inline)) {
        constexpr int PAIRS = decltype(pairs_c)::value;
        const bool more = ST == 2 ? more_in : (ABL != 4 && c + ST - 1 < nchunks);      // is a chunk fetched during this one?
        // chunk c has landed (with more than two stages up to ST - 2 younger chunks, PIECES pieces each, may still be in flight)
        {
            const int younger = min(ST - 2, nchunks - 1 - c);
            if (ST >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
            else if (ST >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                   // ... for everyone; and everyone is done with the buffer refilled below
        // the next chunk's four DMA pieces go out one per two k-steps of this chunk's MFMAs, not in a burst (csrc/conv_pad.hip,
        // tools/mfma_dma_overlap.hip: a piece costs the SIMD 36-57 cycles that are better paid between MFMAs than before them)
        constexpr bool burst = ABL == 5 && ST == 2;       // dev ablation: the earlier burst after the barrier
        const float* nx_src = x_src;
        const float* nb_src = b_src;
        const int nbuf = ST == 2 ? (c + 1) & 1 : (c + ST - 1) % ST;
        if (more && burst) dma_chunk((c + 1) & 1);
        else if (more) { b_src += (size_t)FBK * np; x_src += (size_t)FBK * mp; }
        const float* sX = smem + (ST == 2 ? (c & 1) : c % ST) * F_CHUNK_FLOATS;
        const float* sB = sX + FBK * FM;
        // all fragments of the chunk first (independent LDS reads in flight), then the MFMAs back to back
        float af[PAIRS], bx[PAIRS], by[PAIRS], bz[PAIRS];
#pragma unroll
        for (int k = 0; k < 2 * PAIRS; k += 2) {
            af[k / 2] = sX[(k + kl) * FM + wm * 32 + il];
            const float* brow = sB + (k + kl) * FN + wn * 32 + il;
            bx[k / 2] = brow[0]; by[k / 2] = brow[FV]; bz[k / 2] = brow[2 * FV];
        }
        __builtin_amdgcn_sched_barrier(0);                 // keep the reads ahead of the MFMAs (hipcc would sink each to its use)
#pragma unroll
        for (int k = 0; k < PAIRS; ++k) {
            if (ABL == 2) {                                // no MFMA: keep the fragment reads alive
                acc[0][0] += af[k] * bx[k]; acc[1][0] += af[k] * by[k]; acc[2][0] += af[k] * bz[k];
                continue;
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], bx[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], by[k], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], bz[k], acc[2], 0, 0, 0);
            if (more && !burst && (k & 1) == 0) {
                __builtin_amdgcn_sched_barrier(0);
                dma_piece(k / 2, nbuf, nx_src, nb_src);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int c = 0; c + 1 < nchunks; ++c) do_chunk(std::integral_constant<int, FBK / 2>(), c, ABL != 4);
    do_chunk(std::integral_constant<int, TAIL>(), nchunks - 1, false);

    if (ABL == 3) {                                        // K loop only: one never-taken store keeps the accumulators alive
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r] + acc[2][r];
        if (t == 12345.678f) verts[0].x = t;
        return;
    }

    // Skinning, two passes of 32 meshes: pass p stages A of meshes [16 p, 16 p + 16) and [32 + 16 p, 32 + 16 p + 16) of the
    // tile -- the meshes of accumulator registers r = 8 p .. 8 p + 7 of both mesh groups -- as LDS slots 0..15 and 16..31.
    const int a_stride = JC ? JC * 12 : J * 12;
    const int half_bytes = 16 * a_stride * 4;              // one contiguous source range of J * 768 bytes (whole 1 KiB pieces iff J % 4 == 0;
                                                           // the last piece is cut by the off < valid mask otherwise: tested with J = 22)
    const int slot0 = wm * 16 + 4 * kl;                    // the lane's first slot
    int aoff[K];                                           // float offset of A[slot0][joint_k] in LDS
#pragma unroll
    for (int k = 0; k < K; ++k) aoff[k] = slot0 * a_stride + idx[k];
    char* const vbase = reinterpret_cast<char*>(verts) + (size_t)(m0 + wm * 32) * V * 12;      // wave-uniform
    const unsigned voff = ((unsigned)(4 * kl) * (unsigned)V + (unsigned)v) * 12u;              // per lane
    char* const pbase = PICK ? reinterpret_cast<char*>(picked) + (size_t)(m0 + wm * 32) * n_picked * 12 : nullptr;
    const unsigned poff = PICK ? ((unsigned)(4 * kl) * (unsigned)n_picked + (unsigned)max(pick, 0)) * 12u : 0u;
    // MANY (VS only): the wave's group of 32 meshes has more than two templates (split < 0: the mode / T-pose meshes, one image each --
    // 4 of 204 groups at B = 64, N = 100) and every mesh fetches its own inside the loop.  That form is a COPY of the epilogue, chosen by one
    // wave-uniform branch in front of it: a load -- or a branch around one -- inside the loop of the common form puts a vmcnt(0) there,
    // which on gfx950 also waits for the previous mesh's store, and splits the loop body into blocks hipcc does not schedule across
    // (measured: +7 us on the whole launch, more than the fifteen MFMAs per wave the K = 207 form saves).  Both copies pass the same
    // barriers, so the waves of a workgroup may take different ones.
    auto epilogue = [&](auto many_c) __attribute__((always_inline)) {
    constexpr bool MANY = decltype(many_c)::value;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                                   // operand chunks / the previous pass's transforms are dead
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int mh = m0 + 32 * h + 16 * pass;                                              // first mesh of this range
            const int valid = max(0, min(16, M - mh)) * a_stride * 4;                          // bytes that exist in `a`
            const float* a_src = a + (size_t)mh * a_stride;
            for (int piece = wave; piece * 1024 < half_bytes; piece += FW) {
                const int off = piece * 1024 + lane * 16;
                if (off < valid) lds_dma16((unsigned)off, a_src, lds0 + (unsigned)(h * half_bytes + piece * 1024));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // branch-free: the per-lane parts of every address were formed once, the per-r parts are compile-time / wave-uniform
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * pass + q;
            const int dr = (r & 3) + 8 * (r >> 2);         // mesh row step of accumulator register r (within the wave's 32)
            const int ds = (q & 3) + 8 * (q >> 2);         // ... within the pass's 16 slots of this mesh group
            const int m = m0 + wm * 32 + 4 * kl + dr;
            float tx = 0.f, ty = 0.f, tz = 0.f;
            if (HAS_T) {
                const float* t = transl + (size_t)min(m, M - 1) * 3;
                tx = t[0]; ty = t[1]; tz = t[2];
            }
            f3 base = vt;
            if (VS && !MANY) {                             // template A / B of the group by the split: three selects, no load
                const bool first = dr < split_lane;
                base.x = first ? vt.x : vtb.x; base.y = first ? vt.y : vtb.y; base.z = first ? vt.z : vtb.z;
            }
            if (VS && MANY) base = reinterpret_cast<const f3*>(v_template)[(size_t)mesh_row[min(m, M - 1)] * V + vc];
            f3 pv;
            pv.x = base.x + acc[0][r]; pv.y = base.y + acc[1][r]; pv.z = base.z + acc[2][r];
            f3 o;
            if (ABL == 1) {
                o = pv;
            } else {
                int ao[K];
#pragma unroll
                for (int k = 0; k < K; ++k) ao[k] = aoff[k] + ds * a_stride;
                o = skin_vertex<K>(smem, ao, w, pv, tx, ty, tz);
            }
            // Pin the result in front of the guard: hipcc otherwise sinks the whole skinning of a mesh (12 LDS reads, the FMAs)
            // into the guarded store's block, where the reads cannot be issued under the previous mesh's arithmetic and the
            // block's entry waits vmcnt(0) -- i.e. for the previous mesh's store -- on account of the v_template load.
            asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z));
            if (live_v && m < M) *reinterpret_cast<f3*>(vbase + (size_t)dr * V * 12 + voff) = o;
            if (PICK && pick >= 0 && m < M) *reinterpret_cast<f3*>(pbase + (size_t)dr * n_picked * 12 + poff) = o;
        }
    }
    };
    if (VS && split < 0) epilogue(std::true_type());
    else epilogue(std::false_type());
}

#ifdef HPS_DEV_BUILD
static size_t g_mesh_lds_floor = 0;        // hps_dev_mesh_lds_floor
static int g_mesh_stages = 0;              // hps_dev_mesh_stages: 2 = always the two-stage K loop (A/B and bit-level cross-check)
#endif

template <int K, int ABL, int JC, bool HAS_T>
static int launch_fused_cfg(const float* xt, const float* bmat_p, const float* v_template, const float* a, const int32_t* w_idx,
                        const float* w_val, int J, const float* transl, float* verts, int M, int V, int kp, int mp, int np,
                        hipStream_t s, const int32_t* pick_slot = nullptr, float* picked = nullptr, int n_picked = 0,
                        const int32_t* mesh_row = nullptr, const int32_t* group_rows = nullptr) {
    size_t lds = (size_t)4 * max(2 * F_CHUNK_FLOATS, 32 * J * 12);
#ifdef HPS_DEV_BUILD
    if (g_mesh_lds_floor > lds) lds = g_mesh_lds_floor;     // experiment: fewer workgroups per CU (a larger LDS request, unused)
#endif
    const int tiles_m = ceil_div(M, FM), n_panels = ceil_div(V, FV);
    // at most two mesh tiles (one image at a time: 52 meshes): at most one workgroup per CU -- the four-stage K loop (same bits)
    const bool few = tiles_m <= 2
#ifdef HPS_DEV_BUILD
                     && g_mesh_stages != 2
#endif
        ;
    const int tiles_m_per_xcd = tiles_m >= 8 ? ceil_div(tiles_m, 8) : 0;          // 0: plain block mapping (see the kernel)
    const dim3 grid(tiles_m_per_xcd ? tiles_m_per_xcd * 8 * n_panels : tiles_m * n_panels);
    // the last chunk's data-carrying k-pairs: SMPL (K = 10 + 207 -> kp = 218) has 5 of 8; that case is instantiated for the product
    // configuration, every other tail runs the whole (zero-padded) chunk
    const int tail = (kp - FBK * ((kp + FBK - 1) / FBK - 1)) / 2;
    f3* pk = reinterpret_cast<f3*>(picked);
#define HPS_MESH_LAUNCH(...)                                                                                                       \
    do {                                                                                                                           \
        if (int rc = grant_lds<&mesh_fused_kernel<__VA_ARGS__>>(160 * 1024, "hps_smpl_mesh_fused")) return rc;                    \
        hipLaunchKernelGGL((mesh_fused_kernel<__VA_ARGS__>), grid, dim3(FT), lds, s, xt, bmat_p, v_template, a, w_idx, w_val, J,   \
                           transl, reinterpret_cast<f3*>(verts), M, V, kp, mp, np, tiles_m, tiles_m_per_xcd, pick_slot, pk, n_picked, \
                           mesh_row, group_rows);                                                                                 \
    } while (0)
    if constexpr (K == 4 && JC == 24 && ABL == 0 && !HAS_T) {     // shared shapes (VS): SMPL's 207 pose rows, thirteen whole chunks
        if (group_rows) {
            if (tail != 8 || !pick_slot || !mesh_row) {
                set_error("hps_smpl_mesh_fused_shared_shape: exists for kp = 208 (SMPL's 207 pose rows) with the side output only");
                return HPS_E_UNSUPPORTED;
            }
            if (few) {
                lds = (size_t)4 * max(4 * F_CHUNK_FLOATS, 32 * J * 12);
                HPS_MESH_LAUNCH(K, ABL, JC, HAS_T, 8, 4, true, true);
            } else {
                HPS_MESH_LAUNCH(K, ABL, JC, HAS_T, 8, 2, true, true);
            }
            return check_launch("hps_smpl_mesh_fused_shared_shape");
        }
    }
    if (group_rows) {
        set_error("hps_smpl_mesh_fused_shared_shape: exists for K = 4, 24 joints, no translation");
        return HPS_E_UNSUPPORTED;
    }
    if constexpr (K == 4 && JC == 24 && ABL == 0) {          // the product configuration (SMPL): tail, stages and the side output
        if (tail == 5) {
            if (few) {
                lds = (size_t)4 * max(4 * F_CHUNK_FLOATS, 32 * J * 12);
                if (pick_slot) HPS_MESH_LAUNCH(K, ABL, JC, HAS_T, 5, 4, true);
                else HPS_MESH_LAUNCH(K, ABL, JC, HAS_T, 5, 4, false);
            } else {
                if (pick_slot) HPS_MESH_LAUNCH(K, ABL, JC, HAS_T, 5, 2, true);
                else HPS_MESH_LAUNCH(K, ABL, JC, HAS_T, 5, 2, false);
            }
            return check_launch("hps_smpl_mesh_fused");
        }
    }
    if (pick_slot) {
        set_error("hps_smpl_mesh_fused_picks: the side output exists for K = 4, 24 joints, kp = 218 (SMPL) only");
        return HPS_E_UNSUPPORTED;
    }
    HPS_MESH_LAUNCH(K, ABL, JC, HAS_T);
#undef HPS_MESH_LAUNCH
    return check_launch("hps_smpl_mesh_fused");
}

template <int K, int ABL = 0>
static int launch_fused(const float* xt, const float* bmat_p, const float* v_template, const float* a, const int32_t* w_idx,
                        const float* w_val, int J, const float* transl, float* verts, int M, int V, int kp, int mp, int np,
                        hipStream_t s, const int32_t* pick_slot = nullptr, float* picked = nullptr, int n_picked = 0,
                        const int32_t* mesh_row = nullptr, const int32_t* group_rows = nullptr) {
    if (J == 24 && !transl) return launch_fused_cfg<K, ABL, 24, false>(xt, bmat_p, v_template, a, w_idx, w_val, J, transl, verts, M, V, kp, mp, np, s, pick_slot, picked, n_picked, mesh_row, group_rows);
    if (group_rows) { set_error("hps_smpl_mesh_fused_shared_shape: 24 joints, no translation"); return HPS_E_UNSUPPORTED; }
    if (J == 24) return launch_fused_cfg<K, ABL, 24, true>(xt, bmat_p, v_template, a, w_idx, w_val, J, transl, verts, M, V, kp, mp, np, s, pick_slot, picked, n_picked);
    if (pick_slot) { set_error("hps_smpl_mesh_fused_picks: 24 joints only"); return HPS_E_UNSUPPORTED; }
    if constexpr (K == 4) {          // a run-time joint count costs registers: only the K = 4 instantiation stays free of scratch
        if (!transl) return launch_fused_cfg<K, ABL, 0, false>(xt, bmat_p, v_template, a, w_idx, w_val, J, transl, verts, M, V, kp, mp, np, s);
        return launch_fused_cfg<K, ABL, 0, true>(xt, bmat_p, v_template, a, w_idx, w_val, J, transl, verts, M, V, kp, mp, np, s);
    } else {
        set_error("hps_smpl_mesh_fused: K = %d is fused only for num_joints = 24 (use hps_smpl_blend + hps_smpl_lbs)", K);
        return HPS_E_UNSUPPORTED;
    }
}

}  // namespace hps

using namespace hps;

extern "C" int hps_smpl_mesh_fused_np(int V) { return V > 0 ? ceil_div(V, FV) * FN : 0; }

static int fused_check_args(const void* xt, const void* bmat_p, const void* v_template, const void* a, const void* w_idx,
                            const void* w_val, const void* verts, int num_joints, int M, int V, int kp, int mp, int np) {
    if (!xt || !bmat_p || !v_template || !a || !w_idx || !w_val || !verts) return bad_arg("hps_smpl_mesh_fused: null pointer");
    if (kp <= 0 || kp % 2 != 0) return bad_arg("hps_smpl_mesh_fused: kp must be positive and even");
    if (num_joints < 1 || num_joints > 32) return bad_arg("hps_smpl_mesh_fused: num_joints must be 1..32");
    if (M <= 0 || V <= 0) return 1;                                     /* nothing to do */
    if (mp % FM != 0 || mp < ceil_div(M, FM) * FM) return bad_arg("hps_smpl_mesh_fused: mp must be a multiple of 64 covering M");
    if (np != hps_smpl_mesh_fused_np(V)) return bad_arg("hps_smpl_mesh_fused: np must be hps_smpl_mesh_fused_np(V)");
    return HPS_OK;
}

extern "C" int hps_smpl_mesh_fused(const float* xt, const float* bmat_p, const float* v_template, const float* a,
                                   const int32_t* w_idx, const float* w_val, int K, int num_joints, const float* transl,
                                   float* verts, int M, int V, int kp, int mp, int np, hps_stream_t stream) {
    const int rc = fused_check_args(xt, bmat_p, v_template, a, w_idx, w_val, verts, num_joints, M, V, kp, mp, np);
    if (rc != HPS_OK) return rc > 0 ? HPS_OK : rc;
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 4: return launch_fused<4>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        case 8: return launch_fused<8>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        case 12: return launch_fused<12>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        // K = 24 (dense skinning weights) does not fit the register budget of four workgroups per CU beside the accumulators
        // (104-708 bytes of scratch per lane when instantiated): such models take the unfused pair
        default: set_error("hps_smpl_mesh_fused: K=%d unsupported (4 with any joint count; 8, 12 with 24 joints); use "
                           "hps_smpl_blend + hps_smpl_lbs", K); return HPS_E_UNSUPPORTED;
    }
}

extern "C" int hps_smpl_mesh_fused_picks(const float* xt, const float* bmat_p, const float* v_template, const float* a,
                                         const int32_t* w_idx, const float* w_val, int K, int num_joints, const float* transl,
                                         float* verts, int M, int V, int kp, int mp, int np, const int32_t* pick_slot, float* picked,
                                         int n_picked, hps_stream_t stream) {
    if (!pick_slot || !picked || n_picked <= 0) return bad_arg("hps_smpl_mesh_fused_picks: pick_slot / picked / n_picked");
    const int rc = fused_check_args(xt, bmat_p, v_template, a, w_idx, w_val, verts, num_joints, M, V, kp, mp, np);
    if (rc != HPS_OK) return rc > 0 ? HPS_OK : rc;
    if (K != 4) { set_error("hps_smpl_mesh_fused_picks: K = %d (the side output exists for K = 4 only)", K); return HPS_E_UNSUPPORTED; }
    return launch_fused<4>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, (hipStream_t)stream,
                           pick_slot, picked, n_picked);
}

extern "C" int hps_smpl_mesh_fused_shared_shape(const float* xt_pose, const float* bmat_p_pose, const float* v_shaped,
                                                const int32_t* mesh_row, const int32_t* group_rows, const float* a,
                                                const int32_t* w_idx, const float* w_val, int K, int num_joints, float* verts, int M,
                                                int V, int kp, int mp, int np, const int32_t* pick_slot, float* picked,
                                                int n_picked, hps_stream_t stream) {
    if (!mesh_row || !group_rows || !pick_slot || !picked || n_picked <= 0)
        return bad_arg("hps_smpl_mesh_fused_shared_shape: mesh_row / group_rows / pick_slot / picked / n_picked");
    const int rc = fused_check_args(xt_pose, bmat_p_pose, v_shaped, a, w_idx, w_val, verts, num_joints, M, V, kp, mp, np);
    if (rc != HPS_OK) return rc > 0 ? HPS_OK : rc;
    if (K != 4) { set_error("hps_smpl_mesh_fused_shared_shape: K = %d (exists for K = 4 only)", K); return HPS_E_UNSUPPORTED; }
    return launch_fused<4>(xt_pose, bmat_p_pose, v_shaped, a, w_idx, w_val, num_joints, nullptr, verts, M, V, kp, mp, np,
                           (hipStream_t)stream, pick_slot, picked, n_picked, mesh_row, group_rows);
}

// smplx lbs step (1) for R distinct shapes: v_shaped[r, n] = v_template[n] + sum_l betas[r, l] * shapedirs[l, n]  (n = 3 v + c; one fused
// multiply-add chain over l ascending, the template added last -- smplx: v_template + blend_shapes(betas, shapedirs))
namespace hps {
__global__ __launch_bounds__(256) void v_shaped_kernel(const float* __restrict__ betas, int nb, const float* __restrict__ shape_rows, int ld,
                                                       const float* __restrict__ v_template, float* __restrict__ v_shaped, int n3) {
    const int n = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (n >= n3) return;
    float acc = 0.0f;
    for (int l = 0; l < nb; ++l) acc = __builtin_fmaf(betas[r * nb + l], shape_rows[(size_t)l * ld + n], acc);
    v_shaped[(size_t)r * n3 + n] = v_template[n] + acc;
}
}  // namespace hps

extern "C" int hps_smpl_v_shaped(const float* betas, int num_betas, const float* shape_rows, int ld, const float* v_template,
                                 float* v_shaped, int R, int V, hps_stream_t stream) {
    if (!betas || !shape_rows || !v_template || !v_shaped) return bad_arg("hps_smpl_v_shaped: null pointer");
    if (num_betas < 0 || ld < 3 * V) return bad_arg("hps_smpl_v_shaped: num_betas >= 0 and ld >= 3 V required");
    if (R <= 0 || V <= 0) return HPS_OK;
    if (R > 65535) return bad_arg("hps_smpl_v_shaped: at most 65535 shapes per call");
    hipLaunchKernelGGL(hps::v_shaped_kernel, dim3(ceil_div(3 * V, 256), R), dim3(256), 0, (hipStream_t)stream, betas, num_betas, shape_rows, ld,
                       v_template, v_shaped, 3 * V);
    return check_launch("hps_smpl_v_shaped");
}

#ifdef HPS_DEV_BUILD
extern "C" int hps_dev_mesh_stages(int stages) {
    g_mesh_stages = stages;
    return HPS_OK;
}
extern "C" int hps_dev_mesh_lds_floor(int bytes) {
    g_mesh_lds_floor = bytes > 0 ? (size_t)bytes : 0;
    return HPS_OK;
}

extern "C" int hps_dev_mesh_fused(const float* xt, const float* bmat_p, const float* v_template, const float* a,
                                  const int32_t* w_idx, const float* w_val, int K, int num_joints, const float* transl,
                                  float* verts, int M, int V, int kp, int mp, int np, int ablate, hps_stream_t stream) {
    const int rc = fused_check_args(xt, bmat_p, v_template, a, w_idx, w_val, verts, num_joints, M, V, kp, mp, np);
    if (rc != HPS_OK) return rc > 0 ? HPS_OK : rc;
    if (K != 4) return bad_arg("hps_dev_mesh_fused: K = 4 only");
    hipStream_t s = (hipStream_t)stream;
    switch (ablate) {
        case 0: return launch_fused<4, 0>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        case 1: return launch_fused<4, 1>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        case 2: return launch_fused<4, 2>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        case 3: return launch_fused<4, 3>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        case 4: return launch_fused<4, 4>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        case 5: return launch_fused<4, 5>(xt, bmat_p, v_template, a, w_idx, w_val, num_joints, transl, verts, M, V, kp, mp, np, s);
        default: return bad_arg("hps_dev_mesh_fused: ablate 0..5");
    }
}
#endif

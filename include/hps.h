/*
 * hps.h -- C ABI of libhps.so: the MI355X (gfx950) hot path of HierarchicalProbabilistic3DHuman.
 *
 * The reference is pure Python and has no FFI layer; these entry points are what a
 * binding for its per-image inference path would call.  Each declaration cites the reference
 * code (file:line, relative to the reference repo) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; row-major contiguous fp32
 *     (int32 for indices); the caller owns all memory (PyTorch tensors in the shipped binding);
 *   - stream is a hipStream_t passed as void*; calls only enqueue work (no allocation, no synchronisation, no host round
 *     trip) -- with three documented exceptions: hps_head_pose_levels in its host-LAPACK parity mode (it waits for each
 *     kinematic level's matrices: svd_mode HPS_SVD_HOST), hps_host_svd3_packed (a host routine) and
 *     hps_host_bind_lapack (dlopen, once per process);
 *   - return value: 0 on success, otherwise a hipError_t value (> 0) or a negative HPS_E_* code;
 *     hps_last_error() returns a thread-local description of the last failure;
 *   - no behaviour-changing global state: the library has no tuning switches (those live in libhps_dev.so,
 *     include/hps_dev.h).  What IS process-global: the LAPACK routine bound by hps_host_bind_lapack, the host-SVD worker pool
 *     it feeds, and one-time per-kernel attribute grants (LDS above 64 KiB) -- none alters results.  Safe to call from several
 *     host threads on different streams;
 *   - workspace sizes come from hps_query_workspace (nothing is allocated inside the library).
 */
#ifndef HPS_H_
#define HPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The libraries are built with -fvisibility=hidden: what is declared between this push and the pop at the end of the header is
 * the complete dynamic symbol table of the shared object (tests/test_capi_symbols.py compares it with `nm -D`). */
#pragma GCC visibility push(default)

typedef void* hps_stream_t; /* hipStream_t */

/* Entry points the package's own inference path no longer calls (round 6).  They stay exported, tested on every GPU run and
 * supported -- each is the bit-level partner or the general form of the one that replaced it -- but a new caller should start
 * from the replacement:
 *   hps_stem_winograd + hps_maxpool3x3s2_pad      -> hps_stem_winograd_pooled_nchw (max pool in the stem's epilogue, windows gathered
 *                                                    from the NCHW input); hps_stem_phase_split + hps_stem_winograd_pooled remain the
 *                                                    route of callers that write the phase frames themselves (hps_proxy_rep_phase_frames)
 *   hps_smpl_mesh_fused (no side output)          -> hps_smpl_mesh_fused_picks; with shared shapes (use_mean_shape)
 *                                                    hps_smpl_v_shaped + hps_smpl_mesh_fused_shared_shape
 *   hps_smpl_blend + hps_smpl_lbs                 -> the fused forms (the pair remains SURVEY 8(d)'s definition of the LBS kernel and the
 *                                                    route for skinning weights with more than 12 influences)
 *   one hps_conv2d_bn_act_pad call for a block's
 *   1x1/2 down-sample                             -> hps_conv2d_bn_act_pad_down (rides in the 3x3/2 convolution's launch)
 *   hps_nchw_to_padded_nhwc (+ row-mode stem)     -> the Winograd stem for the released model's shapes; the relayout + direct stem remain
 *                                                    the route of every other (C, H, W) and of the latency mode */

#define HPS_OK 0
#define HPS_E_BADARG (-1)      /* null pointer / size out of range */
#define HPS_E_UNSUPPORTED (-2) /* shape not implemented by this build */

#define HPS_SVD_HOST 0       /* host LAPACK sgesdd (the reference's own routine) */
#define HPS_SVD_DEVICE 1     /* in-kernel SVD that follows sgesdd step by step, rounding flavour HPS_SVD_ROUNDING_REFERENCE */
#define HPS_SVD_DEVICE_FMA 2 /* the same with rounding flavour HPS_SVD_ROUNDING_FMA */

/* Rounding flavours of the in-kernel SVD (csrc/svd3_gesdd.h).  The reference runs torch.svd = MKL sgesdd on the host
 * (models/poseMF_shapeGaussian_net.py:137), and MKL picks its kernels by the host CPU: on Intel hosts they use fused
 * multiply-adds, elsewhere they round like reference BLAS.  Each flavour reproduces one of them BIT FOR BIT (U, S, V of
 * 10^6 matrices of 22 families, tests/test_host_logic.py); between the two, one matrix in 10^4 comes out with a differently
 * signed singular-vector pair.  hps_host_svd_flavor() tells which one the LAPACK bound to this process matches. */
#define HPS_SVD_ROUNDING_REFERENCE 0
#define HPS_SVD_ROUNDING_FMA 1
/* OR'ed into the svd_flavor of hps_head_joint_level_svd / the svd_mode of hps_head_pose_levels (device modes): 1024-thread
 * workgroups with eight K slices for the joint MLPs instead of 256 threads with two -- the latency form for one or a few images
 * (a level 28 -> 21 us at batch 1), too large a footprint beside the chip-filling kernels of the pipelined loop.  The hidden
 * layer's partial sums are then added in another order: results differ in the last bits, so this is a property of the model
 * (PoseMFShapeGaussianNet.set_latency_mode), never of the batch. */
#define HPS_HEAD_WIDE_WORKGROUPS 0x100

#define HPS_ACT_NONE 0
#define HPS_ACT_ELU 1
#define HPS_ACT_RELU 2

int hps_version(void);
const char* hps_last_error(void);

/* Spatial partition of the chip for kernels that cannot usefully time-share it (no reference counterpart: the reference runs
 * one kernel at a time).  Creates a HIP stream restricted to CUs [first_cu, first_cu + num_cus) of EVERY XCD (0 <= first_cu,
 * first_cu + num_cus <= 32 on MI355X: 8 XCDs x 32 CUs); destroy it with hps_stream_destroy.  Used by the step pipeline for
 * small batches: the encoder's persistent 512-register Winograd workgroups need a CU to themselves and starve beside a kernel
 * whose small workgroups keep refilling every CU; on a CU subset of their own they do not.
 * Host call; allocates a stream (the one exception to "the library never allocates"). */
int hps_stream_create_cu_partition(int first_cu, int num_cus, hps_stream_t* stream_out);
int hps_stream_destroy(hps_stream_t stream);

/* Scratch / workspace sizes in BYTES for the buffers the caller hands to the entry points below (the library never allocates):
 *   HPS_WS_CONV_SPLITK (d0 = ksplit, d1 = B*Ho*Wo, d2 = Cout)  splitk_ws of hps_conv2d_bn_act_pad (0 when ksplit <= 1)
 *   HPS_WS_SMPL_MP     (d0 = M)                                 NOT bytes: the padded mesh count mp (M rounded up to 128)
 *   HPS_WS_SMPL_XT     (d0 = M, d1 = kp)                        xt of hps_smpl_pose_prep: kp * mp floats
 *   HPS_WS_SMPL_A      (d0 = M, d1 = num_joints)                a of hps_smpl_pose_prep: M * J * 12 floats
 *   HPS_WS_SMPL_VPOSED (d0 = M, d1 = V)                         v_posed of the unfused hps_smpl_blend with 128-byte aligned rows
 *   HPS_WS_HEAD_F      (d0 = B, d1 = largest level size)        f_level_dev / f_host_pinned of hps_head_pose_levels
 *   HPS_WS_HEAD_USV    (d0 = B, d1 = largest level size)        usv_level_dev / usv_host_pinned of hps_head_pose_levels
 * Unused dims are ignored.  Returns -1 (and sets hps_last_error) for an unknown `what` or negative dims. */
enum { HPS_WS_CONV_SPLITK = 0, HPS_WS_SMPL_MP = 1, HPS_WS_SMPL_XT = 2, HPS_WS_SMPL_A = 3, HPS_WS_SMPL_VPOSED = 4,
       HPS_WS_HEAD_F = 5, HPS_WS_HEAD_USV = 6 };
int64_t hps_query_workspace(int what, int64_t d0, int64_t d1, int64_t d2);

/* ------------------------------------------------------------------------------------------
 * SMPL forward  (models/smpl_official.py:27-41 -> smplx 0.1.26 SMPL.forward / lbs)
 * ---------------------------------------------------------------------------------------- */

/* Rodrigues / rest joints / forward kinematics / blend-GEMM operand, one launch.
 *   glob, body : global_orient and body_pose of M meshes; rotation matrices (M,1,3,3)/(M,J-1,3,3)
 *                if is_rotmat != 0 (pose2rot=False), else axis-angle (M,3)/(M,(J-1)*3) converted by
 *                smplx batch_rodrigues (angle = ||r + 1e-8||).
 *   betas      : (M, num_betas).
 *   j_template : (J,3)   = J_regressor @ v_template          } joint regression folded through the
 *   j_shapedirs: (J,3,nb)= J_regressor @ shapedirs           } linear shape blend (exact algebra)
 *   parents    : (J,) int32, parents[0] = -1;  depth: (J,) int32 tree depth of each joint.
 * outputs
 *   xt   : (kp, mp) blend-GEMM operand, k-major: rows [0,nb) betas, [nb, nb+9(J-1)) pose feature
 *          (R_j - I, j = 1..J-1), remaining rows zero.  Only columns < M are written.
 *   a    : (M,J,12) skinning transforms, top 3 rows of smplx's "A" (world transform with the rest
 *          pose removed), row-major 3x4.
 *   j_posed : (M,J,3) posed joint locations.   rot_out: optional (M,J,3,3) rotation matrices used.
 */
int hps_smpl_pose_prep(const float* glob, const float* body, int is_rotmat, const float* betas,
                       int num_betas, const float* j_template, const float* j_shapedirs,
                       const int32_t* parents, const int32_t* depth, int num_joints, float* xt,
                       int kp, int mp, float* a, float* j_posed, float* rot_out, int M,
                       hps_stream_t stream);

/* Shape + pose blend shapes as one fp32-MFMA GEMM:
 *   v_posed[m, n] = v_template[n] + sum_k xt[k, m] * bmat[k, n]      (n = 3*vertex + coord)
 * bmat: (kp, np) = [shapedirs ; posedirs ; 0] with np = N rounded up to 128 (zero padded),
 * xt: (kp, mp) from hps_smpl_pose_prep, mp = M rounded up to 128, kp a multiple of 16.
 * v_posed rows have pitch ld_out floats (>= N; a multiple of 32 keeps every 128-byte store segment line-aligned).
 * Replaces lbs steps (1) and (3): blend_shapes einsum + pose_feature @ posedirs. */
int hps_smpl_blend(const float* xt, const float* bmat, const float* v_template, float* v_posed,
                   int M, int N, int kp, int mp, int np, int ld_out, hps_stream_t stream);

/* Linear blend skinning, lbs step (5):  verts[m,v] = (sum_k w[v,k] * A[m, idx[v,k]]) . [v_posed[m,v]; 1]
 * Skinning weights in compressed form: w_idx/w_val (V, K) -- the K largest-support entries of each
 * row of lbs_weights, padded with (0, 0.0f); exact for any model with <= K non-zeros per row.
 * v_posed rows have pitch ld_vposed floats (>= 3 V); verts is contiguous (M,V,3) like the reference's output.
 * transl: optional (M,3) added to the result (smplx SMPL.forward step (7)), may be NULL.
 * Algorithmic HBM bytes per mesh: 12 V (v_posed) + 48 J.. (A) + 12 V (verts); SURVEY section 8(d). */
int hps_smpl_lbs(const float* v_posed, int ld_vposed, const float* a, const int32_t* w_idx,
                 const float* w_val, int K, int num_joints, const float* transl, float* verts, int M,
                 int V, hps_stream_t stream);

/* Fused blend shapes + linear blend skinning: lbs steps (1), (3), (5) in ONE kernel -- the blend GEMM tile stays in the
 * MFMA accumulators and is skinned in the epilogue, so v_posed never exists in HBM (the product path of SMPL.forward;
 * hps_smpl_blend + hps_smpl_lbs remain as the unfused definition the LBS roofline bytes of SURVEY section 8(d) refer to).
 *   verts[m,v] = (sum_k w[v,k] A[m, idx[v,k]]) . [v_template[v] + sum_k xt[k,m] bmat_p[k, col(v,c)]; 1] (+ transl[m])
 * bmat_p: the blend matrix of hps_smpl_blend with PANEL-PERMUTED columns, (kp, np) k-major, np = hps_smpl_mesh_fused_np(V):
 *   col(v, c) = (v / 64) * 192 + c * 64 + v % 64  (x, y, z of a 64-vertex panel as three 64-wide column groups),
 *   unused columns zero.  kp here = the K rows that carry data, rounded up to EVEN (SMPL: 10 + 207 -> 218); both operands must be
 *   allocated (and zero) up to the next multiple of 16 rows -- what hps_smpl_pose_prep / hps_smpl_blend call kp (224): the rows
 *   behind kp are not multiplied.  xt, a: from hps_smpl_pose_prep (mp a multiple of 64 covering M).  w_idx / w_val / K / transl /
 *   verts as for hps_smpl_lbs.  Bit-identical to hps_smpl_blend followed by hps_smpl_lbs (same MFMA k order, same
 *   skinning arithmetic: the skipped rows are zeros).  Bound: fp32 MFMA, 2 * kp * 3 V FLOP per mesh; HBM traffic = the 12 V bytes of verts per mesh.
 * Instantiated where it needs no scratch memory beside four workgroups per CU: K = 4 with any num_joints in 1..32, K = 8 and
 * 12 with num_joints = 24; other combinations (e.g. K = 24, dense skinning weights) return HPS_E_UNSUPPORTED -- take the
 * unfused pair, which gives the same bits.
 * Replaces smplx 0.1.26 lbs (reached from models/smpl_official.py:29) steps blend_shapes, pose_feature @ posedirs,
 * W @ A and T @ v_posed. */
int hps_smpl_mesh_fused(const float* xt, const float* bmat_p, const float* v_template, const float* a,
                        const int32_t* w_idx, const float* w_val, int K, int num_joints, const float* transl,
                        float* verts, int M, int V, int kp, int mp, int np, hps_stream_t stream);
/* hps_smpl_mesh_fused with a side output for the joint regression (round 5): pick_slot (V,) int32 gives every vertex its slot in a
 * compact array or -1, and the lane that skins a vertex with a slot also writes it to picked (M, n_picked, 3).  hps_smpl_joints then
 * runs on `picked` (verts = picked, V = n_picked, csr_col = the entries' SLOTS): the same sums over the same values, read from
 * 12 n_picked contiguous bytes per mesh instead of gathered from the whole mesh (SMPL: 276 entries on 198 distinct vertices; the gather
 * was bound by its request rate, 60 us per 6 528 meshes).  Exists for the SMPL configuration (K = 4, 24 joints, kp = 218):
 * HPS_E_UNSUPPORTED otherwise -- use hps_smpl_mesh_fused and gather. */
int hps_smpl_mesh_fused_picks(const float* xt, const float* bmat_p, const float* v_template, const float* a,
                              const int32_t* w_idx, const float* w_val, int K, int num_joints, const float* transl,
                              float* verts, int M, int V, int kp, int mp, int np, const int32_t* pick_slot, float* picked,
                              int n_picked, hps_stream_t stream);
/* The fused mesh kernel for meshes that SHARE their shape (round 6): every mesh of an image carries the image's betas
 * (use_mean_shape = True, the reference's predict default: utils/sampling_utils.py:178-179, predict/...:157-165).  smplx lbs step (1),
 * v_shaped = v_template + blend_shapes(betas), is then formed ONCE per distinct shape (hps_smpl_v_shaped: R = images, not meshes) and
 * the GEMM keeps the 207 pose rows only: xt_pose / bmat_p_pose = the operands of hps_smpl_mesh_fused advanced by num_betas rows, kp =
 * 208 (thirteen whole K chunks: 312 MFMAs per wave against 327), v_posed = v_shaped[row of the mesh] + pose blend -- smplx's own
 * order of the two additions (v_posed = v_shaped + pose_offsets).  The K = 217 form adds template, shape and pose terms in one MFMA
 * chain, so the two agree to rounding, not bit for bit (both within 2e-5 m of the oracle).
 *   v_shaped (R, V, 3); mesh_row (mp,) int32: the row of v_shaped of every mesh (padding meshes: any valid row);
 *   group_rows (mp / 32, 3) int32 per group of 32 consecutive meshes: (row A, row B, split) -- local mesh < split has row A, the
 *   others row B; split = -1: more than two rows in the group, its lanes fetch per mesh through mesh_row.
 * Exists for the SMPL configuration with the side output (K = 4, 24 joints, kp = 208, no translation); HPS_E_UNSUPPORTED otherwise. */
int hps_smpl_mesh_fused_shared_shape(const float* xt_pose, const float* bmat_p_pose, const float* v_shaped,
                                     const int32_t* mesh_row, const int32_t* group_rows, const float* a,
                                     const int32_t* w_idx, const float* w_val, int K, int num_joints, float* verts, int M,
                                     int V, int kp, int mp, int np, const int32_t* pick_slot, float* picked,
                                     int n_picked, hps_stream_t stream);
/* smplx lbs step (1) for R distinct shapes: v_shaped[r, n] = v_template[n] + sum_l betas[r, l] * shape_rows[l * ld + n], n = 3 v + c
 * (shape_rows: the first num_betas rows of the blend matrix of hps_smpl_blend, row pitch ld >= 3 V floats). */
int hps_smpl_v_shaped(const float* betas, int num_betas, const float* shape_rows, int ld, const float* v_template,
                      float* v_shaped, int R, int V, hps_stream_t stream);
/* The shared-shape mesh kernel with its pose blend GEMM on the bf16 matrix pipe AT FP32 ACCURACY ("bf16x3", round 6, opt-in:
 * SMPL.mesh_arith).  Every fp32 operand x is carried as three bf16 pieces x = x1 + x2 + x3 (x1 = RN(x), x2 = RN(x - x1), x3 = RN(x - x1 -
 * x2): 3 x 8 significand bits = fp32's 24, the sum is exact for 2^-110 <= |x| < 3.39e38, where all pieces are normal and finite) and a product is formed as the six piece products of weight >= 2^-16 of
 * the leading one, each exact in the fp32 accumulator; the dropped part is < 2^-23 |a b| -- one fp32 rounding.  Same tile, mapping,
 * skinning arithmetic, side output and HBM traffic as hps_smpl_mesh_fused_shared_shape; the vertices agree with it to rounding (<= 4e-6
 * m asserted; both within 2e-5 m of the oracle and equally close to the float64 twin).  Why it exists: v_mfma_f32_32x32x2_f32 runs at
 * the fp32 vector rate on gfx950 and excludes fp32 VALU work, v_mfma_f32_32x32x16_bf16 is 16 x faster and does not.
 *   hps_smpl_split_bf16x3: src (rows, ld) fp32 k-major (rows behind `rows` up to the next multiple of 16 are written as zeros), its
 *     first `cols` columns in tiles of tile_cols (hps_smpl_split_bf16x3_mesh_tile(): the mesh operand xt of hps_smpl_pose_prep advanced by num_betas rows; 192: bmat_p
 *     advanced likewise) -> dst[tile][16-row chunk][piece][k half][column][8 bf16], hps_smpl_split_bf16x3_bytes(rows, cols) bytes.
 *     The blend matrix is split once per model, the mesh operand once per call.
 *   hps_smpl_mesh_fused_shared_shape_bf16x3: xsplit / bsplit from the above (rows = 207 for SMPL); everything else as for
 *     hps_smpl_mesh_fused_shared_shape.  K = 4, 24 joints; HPS_E_UNSUPPORTED otherwise.  Meshes that do NOT share shapes take the same
 *     entry point with the operands split from row 0 (rows = num_betas + 207: the shape blend inside the GEMM, as in
 *     hps_smpl_mesh_fused), v_shaped = v_template (one row), mesh_row all 0 and every group (0, 0, 32) -- what SMPL.forward does. */
size_t hps_smpl_split_bf16x3_bytes(int rows, int cols);
int hps_smpl_split_bf16x3_mesh_tile(void);   /* tile_cols of the mesh operand = meshes per workgroup tile of the kernel (128) */
int hps_smpl_split_bf16x3(const float* src, int rows, int ld, int cols, int tile_cols, void* dst, hps_stream_t stream);
int hps_smpl_mesh_fused_shared_shape_bf16x3(const void* xsplit, const void* bsplit, const float* v_shaped, const int32_t* mesh_row,
                                            const int32_t* group_rows, const float* a, const int32_t* w_idx, const float* w_val,
                                            int K, int num_joints, float* verts, int M, int V, int rows, int mp,
                                            const int32_t* pick_slot, float* picked, int n_picked, hps_stream_t stream);
/* Column count of bmat_p for a model with V vertices (192 per started panel of 64 vertices). */
int hps_smpl_mesh_fused_np(int V);

/* hps_smpl_joints on the compact side output of hps_smpl_mesh_fused_picks / _shared_shape (picked (M, n_picked, 3), csr_slot = the
 * entries' slots) AND hps_vertex_uncertainty on the call's sample meshes (verts_samples (B, N, V, 3), 8 <= N <= 128) in ONE launch:
 * the two read what the mesh kernel has just written and do not depend on each other (predict/...:112-165: the joints of every mesh,
 * utils/sampling_utils.py:189-190: the per-vertex uncertainty).  Identical bits to the two calls; HPS_E_UNSUPPORTED for other N. */
int hps_joints_and_uncertainty(const float* picked, const float* j_posed, const int32_t* csr_ptr, const int32_t* csr_slot,
                               const float* csr_val, int n_rows, int num_joints, const float* transl, float* joints, int M,
                               int n_picked, const float* verts_samples, float* unc, int B, int N, int V, hps_stream_t stream);

/* Joints: out[m, 0:J] = j_posed[m] ; out[m, J + r] = sum_e csr_val[e] * verts[m, csr_col[e]]
 * for CSR rows r = 0..n_rows-1 (the 21 smplx vertex picks as 1-entry rows, then the extra / cocoplus /
 * h36m regressors of models/smpl_official.py:30-34), each row summed as one chain of fused multiply-adds in entry order.
 * transl optional (M,3). out: (M, J+n_rows, 3). */
int hps_smpl_joints(const float* verts, const float* j_posed, const int32_t* csr_ptr,
                    const int32_t* csr_col, const float* csr_val, int n_rows, int num_joints,
                    const float* transl, float* joints, int M, int V, hps_stream_t stream);

/* Per-vertex uncertainty of utils/sampling_utils.py:189-190, batched over images:
 * verts (B,N,V,3) -> unc (B,V) = mean_s || verts[b,s,v] - mean_s' verts[b,s',v] ||. */
int hps_vertex_uncertainty(const float* verts, float* unc, int B, int N, int V, hps_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Matrix-Fisher sampling  (utils/sampling_utils.py:10-143)
 * ---------------------------------------------------------------------------------------- */

/* One workgroup (1-8 wavefronts, chosen from num_samples alone; results do not depend on it) per (image, joint) call
 * c = image*num_joints + joint, C calls in total.
 * pose_u/pose_v (C,3,3), pose_s (C,3): raw SVD factors; the proper-SVD fix (:104-111), Bingham A,
 * ACG Omega, Gaussian std (:118-124), the 8N-proposal rejection test (:51-61), in-order compaction
 * of the first N accepted proposals (:64-65), quat_to_rotmat (:139) and U_p R V_p^T (:140-141) are
 * all done in the kernel.  r_out is (B, N, num_joints, 3, 3); quat_out optional (B,N,num_joints,4).
 * A round is decided once N proposals have been accepted: the proposals behind the N-th accept (the reference evaluates all
 * 8N, :51-61, and then keeps the first N accepted) are not evaluated -- the same N samples.
 * accepted (C,) int32 receives the number of accepted proposals of the round that was used, counted up to the iteration
 * that reached N (>= N on success); with quat_out != NULL (the Bingham entry point, whose accept_ratio :67 needs it) the
 * round's total over all n_prop proposals.
 * bingham_a: optional (C,4) Bingham parameter used instead of the one derived from pose_s -- the
 * entry point of bingham_sampling_for_matrix_fisher_torch (:10-71), which takes A directly.
 * acg_override: optional (C,8) = [Omega (4) | Gaussian_std (4)] used instead of the values derived from A and b
 * (:42-45 make those only the DEFAULTS of the reference's entry point; m_star is always the caller's).
 *
 * Noise source
 *   eps/w != NULL : proposals come from host-drawn noise, eps (D, n_prop, 4) standard normals and
 *                   w (D, n_prop) uniforms (the reference's torch.randn / torch.rand stream, :51/:60);
 *                   call c reads draw slot draw_idx[c].  A call with fewer than N accepted proposals
 *                   reports accepted[c] < N and its outputs are to be discarded (the caller retries with
 *                   the next draw, as the reference does at :68-69).
 *   eps == NULL   : counter-based Philox4x32-10 in the kernel, keyed by (seed, call_offset + c,
 *                   round, proposal) so results do not depend on how images are sharded over GPUs;
 *                   rounds are redrawn in-kernel until N proposals are accepted; a call that exhausts
 *                   max_rounds (non-finite concentrations) gets NaN outputs and accepted[c] < N.
 *                   seed_dev (optional, device, two uint64: seed, call_offset) overrides the by-value pair: a launch captured in
 *                   a hipGraph reads its key at run time, so every replay draws new samples.
 */
int hps_mf_sample(const float* pose_u, const float* pose_s, const float* pose_v,
                  const float* bingham_a, const float* acg_override, int C, int num_joints, int num_samples, int n_prop,
                  float b, float m_star,
                  const float* eps, const float* w, const int32_t* draw_idx, uint64_t seed,
                  int64_t call_offset, const uint64_t* seed_dev, int max_rounds, float* r_out, float* quat_out,
                  int32_t* accepted, hps_stream_t stream);

/* Inputs of ONE flattened SMPL call over the M = B (N + 2) meshes [mode (B) | T-pose (B) | samples (B N)] of the inference
 * core (predict/predict_poseMF_shapeGaussian_net.py:112-115 mode mesh, :136 T-pose mesh, utils/sampling_utils.py:178-185 sample
 * meshes): body (M, J-1, 3, 3) rows [0, 2B) (mode rotations, then identities; rows [2B, M) are where hps_mf_sample wrote its
 * samples), glob_all (M, 3, 3) (glob_rotmats of the image; identity for the T-pose), betas_all (M, nb) (the shape mean of the
 * image, or betas_samples (B, N, nb) for the sample meshes if non-NULL: use_mean_shape = False, :180-181). */
int hps_infer_assemble(const float* mode, const float* glob_rotmats, const float* loc, const float* betas_samples,
                       float* body, float* glob_all, float* betas_all, int B, int N, int num_body_joints,
                       int num_betas, hps_stream_t stream);

/* utils/rigid_transform_utils.py:113-133 */
int hps_quat_to_rotmat(const float* quat, float* rotmat, int n, hps_stream_t stream);
/* utils/rigid_transform_utils.py:80-94 (cross product along dim 1 for every batch size) */
int hps_rot6d_to_rotmat(const float* x6, float* rotmat, int n, hps_stream_t stream);
/* smplx.lbs.batch_rodrigues (reference import: predict/predict_poseMF_shapeGaussian_net.py:7) */
int hps_batch_rodrigues(const float* aa, float* rotmat, int n, hps_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Distribution-prediction head  (models/poseMF_shapeGaussian_net.py:95-162)
 * ---------------------------------------------------------------------------------------- */

/* out[b, n] = act( sum_k x[b, k] * wt[k, n] + bias[n] (+ addend[n]) ),  wt = weight^T (K,N).
 * x rows have stride ldx, out rows stride ldo (lets the trunk write straight into the fc_embed
 * input buffer).  addend optional (init_glob / init_cam, :106-107). */
int hps_linear(const float* x, int ldx, const float* wt, const float* bias, const float* addend,
               float* out, int ldo, int B, int K, int N, int act, hps_stream_t stream);

/* The FC trunk (models/poseMF_shapeGaussian_net.py:95-110) as three launches and nothing else:
 *   x      = ELU(fc1(feats))                                        x_ws (B, hidden), scratch
 *   sgc    = [fc_shape | fc_glob | fc_cam](x) + sgc_add             sgc_out (B, 2 num_shape + num_glob + num_cam): shape_params,
 *            glob + init_glob, cam + init_cam -- the tail of fc_embed's input; the same launch also writes
 *            shape_loc = mean, shape_scale = exp(log std) (the Normal of :100-101), glob, cam, each contiguous
 *   embed  = ELU(fc_embed(cat[feats, sgc]))                         the concatenation (:108) is never materialised
 * Weights pre-transposed (K, N) like hps_linear; sgc_wt / sgc_b / sgc_add = the three layers stacked (sgc_add: zeros, init_glob,
 * init_cam).  feats rows have stride ldf. */
int hps_head_trunk(const float* feats, int ldf, const float* fc1_wt, const float* fc1_b, const float* sgc_wt,
                   const float* sgc_b, const float* sgc_add, const float* embed_wt, const float* embed_b, float* x_ws,
                   float* sgc_out, float* embed, float* shape_loc, float* shape_scale, float* glob, float* cam, int B,
                   int num_feats, int hidden, int num_shape, int num_glob, int num_cam, int embed_dim, hps_stream_t stream);

/* One kinematic level of the joint loop (:121-135): for every joint g in the level
 *   in  = cat[embed(b), U_proper[b, anc].flat, S_proper[b, anc].flat, mode[b, anc].flat]
 *   F   = W2 . ELU(W1 . in + b1) + b2 + delta_i_weight * I
 * joint_ids (n_level,) int32; anc_ptr (23+1,) / anc_idx: CSR list of ancestors (nearest first);
 * w1t_ptrs[joint] -> (in_dim, hidden) = fc_pose[j].0.weight^T, b1_ptrs -> (hidden,),
 * w2_ptrs -> (9, hidden), b2_ptrs -> (9,)  (device arrays of device pointers, indexed by joint id).
 * u_proper/mode (B,23,9), s_proper (B,23,3) hold the ancestors' results; pose_f (B,23,9) output;
 * f_level: optional compact copy (B, n_level, 9) of the level's F matrices (what goes to the host SVD). */
int hps_head_joint_level(const float* embed, int embed_dim, int hidden, const int32_t* joint_ids,
                         int n_level, const int32_t* anc_ptr, const int32_t* anc_idx,
                         const float* const* w1t_ptrs, const float* const* b1_ptrs,
                         const float* const* w2_ptrs, const float* const* b2_ptrs,
                         const float* u_proper, const float* s_proper, const float* mode,
                         float delta_i_weight, float* pose_f, float* f_level, int B,
                         int num_body_joints, hps_stream_t stream);

/* hps_head_joint_level with the level's 3x3 SVDs (models/poseMF_shapeGaussian_net.py:137), the proper-SVD fix and the mode
 * (:139-152) done INSIDE the kernel: pose_u / pose_s / pose_v and u_proper / s_proper / mode rows of the level's joints are
 * written, no host round trip.  The SVD follows LAPACK's sgesdd step by step (csrc/svd3_gesdd.h) with the roundings of the
 * host's MKL (svd_flavor: HPS_SVD_ROUNDING_*), so that the factors -- and with them the signs the reference's torch.svd
 * gives the singular vectors, which are inputs of the child joints (:126-130) -- are reproduced bit for bit. */
int hps_head_joint_level_svd(const float* embed, int embed_dim, int hidden, const int32_t* joint_ids,
                             int n_level, const int32_t* anc_ptr, const int32_t* anc_idx,
                             const float* const* w1t_ptrs, const float* const* b1_ptrs,
                             const float* const* w2_ptrs, const float* const* b2_ptrs, float* u_proper,
                             float* s_proper, float* mode, float delta_i_weight, float* pose_f, float* pose_u,
                             float* pose_s, float* pose_v, int B, int num_body_joints, int svd_flavor,
                             hps_stream_t stream);

/* The same device SVD for n row-major 3x3 matrices: f (n,9) -> usv (n,21) packed [U (9) | S (3) | V (9)]; replaces
 * torch.svd(F.cpu()) (:137).  Non-finite or non-converging input (LAPACK: INFO != 0) gives NaN factors. */
int hps_svd3_packed(const float* f, float* usv, int n, int svd_flavor, hps_stream_t stream);

/* HOST function: the algorithm of hps_svd3_packed compiled for the host (single thread) -- for measuring / testing its
 * agreement with LAPACK without a GPU.  f_host (n,9) -> usv_host (n,21). */
int hps_host_svd3_emulated(const float* f_host, float* usv_host, int n, int svd_flavor);

/* HOST function: the rounding flavour (HPS_SVD_ROUNDING_*) whose factors are bit-identical to those of the LAPACK sgesdd_
 * this process is bound to (hps_host_bind_lapack) on 4 096 fixed test matrices, or -1 if neither is.  The shipped binding
 * calls it once and runs the device SVD in that flavour, so that svd_mode device and svd_mode host give the same bits. */
int hps_host_svd_flavor(void);

/* HOST function (no device work): SVD of n row-major 3x3 matrices through the LAPACK sgesdd_ exported by the
 * process's libtorch_cpu.so -- the routine behind the reference's torch.svd(F.cpu()) (:137), so factors and
 * column signs are bit-identical -- spread over num_threads threads.  f_host (n,9) -> usv_host (n,21) packed
 * [U (9) | S (3) | V (9)].  HPS_E_UNSUPPORTED if sgesdd_ cannot be resolved. */
int hps_host_svd3_packed(const float* f_host, float* usv_host, int n, int num_threads);
/* HOST: resolve sgesdd_ from the given shared library (the binding passes torch's libtorch_cpu.so). */
int hps_host_bind_lapack(const char* library_path);

/* Proper-SVD fix and mode (:139-152) for the joints of one level.  usv_level (B, n_level, 21) holds the SVD
 * factors of the level's F matrices packed as [U (9) | S (3) | V (9)] per matrix; they are scattered into
 * pose_u / pose_s / pose_v (B,23,..) and u_proper, s_proper, mode = U_p V_p^T are written. */
int hps_head_svd_finish(const float* usv_level, const int32_t* joint_ids, int n_level,
                        float* pose_u, float* pose_s, float* pose_v, float* u_proper,
                        float* s_proper, float* mode, int B, int num_body_joints,
                        hps_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ResNet-18 encoder  (models/resnet.py:202-217) -- NHWC activations
 * ---------------------------------------------------------------------------------------- */

/* Halo-padded generation of the convolution (csrc/conv_pad.hip), same arithmetic and summation order as
 * hps_conv2d_bn_act_v3.  x is (B, H + 2 ipad, W + 2 ipad, Cin) NHWC with a ZERO halo of ipad >= pad pixels, so every
 * filter tap is in bounds; y (and residual) are frames (B, Ho + 2 opad, Wo + 2 opad, Cout) of which only the interior
 * is written (the owner zeroes the halo once).  wn: n-major filter (Cout, KH*KW*Cin), Cin % 32 == 0; or, row_mode != 0
 * (the 18-channel 7x7 stem, models/resnet.py:150, :203): (Cout, KH * ceil32(KW*Cin)), one filter row = KW*Cin contiguous
 * NHWC floats taken as a single tap, zero filled tail.  variant: 0 automatic, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x64
 * workgroup tiles, 5 = 64x64 with a four-stage K loop (chunks fetched three ahead: the form for one or a few images, where a CU
 * holds one workgroup and nothing else covers the fetch latency).  ksplit > 1 ((KH*KW*Cin/32) divisible by ksplit): K in ksplit
 * slices, on the 128-row tiles unless variant is 3 or 5; splitk_ws (ksplit, B*Ho*Wo, Cout) floats; the slices are added in slice
 * order by a second kernel that also applies BN / residual / ReLU (deterministic, no atomics).  The tile shape never changes an
 * output's summation order.  Lane offsets are 32-bit: tensors < 4 GiB. */
int hps_conv2d_bn_act_pad(const float* x, const float* wn, const float* scale, const float* shift,
                          const float* residual, float* y, int B, int H, int W, int ipad, int Cin, int Cout,
                          int KH, int KW, int stride, int pad, int opad, int relu, int row_mode, int variant,
                          int ksplit, float* splitk_ws, hps_stream_t stream);

/* BasicBlock entry with a down-sample branch (models/resnet.py:62-78 with :71-72 `identity = self.downsample(x)`, built at
 * :184-188): the block's k x k / stride / pad = k/2 convolution + BatchNorm (+ ReLU) -> y AND its 1x1 / stride / 0 down-sample
 * convolution + BatchNorm -> y_down, both from the same input frame, in ONE launch (extra workgroups walk the main window's centre
 * tap with the down-sample's filter).  wn_down: (Cout, Cin) n-major; y_down: a frame of y's geometry.  Every output has the bits of
 * hps_conv2d_bn_act_pad called once per convolution (same chunks, same order; tile shapes do not change a summation order).
 * variant 1, 2, 3, 5 or 0 (automatic) as above; ksplit applies to the main convolution only.  Cin % 32 == 0. */
int hps_conv2d_bn_act_pad_down(const float* x, const float* wn, const float* scale, const float* shift, float* y,
                               const float* wn_down, const float* scale_down, const float* shift_down, float* y_down,
                               int B, int H, int W, int ipad, int Cin, int Cout, int KH, int KW, int stride, int pad,
                               int opad, int relu, int variant, int ksplit, float* splitk_ws, hps_stream_t stream);

/* 3x3 / stride 1 / pad 1 convolution + BatchNorm (+ residual) (+ ReLU) by Winograd F(2x2, 3x3) on the fp32 MFMA pipe
 * (csrc/conv_wino.hip): 16 multiplications per 2x2 output tile and (cin, cout) pair instead of 36 -- the stride-1 3x3
 * layers of the BasicBlocks (models/resnet.py:62-78).  x / y / residual are halo-padded NHWC frames as for
 * hps_conv2d_bn_act_pad (ipad >= 1).  u: the transformed filters U = G g G^T, prepared once by the host in the layout
 *   u[chunk = cin / 8][cout tile = cout / 64][position p = 4 a + b][k-quad = (cin % 8) / 4][cout % 64][cin % 4]
 * with G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1].  Requirements: Cin % 8 == 0, Cout % 64 == 0 and either
 * H % 16 == 0, W % 16 == 0 (items of 8 x 8 tiles of one image) or H == W == 8 (layer4: items of four images; K is cut
 * into slices whose partial sums go through splitk_ws -- hps_conv3x3_winograd_workspace(B, H, W, Cin, Cout) bytes, 0 for
 * the first geometry, where splitk_ws may be NULL -- and are added in slice order by a second kernel).  Results equal the
 * direct convolution up to fp32 rounding of a different summation order; the order depends on the layer only, never on
 * the batch size. */
int hps_conv3x3_winograd(const float* x, const float* u, const float* scale, const float* shift,
                         const float* residual, float* y, int B, int H, int W, int ipad, int Cin, int Cout,
                         int opad, int relu, float* splitk_ws, hps_stream_t stream);
size_t hps_conv3x3_winograd_workspace(int B, int H, int W, int Cin, int Cout);

/* The stem -- conv1 7x7 / stride 2 / pad 3 on the 18-channel proxy representation + bn1 + relu (models/resnet.py:147-150,
 * :203-206) -- in Winograd form (csrc/stem_wino.hip): the stride-2 correlation is the sum of four stride-1 correlations on the
 * even / odd sub-lattices of the input (4x4, 4x3, 3x4 and 3x3 taps), each computed as F(2x2, r x s): 81 multiplications per 2x2
 * output tile and (cin, cout) pair instead of 196.  Same result as the direct sum up to fp32 rounding (both are within 1e-6 of the
 * output scale of the fp64 sum).  H and W multiples of 32, Cin = 18, Cout = 64.
 *   hps_stem_phase_split: x (B,18,H,W) NCHW -> frames[b][2 ry + rx][i][j][18] = x[b][:, 2 i + ry - 3, 2 j + rx - 3], four
 *     (H/2 + 4) x (W/2 + 4)-pixel frames per image whose out-of-image pixels the OWNER zeroes once (only in-image pixels are ever
 *     written); the channels of a pixel are stored in the order 0 2 1 3 | 4 6 5 7 | 8 10 9 11 | 12 14 13 15 | 16 17, and two pixels
 *     are followed by two floats of padding (a frame row is (W/2 + 4) / 2 pixel pairs of 38 floats: LDS bank spread).  The buffer
 *     holds hps_stem_phase_frames_bytes(B, H, W) bytes (the frames plus the slack the last window's DMA over-reads).
 *   hps_stem_winograd: frames -> y, the interior of the (B, H/2 + 2 opad, W/2 + 2 opad, 64) NHWC frame.  u: the transformed
 *     filters of the 81 positions, position-major in the row order (phase 2 ry + rx; row i; column j), 1152 floats each:
 *       u[position][co / 32][ k-block: (c % 16) / 8 ][ c % 2 ][co % 32][ (c % 8) / 2 ]   for c < 16   (2 x 2 x 32 x 4 floats)
 *       u[position][co / 32][512 + (c - 16) * 32 + co % 32]                              for c = 16, 17
 *     U = G_y g_{ry,rx} G_x^T with g_{ry,rx}[a][b] = w[co][c][2a + ry][2b + rx] and
 *     G (4 taps) = [1/2 0 0 0; -1/2 -1/2 -1/2 -1/2; -1/6 1/6 -1/6 1/6; 1/6 1/3 2/3 4/3; 0 0 0 1],
 *     G (3 taps) = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]; 81 * 1152 floats plus 256 floats of readable slack. */
size_t hps_stem_phase_frames_bytes(int B, int H, int W);
int hps_stem_phase_split(const float* x, float* frames, int B, int C, int H, int W, hps_stream_t stream);
/* The proxy representation (predict/predict_poseMF_shapeGaussian_net.py:93-100; heat-maps: utils/label_conversions.py:105-124) written
 * straight into the phase frames: exactly what hps_proxy_rep (edge -> channel 0, visibility-masked Gaussians of the K = 17 joints ->
 * channels 1..17) followed by hps_stem_phase_split leaves in `frames`, bit for bit, without the (B,18,H,W) tensor in between.
 * edge: (B,H,W) plane or NULL (zeros); joints2d (B,17,2) (u = column, v = row); visib (B,17) or NULL.  For callers that build the
 * network input on the device anyway (the from-RGB pipeline); the NCHW entry stays the contract of the reference's call surface. */
int hps_proxy_rep_phase_frames(const float* edge, const float* joints2d, const float* visib, float* frames, int B, int K, int H,
                               int W, float std, hps_stream_t stream);
int hps_stem_winograd(const float* frames, const float* u, const float* scale, const float* shift, float* y, int B, int H,
                      int W, int opad, int relu, hps_stream_t stream);
/* The stem AND the 3x3 / 2 / pad 1 max pool behind it (models/resnet.py:150, :206: self.maxpool) in one call: frames -> the interior of
 * the (B, H/4 + 2 opad, W/4 + 2 opad, 64) NHWC frame of the POOLED map; the stem's full-resolution output (which nothing else reads) is
 * never written.  Every pooled pixel is the maximum of the same nine stem outputs hps_maxpool3x3s2_pad takes it over (formed tile by tile
 * in the stem kernel's epilogue; the pixels of an 8 x 8 block of pooled pixels that also see a neighbouring block are completed by a
 * small second kernel from `side`): identical values.  side: scratch of hps_stem_pool_side_bytes(B, H, W) bytes, written and read by
 * this call only. */
size_t hps_stem_pool_side_bytes(int B, int H, int W);
int hps_stem_winograd_pooled(const float* frames, const float* u, const float* scale, const float* shift, float* pooled, float* side,
                             int B, int H, int W, int opad, int relu, hps_stream_t stream);
/* The same from the network input itself: x (B,18,H,W) NCHW fp32, exactly what models/resnet.py:202-206 receives.  The kernel gathers its
 * phase windows from x (global loads, out-of-image pixels stay zero) into the LDS layout the frames would have given it:
 * hps_stem_phase_split and the four frames per image are not needed.  Identical values to hps_stem_phase_split + hps_stem_winograd_pooled. */
int hps_stem_winograd_pooled_nchw(const float* x, const float* u, const float* scale, const float* shift, float* pooled, float* side,
                                  int B, int H, int W, int opad, int relu, hps_stream_t stream);

/* One launch of the encoder's operation list (hps_encoder_run).  kind: HPS_ENC_RELAYOUT = hps_nchw_to_padded_nhwc
 * (x, y, B, Cin = C, H, W, opad = P), HPS_ENC_CONV = hps_conv2d_bn_act_pad (all fields), HPS_ENC_MAXPOOL =
 * hps_maxpool3x3s2_pad (x, y, B, H, W, Cin = C, opad), HPS_ENC_AVGPOOL = hps_global_avgpool_pad (x, y, B, H, W,
 * Cin = C, ipad = P), HPS_ENC_CONV_WINOGRAD = hps_conv3x3_winograd (x, w = u, scale, shift, residual, y, B, H, W, ipad, Cin,
 * Cout, opad, relu, splitk_ws), HPS_ENC_STEM_SPLIT = hps_stem_phase_split (x, y = frames, B, Cin = C, H, W),
 * HPS_ENC_STEM_WINOGRAD = hps_stem_winograd (x = frames, w = u, scale, shift, y, B, H, W, opad, relu), HPS_ENC_STEM_WINOGRAD_POOLED =
 * hps_stem_winograd_pooled (x = frames, w = u, scale, shift, y = pooled frame, splitk_ws = side, B, H, W, opad, relu),
 * HPS_ENC_STEM_WINOGRAD_POOLED_NCHW = hps_stem_winograd_pooled_nchw (the same fields with x = the NCHW input). */
enum { HPS_ENC_RELAYOUT = 0, HPS_ENC_CONV = 1, HPS_ENC_MAXPOOL = 2, HPS_ENC_AVGPOOL = 3, HPS_ENC_CONV_WINOGRAD = 4,
       HPS_ENC_STEM_SPLIT = 5, HPS_ENC_STEM_WINOGRAD = 6,
       HPS_ENC_RELAYOUT_GENERIC = 7 /* hps_nchw_to_padded_nhwc_generic (x, y, B, Cin = C, Cout = CP, H, W, KW = WF, opad = P) */,
       HPS_ENC_STEM_WINOGRAD_POOLED = 8, HPS_ENC_STEM_WINOGRAD_POOLED_NCHW = 9,
       HPS_ENC_CONV_DOWN = 10 /* hps_conv2d_bn_act_pad_down: the HPS_ENC_CONV fields + w_down / scale_down / shift_down / y_down */ };
typedef struct hps_enc_op {
    int kind;
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    float* splitk_ws;
    int B, H, W, ipad, Cin, Cout, KH, KW, stride, pad, opad, relu, row_mode, variant, ksplit;
    const float* w_down;       /* HPS_ENC_CONV_DOWN only (NULL otherwise) */
    const float* scale_down;
    const float* shift_down;
    float* y_down;
} hps_enc_op;

/* sizeof(hps_enc_op) as compiled into the library: lets a hand-written mirror of the struct check itself. */
int hps_sizeof_enc_op(void);

/* models/resnet.py:202-217 in one call: issues ops[0..n_ops) in order on `stream` (same launches as the individual
 * entry points; exists because issuing ~27 launches through a scripting-language FFI costs more host time than the
 * GPU needs to run them). */
int hps_encoder_run(const hps_enc_op* ops, int n_ops, hps_stream_t stream);

/* models/poseMF_shapeGaussian_net.py:121-160 in one call, one kinematic level after the other.
 *   svd_mode HPS_SVD_DEVICE / HPS_SVD_DEVICE_FMA (rounding flavour reference / fused): hps_head_joint_level_svd per level --
 *            8 stream-ordered launches for the SMPL tree, no synchronisation, capturable in a hipGraph; the staging buffers
 *            may be NULL.
 *   svd_mode HPS_SVD_HOST (parity mode: the very LAPACK routine of the reference): hps_head_joint_level -> D2H of the level's F
 *            matrices -> stream synchronise -> hps_host_svd3_packed (:137) -> H2D -> hps_head_svd_finish.
 * level_joints: DEVICE int32 array, the levels' joint ids concatenated; level_sizes_host: HOST
 * array of n_levels sizes; f_level_dev (B*max_level*9) / usv_level_dev (B*max_level*21): device scratch;
 * f_host_pinned / usv_host_pinned: page-locked host staging of the same sizes.  Blocks the calling thread (it waits
 * for each level's matrices); other streams keep running. */
int hps_head_pose_levels(const float* embed, int embed_dim, int hidden, const int32_t* level_joints,
                         const int32_t* level_sizes_host, int n_levels, const int32_t* anc_ptr,
                         const int32_t* anc_idx, const float* const* w1t_ptrs, const float* const* b1_ptrs,
                         const float* const* w2_ptrs, const float* const* b2_ptrs, float* u_proper,
                         float* s_proper, float* mode, float delta_i_weight, float* pose_f, float* pose_u,
                         float* pose_s, float* pose_v, float* f_level_dev, float* usv_level_dev,
                         float* f_host_pinned, float* usv_host_pinned, int B, int num_body_joints,
                         int svd_threads, int svd_mode, hps_stream_t stream);

/* (B,C,H,W) -> interior of the (B, H + 2P, W + 2P, C) NHWC frame; C in {4, 18, 64}
 * (predict/predict_poseMF_shapeGaussian_net.py:103 hands the net an NCHW proxy representation). */
int hps_nchw_to_padded_nhwc(const float* x, float* y, int B, int C, int H, int W, int P, hps_stream_t stream);

/* The same for ANY channel count and width (models/resnet.py:127-176 accepts any in_channels / image size): channel c < C of
 * pixel (h, w) -> interior of a (B, H + 2P, WF + 2P, CP) frame, CP >= C, WF >= W; channels C..CP-1, columns W..WF-1 and the
 * halo are left as the owner zeroed them.  The encoder uses CP = C rounded up to 4 and WF = W rounded up to even, which gives
 * the row-mode stem of hps_conv2d_bn_act_pad its 16-byte aligned windows for every shape (zero channels / a zero column are
 * exactly the convolution's zero padding). */
int hps_nchw_to_padded_nhwc_generic(const float* x, float* y, int B, int C, int CP, int H, int W, int WF, int P,
                                    hps_stream_t stream);

/* nn.MaxPool2d(3, 2, 1) (models/resnet.py:152, :207): x (B,H,W,C) plain NHWC -> interior of the
 * (B, Ho + 2 opad, Wo + 2 opad, C) frame. */
int hps_maxpool3x3s2_pad(const float* x, float* y, int B, int H, int W, int C, int opad, hps_stream_t stream);

/* AdaptiveAvgPool2d((1,1)) + flatten (models/resnet.py:214-215) over the interior of a (B, H + 2P, W + 2P, C) frame. */
int hps_global_avgpool_pad(const float* x, float* y, int B, int H, int W, int C, int P, hps_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Proxy-representation front end  
 * ---------------------------------------------------------------------------------------- */

/* models/canny_edge_detector.py:104-166: per-channel separable Gaussian blur (zero padded), Sobel gradients averaged
 * over channels, magnitude, orientation binned to 45 degrees, threshold, and (nms != 0) non-max suppression.
 * img (B,C,H,W); gauss_taps_host: HOST array of gauss_size (= 5) normalised taps; outputs blurred (B,C,H,W) and
 * grad_mag, grad_ori, thr_mag, thin, thr_thin (B,1,H,W) -- the entries of the reference's output dict. */
int hps_canny_edges(const float* img, const float* gauss_taps_host, int gauss_size, float* blurred,
                    float* grad_mag, float* grad_ori, float* thr_mag, float* thin, float* thr_thin,
                    int B, int C, int H, int W, float threshold, int nms, hps_stream_t stream);

/* The edge map ALONE -- what predict/predict_poseMF_shapeGaussian_net.py:92-93 takes out of the detector's dict
 * ('thresholded_thin_edges' if nms != 0, else 'thresholded_grad_magnitude'): same kernel and arithmetic as hps_canny_edges,
 * none of the other five outputs written.  Image b's map goes to edge_out + b * edge_batch_stride (floats), so that it can be
 * channel 0 of a (B, K+1, H, W) proxy representation directly (edge_batch_stride = (K+1) H W; hps_proxy_rep with edge = NULL
 * then fills channels 1..K only): 4 planes of traffic per image instead of 11 + 2. */
int hps_canny_edge_map(const float* img, const float* gauss_taps_host, int gauss_size, float* edge_out,
                       int64_t edge_batch_stride, int B, int C, int H, int W, float threshold, int nms, hps_stream_t stream);

/* predict/predict_poseMF_shapeGaussian_net.py:93-100 with utils/label_conversions.py:105-124: out (B,K+1,H,W),
 * channel 0 = edge (B,1,H,W) -- or left as it is when edge is NULL (hps_canny_edge_map wrote it in place) --,
 * channel 1+k = visib[b,k] * exp(-((row - v)/std)^2/2 - ((col - u)/std)^2/2) for
 * joints2d (B,K,2) = (u, v); visib (B,K) float 0/1 or NULL; K <= 32. */
int hps_proxy_rep(const float* edge, const float* joints2d, const float* visib, float* out, int B,
                  int K, int H, int W, float std, hps_stream_t stream);

/* utils/label_conversions.py:127-155: heatmaps (BK, H, W) -> joints2d (BK,2) = (x, y) of the maximum (first index on
 * ties), visib (BK,) = 1.0 if max > eps else 0.0 (then joints2d = -1). */
int hps_heatmaps_to_joints2d(const float* heatmaps, float* joints2d, float* visib, int BK, int H, int W,
                             float eps, hps_stream_t stream);

/* utils/sampling_utils.py:210-229: err[s] = max over visible input joints k of the pixel distance
 * between in_j2d[k] and the weak-perspective projection of joints[s, coco_map[k]] flipped 180 degrees about x.
 * joints (N, n_joints_all, 3); in_j2d (K,2); in_vis (K,); cam (3,) = (scale, tx, ty). */
int hps_sample_joints2d_error(const float* joints, const int32_t* coco_map, int n_joints_all,
                              const float* in_j2d, const float* in_vis, const float* cam, float img_wh,
                              float* err, int N, int K, hps_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation metrics  
 * ---------------------------------------------------------------------------------------- */

/* Per point-set L2 error sums (metrics/eval_metrics_tracker.py:89-269).  pred (S,P,3); prediction s is compared with
 * target set s / group of target (ceil(S/group),P,3).  mode 0: raw; 1: after scale_and_translation_transform_batch
 * (utils/eval_utils.py:70-89); 2: after the Procrustes similarity transform (utils/eval_utils.py:11-59).
 * err_sum (S,) double = sum_i || transform(pred_i) - target_i ||.  Workspaces: stats_ws (S,17) double, xf_ws (S,12)
 * float (receives the applied 3x4 transforms).  transformed: optional (S,P,3) transformed predictions. */
int hps_pointset_errors(const float* pred, const float* target, int S, int group, int P, int mode,
                        double* stats_ws, float* xf_ws, double* err_sum, float* transformed,
                        hps_stream_t stream);

/* Result checksums (bench.py / the sharding-invariance tests; no reference counterpart): out[0] = first, out[1 + j] =
 * sum of xs[j][0 .. ns[j]) (of |x| where take_abs[j]) accumulated in float64 in a fixed order, for count <= 4 float tensors
 * in two launches.  xs / ns / take_abs are HOST arrays; partial_ws: 4 * 128 doubles; out: 1 + count doubles (device).
 * accumulate (optional, 1 + count doubles, device): a running total, accumulate[k] += out[k] in the second launch (a loop's
 * accumulator without a launch of its own). */
int hps_sums_f64(const float* const* xs, const int64_t* ns, const int32_t* take_abs, int count, double first,
                 double* partial_ws, double* out, double* accumulate, hps_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* HPS_H_ */

/*
 * hps_dev.h -- development / cross-check entry points of libhps_dev.so (the library built with -DHPS_DEV_BUILD).
 *
 * NOT part of the product ABI: libhps.so exports none of these.  The dev library contains the product code plus
 *   - the earlier, un-padded convolution generations (csrc/conv.hip: v1 / v2 / v3) and their relayout / pool kernels,
 *     kept as a bit-level cross-check of the halo-padded product kernels (tests/devlib.py: plain_conv, plain_forward);
 *   - alternate kernel variants (stationary-A blend GEMM, LDS-resident uncertainty kernels, LBS launch geometries);
 *   - process-global tuning switches (hps_dev_*) and ablation launches for profiling.
 * Only tests/ and tests/dev/ load it (_capi.dev_library(); the ctypes prototypes of these entry points live in tests/devlib.py).
 */
#ifndef HPS_DEV_H_
#define HPS_DEV_H_

#include "hps.h"

#ifdef __cplusplus
extern "C" {
#endif
/* The libraries are built with -fvisibility=hidden: what is declared between this push and the pop at the end of the header is
 * the complete dynamic symbol table of the shared object (tests/test_capi_symbols.py compares it with `nm -D`). */
#pragma GCC visibility push(default)

/* Kinematic depth levels the single-launch head experiment (hps_dev_head_pose_levels_fused) handles (the body tree has 8), and the
 * hps_query_workspace item -- of the DEV build only -- that sizes its counter workspace: d0 = B -> bytes (ZERO before its first use). */
#define HPS_HEAD_MAX_LEVELS 32
#define HPS_DEV_WS_HEAD_SYNC 7

/* Development / tuning entry: hps_smpl_lbs with an explicit kernel variant (0..4: meshes per barrier G and
 * vertices per lane VPT = (4,1) (8,1) (4,2) (2,2) (2,1)) and resident-workgroup target. Same results. */
int hps_dev_lbs_variant(const float* v_posed, int ld_vposed, const float* a, const int32_t* w_idx,
                        const float* w_val, int K, int num_joints, const float* transl, float* verts,
                        int M, int V, int variant, int target_blocks, hps_stream_t stream);

/* (B,C,H,W) -> (B,H,W,Cp) with channels zero-padded to Cp. */
int hps_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, int Cp,
                     hps_stream_t stream);

/* Implicit-GEMM convolution on fp32 MFMA with fused eval-mode BatchNorm, residual add and ReLU:
 *   y[b,ho,wo,co] = act( scale[co] * sum_{kh,kw,ci} x[b, ho*s-p+kh, wo*s-p+kw, ci] * wk[(kh,kw,ci), co]
 *                        + shift[co] (+ residual[b,ho,wo,co]) )
 * x (B,H,W,Cin) NHWC, wk (ceil16(KH*KW*Cin), Cout) k-major filter (rows zero-padded to a multiple of
 * 16), scale/shift (Cout,) from BN running stats
 * (models/resnet.py:62-78, :202-206).  Cin % 4 == 0 (pad), Cout % 64 == 0. */
int hps_conv2d_bn_act(const float* x, const float* wk, const float* scale, const float* shift,
                      const float* residual, float* y, int B, int H, int W, int Cin, int Cout,
                      int KH, int KW, int stride, int pad, int relu, hps_stream_t stream);

/* Same convolution for Cin % 32 == 0 (all of ResNet-18 after the stem), faster kernel: K-chunks of 32 inside one
 * filter tap, 128-bit LDS fragment traffic.  wn: filter stored n-major (Cout, KH*KW*Cin) = weight.permute(0,2,3,1).
 * variant: 0 automatic tile choice, 1 = 128x128, 2 = 128x64, 3 = 64x64 workgroup tiles (tuning). */
int hps_conv2d_bn_act_v2(const float* x, const float* wn, const float* scale, const float* shift,
                         const float* residual, float* y, int B, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, int relu, int variant,
                         hps_stream_t stream);

/* As hps_conv2d_bn_act_v2 but the LDS tiles are filled by direct global->LDS DMA (global_load_lds_dwordx4) with a
 * source-side XOR swizzle.  zeros: device buffer of >= 64 zero bytes (source of out-of-image taps).
 * variant: 0 automatic, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x64 workgroup tiles (2x / 3x: tuning ablations).
 * ksplit > 1 (Cout % 128 == 0, KH*KW*Cin/32 divisible by ksplit): split-K over ksplit slices on 128x128 tiles for
 * layers with too few output tiles to fill 256 CUs; splitk_ws: (ksplit, B*Ho*Wo, Cout) floats of workspace; the slices
 * are summed in slice order by a second kernel that also applies BN / residual / ReLU (deterministic, no atomics). */
int hps_conv2d_bn_act_v3(const float* x, const float* wn, const float* zeros, const float* scale,
                         const float* shift, const float* residual, float* y, int B, int H, int W,
                         int Cin, int Cout, int KH, int KW, int stride, int pad, int relu, int variant,
                         int ksplit, float* splitk_ws, hps_stream_t stream);

/* First generations of hps_smpl_pose_prep (8 meshes per workgroup, operand rows written 4 bytes at a time) and hps_smpl_joints (one
 * workgroup per mesh, dependent load chains per row): same arguments, same bits -- the cross-check of the round-5 kernels
 * (tests/test_gpu_smpl.py). */
int hps_dev_smpl_pose_prep_v1(const float* glob, const float* body, int is_rotmat, const float* betas, int num_betas,
                              const float* j_template, const float* j_shapedirs, const int32_t* parents, const int32_t* depth,
                              int num_joints, float* xt, int kp, int mp, float* a, float* j_posed, float* rot_out, int M,
                              hps_stream_t stream);
int hps_dev_smpl_joints_v1(const float* verts, const float* j_posed, const int32_t* csr_ptr, const int32_t* csr_col,
                           const float* csr_val, int n_rows, int num_joints, const float* transl, float* joints, int M, int V,
                           hps_stream_t stream);

/* EXPERIMENT (measured, not adopted: csrc/head.hip).  The whole joint loop (models/poseMF_shapeGaussian_net.py:121-160) in ONE launch, device SVD: what hps_head_pose_levels issues as
 * one hps_head_joint_level_svd launch per kinematic level (level_joints: the levels' joint ids back to back, level_sizes_host: HOST
 * array of the n_levels <= HPS_HEAD_MAX_LEVELS level sizes), with the same per-joint code -- identical bits.  Workgroup (slot, tile)
 * walks the levels for its four images; the <= widest-level workgroups of a tile meet between levels at counters in sync_ws
 * (hps_query_workspace(HPS_DEV_WS_HEAD_SYNC, B) bytes -- dev build only; ZERO before the first use, the kernel leaves it zero).  For latency-bound calls
 * on one or a few images: eight dispatches become one, and the branchy LAPACK-faithful SVD runs from a warm instruction cache from
 * the second level on.  Returns HPS_E_UNSUPPORTED when widest level x ceil(B / 4) workgroups exceed the device's CU count (the
 * waiting workgroups must all be schedulable at once): use hps_head_pose_levels then. */
int hps_dev_head_pose_levels_fused(const float* embed, int embed_dim, int hidden, const int32_t* level_joints,
                               const int32_t* level_sizes_host, int n_levels, const int32_t* anc_ptr,
                               const int32_t* anc_idx, const float* const* w1t_ptrs, const float* const* b1_ptrs,
                               const float* const* w2_ptrs, const float* const* b2_ptrs, float* u_proper,
                               float* s_proper, float* mode, float delta_i_weight, float* pose_f, float* pose_u,
                               float* pose_s, float* pose_v, int B, int num_body_joints, int svd_flavor,
                               int32_t* sync_ws, hps_stream_t stream);

/* Tuning hook (tests/dev only): kernel choice of hps_smpl_blend: 0 / 1 = tiled (default), 2 = stationary-A (same bits). */
int hps_dev_blend_mode(int mode);

/* Tuning hook (tests/dev only): kernel choice of hps_vertex_uncertainty: 0 = automatic, 1 = two-sweep, 2 = single pass
 * with 128 vertices per workgroup in LDS, 3 = with 64. */
int hps_dev_unc_mode(int mode);

/* Cross-check hook: 2 = hps_smpl_mesh_fused always runs its two-stage K loop (calls of at most two mesh tiles otherwise take the
 * four-stage form: same bits), 0 = the product rule. */
int hps_dev_mesh_stages(int stages);
/* hps_smpl_mesh_fused_shared_shape_bf16x3: mg = 2 selects the 64-mesh tile (four waves, three workgroups per CU) for A/B runs; any other
 * value = the product's 128-mesh tile.  hps_smpl_split_bf16x3_mesh_tile() follows the switch. */
int hps_dev_mesh_split_groups(int mg);
/* timing ablations of the bf16x3 mesh kernel (results garbage): 1 = K loop only, 2 = skinning only, 3 = operand stream without MFMAs */
int hps_dev_mesh_split_ablate(int ablate);   /* ... 4 = DMA pieces in one burst, 5 = operands one chunk ahead (both: results valid); 6-9 = K loop
                                              * without its operand stream / its barriers; 10-12 = skinning only without the transforms' DMA / the
                                              * stores / the arithmetic; 13 = the stage-free barrier behind the first two product groups (results valid)
                                              * (csrc/mesh_split.hip, tests/dev/mesh_split_time.py) */
/* start delay of the second resident workgroup of every CU, in units of ~1.5 us (< 0: the product's value) */
int hps_dev_mesh_split_stagger(int units);

/* Experiment: request at least `bytes` of dynamic LDS for the fused mesh kernel (unused space), i.e. cap its workgroups per CU
 * (36 KiB -> 4, 52 KiB -> 3, 72 KiB -> 2, 150 KiB -> 1).  0 restores the product value. */
int hps_dev_mesh_lds_floor(int bytes);

/* Experiment hook: K slices of hps_conv3x3_winograd's 8 x 8 geometry (0 = the product rule: four when >= 32 chunks).  Also changes
 * hps_conv3x3_winograd_workspace's answer. */
int hps_dev_wino_quad_ksplit(int ks);
/* profiling: hps_stem_winograd_pooled_nchw with ablate = 8 (no gather loads) / 9 (no window stores); 0 = the product */
int hps_dev_stem_winograd_pooled_nchw(const float* x, const float* u, const float* scale, const float* shift, float* pooled, float* side,
                                      int B, int H, int W, int opad, int relu, int ablate, hps_stream_t stream);
/* experiment: Winograd layer on half items (4 x 8 tiles), two four-wave workgroups per CU; u4 = the half-chunk packing of the filters */
int hps_dev_conv3x3_winograd_half(const float* x, const float* u4, const float* scale, const float* shift, const float* residual,
                                  float* y, int B, int H, int W, int ipad, int Cin, int Cout, int opad, int relu, int ablate,
                                  int wgs_per_cu, hps_stream_t stream);
/* profiling: shader-clock stamps written by hps_dev_conv3x3_winograd(ablate = 11): workgroup b's phase k at [16 b + k] */
int hps_dev_wino_stamps(unsigned long long* host_out, int n);

/* Tuning hook (tests/dev only): 1 = hps_conv2d_bn_act_pad skips its epilogue (results are garbage), 0 = normal. */
int hps_dev_conv_pad_ablate(int mode);

/* nn.MaxPool2d(3, stride 2, pad 1) on NHWC (models/resnet.py:152, :207). */
int hps_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, hps_stream_t stream);

/* AdaptiveAvgPool2d((1,1)) + flatten on NHWC (models/resnet.py:214-215): (B,H,W,C) -> (B,C). */
int hps_global_avgpool(const float* x, float* y, int B, int HW, int C, hps_stream_t stream);

/* Profiling ablations of hps_smpl_mesh_fused (K = 4 only): 1 = no skinning (stores v_posed), 2 = no MFMA, 3 = K loop only
 * (no A fetch, no skinning, no stores), 4 = no operand DMA (garbage results).  0 = the product kernel. */
int hps_dev_mesh_fused(const float* xt, const float* bmat_p, const float* v_template, const float* a,
                       const int32_t* w_idx, const float* w_val, int K, int num_joints, const float* transl,
                       float* verts, int M, int V, int kp, int mp, int np, int ablate, hps_stream_t stream);

/* Profiling ablations of hps_conv3x3_winograd: 1 = no patch loads / input transform, 2 = no MFMA, 3 = no filter DMA,
 * 4 = no epilogue, 5 = raw DMA but no transform, 6 = transform but no raw DMA, 7 = raw DMA from one line, 8 = every item
 * reads the first window, 9 = DMAs issued in one burst per chunk, 10 = no barrier per chunk (races; results are garbage except for 9); 0 = the product
 * kernel.  16 x 16-block geometry only.  The eight-wave product form has its own set (21-44, csrc/conv_wino.hip), among them the two that
 * price a bf16x3 arithmetic for these layers, timing only: 43 = every position's four fp32 MFMAs per 8 k as three v_mfma_f32_32x32x16_bf16,
 * 44 = 43 + the VALU work of splitting the transformed input into three bf16 pieces (tests/dev/wino_bf16x3_price.py). */
int hps_dev_conv3x3_winograd(const float* x, const float* u, const float* scale, const float* shift,
                             const float* residual, float* y, int B, int H, int W, int ipad, int Cin, int Cout,
                             int opad, int relu, float* splitk_ws, int ablate, hps_stream_t stream);

/* hps_stem_winograd with a profiling ablation (csrc/stem_wino.hip: 1 = patch pixels not read, 2 = no MFMAs, 3 = no output
 * transform, 4 = no barriers, 5 = no filter fragment reads; results are wrong by construction). */
int hps_dev_stem_winograd(const float* frames, const float* u, const float* scale, const float* shift, float* y, int B, int H,
                          int W, int opad, int relu, int ablate, hps_stream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* HPS_DEV_H_ */

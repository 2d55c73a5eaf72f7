// Host-side composites: sequences of libhps launches issued from native code in ONE call across the C ABI.
//
// Why: after the convolution work the inference step became host-paced -- a kernel-trace timeline of the bench showed
// the Python side needing ~0.9 ms to issue the encoder's 27 launches (ctypes marshalling, ~30 us each) and ~1.8 ms to
// walk the head's 8 kinematic levels (launch, D2H, stream sync, LAPACK, H2D, launch: ~145 us of interpreter and
// torch overhead per level on top of ~85 us of kernel time), which together exceeded the GPU time of the step.
// These entry points issue exactly the same launches in the same order on the same stream.
#include <vector>

#include "hps_common.h"

using namespace hps;

// ABI guard for bindings that mirror hps_enc_op by hand (ctypes.Structure, cgo, ...)
extern "C" int hps_sizeof_enc_op(void) { return (int)sizeof(hps_enc_op); }

// models/resnet.py:202-217 as a list of operations on caller-owned buffers (padded frames, weights)
extern "C" int hps_encoder_run(const hps_enc_op* ops, int n_ops, hps_stream_t stream) {
    if (!ops && n_ops > 0) return bad_arg("hps_encoder_run: null op list");
    for (int i = 0; i < n_ops; ++i) {
        const hps_enc_op& o = ops[i];
        int rc;
        switch (o.kind) {
            case HPS_ENC_RELAYOUT:
                rc = hps_nchw_to_padded_nhwc(o.x, o.y, o.B, o.Cin, o.H, o.W, o.opad, stream);
                break;
            case HPS_ENC_RELAYOUT_GENERIC:
                rc = hps_nchw_to_padded_nhwc_generic(o.x, o.y, o.B, o.Cin, o.Cout, o.H, o.W, o.KW, o.opad, stream);
                break;
            case HPS_ENC_CONV:
                rc = hps_conv2d_bn_act_pad(o.x, o.w, o.scale, o.shift, o.residual, o.y, o.B, o.H, o.W, o.ipad, o.Cin, o.Cout,
                                           o.KH, o.KW, o.stride, o.pad, o.opad, o.relu, o.row_mode, o.variant, o.ksplit,
                                           o.splitk_ws, stream);
                break;
            case HPS_ENC_CONV_DOWN:
                rc = hps_conv2d_bn_act_pad_down(o.x, o.w, o.scale, o.shift, o.y, o.w_down, o.scale_down, o.shift_down, o.y_down, o.B, o.H,
                                                o.W, o.ipad, o.Cin, o.Cout, o.KH, o.KW, o.stride, o.pad, o.opad, o.relu, o.variant, o.ksplit,
                                                o.splitk_ws, stream);
                break;
            case HPS_ENC_CONV_WINOGRAD:
                rc = hps_conv3x3_winograd(o.x, o.w, o.scale, o.shift, o.residual, o.y, o.B, o.H, o.W, o.ipad, o.Cin, o.Cout, o.opad,
                                          o.relu, o.splitk_ws, stream);
                break;
            case HPS_ENC_STEM_SPLIT:
                rc = hps_stem_phase_split(o.x, o.y, o.B, o.Cin, o.H, o.W, stream);
                break;
            case HPS_ENC_STEM_WINOGRAD:
                rc = hps_stem_winograd(o.x, o.w, o.scale, o.shift, o.y, o.B, o.H, o.W, o.opad, o.relu, stream);
                break;
            case HPS_ENC_STEM_WINOGRAD_POOLED:
                rc = hps_stem_winograd_pooled(o.x, o.w, o.scale, o.shift, o.y, o.splitk_ws, o.B, o.H, o.W, o.opad, o.relu, stream);
                break;
            case HPS_ENC_STEM_WINOGRAD_POOLED_NCHW:
                rc = hps_stem_winograd_pooled_nchw(o.x, o.w, o.scale, o.shift, o.y, o.splitk_ws, o.B, o.H, o.W, o.opad, o.relu, stream);
                break;
            case HPS_ENC_MAXPOOL:
                rc = hps_maxpool3x3s2_pad(o.x, o.y, o.B, o.H, o.W, o.Cin, o.opad, stream);
                break;
            case HPS_ENC_AVGPOOL:
                rc = hps_global_avgpool_pad(o.x, o.y, o.B, o.H, o.W, o.Cin, o.ipad, stream);
                break;
            default:
                return bad_arg("hps_encoder_run: unknown op kind");
        }
        if (rc != HPS_OK) return rc;
    }
    return HPS_OK;
}

// models/poseMF_shapeGaussian_net.py:121-160: the joint loop, one kinematic depth level at a time, with the host
// LAPACK SVD (:137) between the two kernels of a level.  Pinned staging buffers are the caller's; the stream
// synchronisation of level l + 1 also retires the upload of level l, so one pair of staging buffers suffices.
extern "C" int hps_head_pose_levels(const float* embed, int embed_dim, int hidden, const int32_t* level_joints,
                                    const int32_t* level_sizes_host, int n_levels, const int32_t* anc_ptr,
                                    const int32_t* anc_idx, const float* const* w1t_ptrs, const float* const* b1_ptrs,
                                    const float* const* w2_ptrs, const float* const* b2_ptrs, float* u_proper,
                                    float* s_proper, float* mode, float delta_i_weight, float* pose_f, float* pose_u,
                                    float* pose_s, float* pose_v, float* f_level_dev, float* usv_level_dev,
                                    float* f_host_pinned, float* usv_host_pinned, int B, int num_body_joints,
                                    int svd_threads, int svd_mode, hps_stream_t stream) {
    if (!embed || !level_joints || !level_sizes_host) return bad_arg("hps_head_pose_levels: null pointer");
    const int wide = svd_mode & HPS_HEAD_WIDE_WORKGROUPS;
    svd_mode &= ~HPS_HEAD_WIDE_WORKGROUPS;
    if (svd_mode == HPS_SVD_DEVICE || svd_mode == HPS_SVD_DEVICE_FMA) {
        const int flavor = (svd_mode == HPS_SVD_DEVICE_FMA ? HPS_SVD_ROUNDING_FMA : HPS_SVD_ROUNDING_REFERENCE) | wide;
        // every level is ONE kernel (MLPs + in-kernel gesdd-faithful SVD + proper fix): stream-ordered, no host round trip
        int first_d = 0;
        for (int l = 0; l < n_levels; ++l) {
            const int n_level = level_sizes_host[l];
            const int rc = hps_head_joint_level_svd(embed, embed_dim, hidden, level_joints + first_d, n_level, anc_ptr, anc_idx,
                                                    w1t_ptrs, b1_ptrs, w2_ptrs, b2_ptrs, u_proper, s_proper, mode, delta_i_weight,
                                                    pose_f, pose_u, pose_s, pose_v, B, num_body_joints, flavor, stream);
            if (rc != HPS_OK) return rc;
            first_d += n_level;
        }
        return HPS_OK;
    }
    if (svd_mode != HPS_SVD_HOST) return bad_arg("hps_head_pose_levels: svd_mode");
    if (!f_level_dev || !usv_level_dev || !f_host_pinned || !usv_host_pinned)
        return bad_arg("hps_head_pose_levels: the host-LAPACK mode needs the staging buffers");
    hipStream_t s = (hipStream_t)stream;
    int first = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int n_level = level_sizes_host[l];
        const int32_t* ids = level_joints + first;
        const size_t count = (size_t)B * n_level;
        int rc = hps_head_joint_level(embed, embed_dim, hidden, ids, n_level, anc_ptr, anc_idx, w1t_ptrs, b1_ptrs, w2_ptrs,
                                      b2_ptrs, u_proper, s_proper, mode, delta_i_weight, pose_f, f_level_dev, B,
                                      num_body_joints, stream);
        if (rc != HPS_OK) return rc;
        hipError_t e = hipMemcpyAsync(f_host_pinned, f_level_dev, count * 9 * sizeof(float), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            set_error("hps_head_pose_levels: %s", hipGetErrorString(e));
            return (int)e;
        }
        rc = hps_host_svd3_packed(f_host_pinned, usv_host_pinned, (int)count, svd_threads);
        if (rc != HPS_OK) return rc;
        e = hipMemcpyAsync(usv_level_dev, usv_host_pinned, count * 21 * sizeof(float), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) {
            set_error("hps_head_pose_levels: %s", hipGetErrorString(e));
            return (int)e;
        }
        rc = hps_head_svd_finish(usv_level_dev, ids, n_level, pose_u, pose_s, pose_v, u_proper, s_proper, mode, B,
                                 num_body_joints, stream);
        if (rc != HPS_OK) return rc;
        first += n_level;
    }
    return HPS_OK;
}

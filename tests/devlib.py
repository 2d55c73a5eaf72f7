"""Test-side binding of libhps_dev.so (include/hps_dev.h): the ctypes prototypes of its extra entry points and the helpers that drive
the earlier kernel generations / measured-and-rejected experiments kept there as bit-level cross-checks.  Nothing in the product package
names a dev symbol; hierarchicalprobabilistic3dhuman_amd._capi.dev_library() -- the loader -- reads DEV_PROTOTYPES from this file.

    from devlib import dev_library, plain_conv, plain_forward, head_levels_fused
"""
import ctypes as _c

import torch

_P = _c.c_void_p
_I = _c.c_int

# name -> argtypes of the entry points libhps_dev.so exports beside the product ABI (restype int)
DEV_PROTOTYPES = {
    "hps_dev_lbs_variant": [_P, _I, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P],
    "hps_conv2d_bn_act": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "hps_conv2d_bn_act_v2": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "hps_conv2d_bn_act_v3": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "hps_dev_conv_pad_ablate": [_I],
    "hps_dev_wino_quad_ksplit": [_I],
    "hps_dev_wino_stamps": [_P, _I],
    "hps_dev_conv3x3_winograd_half": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "hps_dev_unc_mode": [_I],
    "hps_dev_mesh_lds_floor": [_I],
    "hps_dev_mesh_stages": [_I],
    "hps_dev_mesh_split_groups": [_I],
    "hps_dev_mesh_split_ablate": [_I],
    "hps_dev_mesh_split_stagger": [_I],
    "hps_dev_blend_mode": [_I],
    "hps_nchw_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _P],
    "hps_maxpool3x3s2": [_P, _P, _I, _I, _I, _I, _P],
    "hps_global_avgpool": [_P, _P, _I, _I, _I, _P],
    "hps_dev_stem_winograd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hps_dev_stem_winograd_pooled_nchw": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hps_dev_conv3x3_winograd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "hps_dev_head_pose_levels_fused": [_P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_float, _P, _P, _P, _P, _I, _I, _I,
                                   _P, _P],
    "hps_dev_smpl_pose_prep_v1": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _I, _P],
    "hps_dev_smpl_joints_v1": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P],
    "hps_dev_mesh_fused": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
}

DEV_WS_HEAD_SYNC = 7          # include/hps_dev.h: HPS_DEV_WS_HEAD_SYNC (hps_query_workspace of the DEV build only)


def dev_library():
    from hierarchicalprobabilistic3dhuman_amd import _capi
    return _capi.dev_library()


def plain_conv(cb, x, residual=None, relu=True, kernel="v3"):
    """The un-padded kernel generations of csrc/conv.hip on a resnet._ConvBN's filters: x (B,H,W,Cin) plain NHWC -> (B,Ho,Wo,Cout).
    kernel: "v3" direct global->LDS, "v2" register-staged, "v1" generic (any Cin % 4 == 0).  cb.variant / cb.ksplit select the tile
    and the K slices as they do for the product kernel."""
    from hierarchicalprobabilistic3dhuman_amd import _capi
    B, H, W, C = x.shape
    assert C == cb.cin_p
    Ho = (H + 2 * cb.pad - cb.kh) // cb.stride + 1
    Wo = (W + 2 * cb.pad - cb.kw) // cb.stride + 1
    y = torch.empty(B, Ho, Wo, cb.cout, device=x.device, dtype=torch.float32)
    P = _capi.ptr
    with _capi.dev_library():
        if cb.wn is not None and kernel == "v3":
            ksplit = cb.ksplit if cb.ksplit > 0 else cb._auto_ksplit(Ho * Wo)
            ws = torch.empty(ksplit, B * Ho * Wo, cb.cout, device=x.device, dtype=torch.float32) if ksplit > 1 else None
            _capi.call("hps_conv2d_bn_act_v3", P(x), P(cb.wn), P(cb.zeros), P(cb.scale), P(cb.shift),
                       P(residual) if residual is not None else None, P(y), B, H, W, C, cb.cout, cb.kh, cb.kw,
                       cb.stride, cb.pad, 1 if relu else 0, cb.variant if ksplit <= 1 else 1, ksplit,
                       P(ws) if ws is not None else None, _capi.stream())
        elif cb.wn is not None and kernel == "v2":
            _capi.call("hps_conv2d_bn_act_v2", P(x), P(cb.wn), P(cb.scale), P(cb.shift),
                       P(residual) if residual is not None else None, P(y), B, H, W, C, cb.cout, cb.kh, cb.kw,
                       cb.stride, cb.pad, 1 if relu else 0, cb.variant, _capi.stream())
        else:
            _capi.call("hps_conv2d_bn_act", P(x), P(cb.wk), P(cb.scale), P(cb.shift),
                       P(residual) if residual is not None else None, P(y), B, H, W, C, cb.cout, cb.kh, cb.kw,
                       cb.stride, cb.pad, 1 if relu else 0, _capi.stream())
    return y


def plain_forward(enc, x):
    """models/resnet.py:202-217 on the un-padded kernel generation (plain NHWC activations, csrc/conv.hip): the cross-check of the
    product's halo-padded encoder."""
    from hierarchicalprobabilistic3dhuman_amd import _capi
    prep = enc._prepared or enc.prepare()
    x = _capi.f32c(x)
    B, C, H, W = x.shape
    P = _capi.ptr
    cp = enc._cin_pad
    assert cp in (4, 20, 64), "hps_nchw_to_nhwc handles 4 / 20 / 64 padded channels"
    with _capi.dev_library():
        s = _capi.stream()
        xh = torch.empty(B, H, W, cp, device=x.device, dtype=torch.float32)
        _capi.call("hps_nchw_to_nhwc", P(x), P(xh), B, C, H, W, cp, s)
        y = plain_conv(prep["stem"], xh, relu=True)                        # conv1 + bn1 + relu
        Bh, Hh, Wh, Ch = y.shape
        Hp, Wp = (Hh + 2 - 3) // 2 + 1, (Wh + 2 - 3) // 2 + 1
        yp = torch.empty(B, Hp, Wp, Ch, device=x.device, dtype=torch.float32)
        _capi.call("hps_maxpool3x3s2", P(y), P(yp), B, Hh, Wh, Ch, s)
        y = yp
        for c1, c2, down in prep["blocks"]:                                # BasicBlock.forward :62-78
            identity = plain_conv(down, y, relu=False) if down is not None else y
            out = plain_conv(c1, y, relu=True)
            y = plain_conv(c2, out, residual=identity, relu=True)
        Bq, Hq, Wq, Cq = y.shape
        feats = torch.empty(B, Cq, device=x.device, dtype=torch.float32)
        _capi.call("hps_global_avgpool", P(y), P(feats), B, Hq * Wq, Cq, s)
    return feats


_SYNC_WS = {}


def head_levels_fused(net, feats):
    """PoseMFShapeGaussianNet.forward(None, input_feats=feats) with ALL kinematic levels in ONE launch (hps_dev_head_pose_levels_fused:
    the workgroups of an image tile hand their level's results to each other through counters in a small zeroed workspace, one per
    stream) -- the round-5 experiment that was measured and not adopted (not faster than eight launches), kept as a bit-level
    cross-check of the per-level path.  Device SVD, wide workgroups (the latency mode's form).  Returns the net's 8-tuple, or None
    when the single launch's grid would not fit the chip at once (the product path is the only one then)."""
    from hierarchicalprobabilistic3dhuman_amd import _capi
    p = net._prepared or net.prepare()
    feats = _capi.f32c(feats)
    B, dev = feats.shape[0], feats.device
    sizes = p["level_sizes_host"]
    if len(p["levels"]) > 32 or int(sizes.max()) * ((B + 3) // 4) > torch.cuda.get_device_properties(dev).multi_processor_count:
        return None
    embed, shape_dist, glob, cam = net._trunk(feats, p)
    out = net._pose_buffers(B, dev)
    pose_F, pose_U, pose_S, pose_V, U_proper, S_proper, mode = out
    embed_dim = net.config.MODEL.EMBED_DIM
    delta = float(net.config.MODEL.DELTA_I_WEIGHT) if net.config.MODEL.DELTA_I else 0.0
    P, VP = _capi.ptr, _capi._P
    with _capi.dev_library():
        key = (id(net), torch.cuda.current_stream().cuda_stream, (B + 3) // 4)
        sync = _SYNC_WS.get(key)
        if sync is None:
            sync = _SYNC_WS[key] = torch.zeros(_capi.query_workspace(DEV_WS_HEAD_SYNC, B) // 4, dtype=torch.int32, device=dev)
        _capi.call("hps_dev_head_pose_levels_fused", P(embed), embed_dim, embed_dim // 2, _capi.iptr(p["level_joints"]),
                   VP(sizes.data_ptr()), len(p["levels"]), _capi.iptr(p["anc_ptr"]), _capi.iptr(p["anc_idx"]),
                   VP(p["w1t_ptrs"].data_ptr()), VP(p["b1_ptrs"].data_ptr()), VP(p["w2_ptrs"].data_ptr()),
                   VP(p["b2_ptrs"].data_ptr()), P(U_proper), P(S_proper), P(mode), delta, P(pose_F), P(pose_U), P(pose_S),
                   P(pose_V), B, net.num_joints, net._flavor() | _capi.HEAD_WIDE_WORKGROUPS, _capi.iptr(sync), _capi.stream())
    return pose_F, pose_U, pose_S, pose_V, mode, shape_dist, glob, cam


def sync_workspaces(net):
    """The counter workspaces head_levels_fused has allocated for ``net`` (the kernel leaves them zero)."""
    return {k: v for k, v in _SYNC_WS.items() if k[0] == id(net)}


def enable_plain_call():
    """tests/dev bring-up scripts: ``cb(x, residual=None, relu=True)`` on a resnet._ConvBN runs the un-padded kernel generation selected by
    ``cb.kernel`` ("v3" default, "v2", "v1") -- the call form those scripts were written against, kept out of the product class."""
    from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN
    _ConvBN.__call__ = lambda self, x, residual=None, relu=True: plain_conv(self, x, residual, relu, kernel=getattr(self, "kernel", "v3"))

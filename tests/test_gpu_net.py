"""GPU parity: ResNet-18 encoder (fp32 MFMA implicit GEMM) and the distribution head against the golden
vectors of the imported reference and against the oracle.

Stated tolerances: encoder features <= 1e-4 relative; head F, S, mode <= 1e-4 absolute given
identical features and the host LAPACK SVD; U, V <= 1e-3 (they are sign-ambiguous and ill-conditioned where
singular values are close; the sign choice is LAPACK's, the same as the reference's)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as O
from hierarchicalprobabilistic3dhuman_amd import configs, _capi
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN
from conftest import maxerr

pytestmark = pytest.mark.gpu


def test_encoder_reproduces_reference_features(dev, net_gpu, golden, golden_input):
    feats = net_gpu.image_encoder(golden_input.to(dev))
    ref = golden["net_feats"]
    assert feats.shape == (2, 512)
    assert maxerr(feats, ref) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("svd_mode", ["device", "host"])
def test_net_reproduces_reference_outputs(svd_mode, dev, net_gpu, golden, golden_input):
    """Both SVD modes against the reference's outputs: "host" = MKL sgesdd (the reference's routine), "device" = the in-kernel
    SVD that follows sgesdd step by step -- same signs on the golden matrices, so U and V agree too."""
    net_gpu.svd_mode = svd_mode
    try:
        pose_F, pose_U, pose_S, pose_V, mode, shape_dist, glob, cam = net_gpu(golden_input.to(dev))
    finally:
        net_gpu.svd_mode = "device"
    assert maxerr(pose_F, golden["net_F"]) <= 1e-4
    assert maxerr(pose_S, golden["net_S"]) <= 1e-5 * max(1.0, float(golden["net_S"].max()))
    assert maxerr(mode, golden["net_mode"]) <= 1e-5 if svd_mode == "device" else maxerr(mode, golden["net_mode"]) <= 1e-4
    assert maxerr(pose_U, golden["net_U"]) <= 1e-3 and maxerr(pose_V, golden["net_V"]) <= 1e-3
    assert maxerr(shape_dist.loc, golden["net_shape_loc"]) <= 1e-4
    assert maxerr(shape_dist.scale, golden["net_shape_scale"]) <= 1e-4
    assert maxerr(glob, golden["net_glob"]) <= 1e-4 and maxerr(cam, golden["net_cam"]) <= 1e-4
    assert isinstance(shape_dist, torch.distributions.Normal)


@pytest.mark.parametrize("B", [1, 3, 17, 64])
def test_head_from_features_matches_oracle(B, dev, net_gpu, net_cpu):
    feats = torch.rand(B, 512, generator=torch.Generator().manual_seed(B)) * 2
    ref = O.head_forward(net_cpu[1], feats, configs.SMPL_PARENTS)
    out = net_gpu(None, input_feats=feats.to(dev))
    assert maxerr(out[0], ref[0]) <= 1e-4 and maxerr(out[2], ref[2]) <= 1e-4 and maxerr(out[4], ref[4]) <= 1e-4
    assert maxerr(out[5].loc, ref[5][0]) <= 1e-4 and maxerr(out[6], ref[6]) <= 1e-4 and maxerr(out[7], ref[7]) <= 1e-4


@pytest.mark.parametrize("cfg", [
    # B, H, Cin, Cout, k, stride, pad   -- one case per tile configuration / layer type of ResNet-18
    (2, 64, 64, 64, 3, 1, 1), (2, 64, 64, 128, 3, 2, 1), (2, 64, 64, 128, 1, 2, 0), (64, 8, 512, 512, 3, 1, 1),
    (3, 16, 256, 256, 3, 1, 1), (1, 8, 512, 512, 3, 1, 1), (1, 30, 20, 64, 7, 2, 3), (2, 9, 128, 256, 3, 2, 1)])
def test_conv_bn_relu_kernel(cfg, dev):
    B, H, Cin, Cout, k, s, p = cfg
    torch.manual_seed(sum(cfg))
    conv = torch.nn.Conv2d(Cin, Cout, k, s, p, bias=False)
    bn = torch.nn.BatchNorm2d(Cout).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    x = torch.randn(B, Cin, H, H)
    Ho = (H + 2 * p - k) // s + 1
    res = torch.randn(B, Cout, Ho, Ho)
    with torch.no_grad():
        want = F.relu(bn(conv(x)) + res)
        want_nores = bn(conv(x))
    cb = _ConvBN(conv.to(dev), bn.to(dev))
    xh = x.to(dev).permute(0, 2, 3, 1).contiguous()
    got = cb(xh, residual=res.to(dev).permute(0, 2, 3, 1).contiguous(), relu=True)
    assert maxerr(got.permute(0, 3, 1, 2), want) <= 1e-4 * max(1.0, float(want.abs().max()))
    got2 = cb(xh, relu=False)
    assert maxerr(got2.permute(0, 3, 1, 2), want_nores) <= 1e-4 * max(1.0, float(want_nores.abs().max()))
    # every kernel generation / tile shape that supports this layer gives the same answer
    kernels = [("v1", 0)] + ([(kern, v) for kern in ("v2", "v3") for v in (1, 2, 3)] if Cin % 32 == 0 else [])
    for kern, v in kernels:
        cb.kernel, cb.variant = kern, v
        alt = cb(xh, residual=res.to(dev).permute(0, 2, 3, 1).contiguous(), relu=True)
        assert maxerr(alt, got) <= 1e-4 * max(1.0, float(want.abs().max())), (kern, v)
    # split-K (two-pass, deterministic) where the shape allows it
    if Cin % 32 == 0 and Cout % 128 == 0:
        chunks = k * k * Cin // 32
        for ks in (2, 4):
            if chunks % ks == 0:
                cb.kernel, cb.variant, cb.ksplit = "v3", 0, ks
                alt = cb(xh, residual=res.to(dev).permute(0, 2, 3, 1).contiguous(), relu=True)
                assert maxerr(alt, got) <= 1e-4 * max(1.0, float(want.abs().max())), ("split-K", ks)
                assert torch.equal(alt, cb(xh, residual=res.to(dev).permute(0, 2, 3, 1).contiguous(), relu=True))


def _frame(t_nhwc, pad):
    """(B,H,W,C) -> zero-haloed (B,H+2p,W+2p,C) frame."""
    return F.pad(t_nhwc, (0, 0, pad, pad, pad, pad)).contiguous()


@pytest.mark.parametrize("cfg", [
    # B, H, Cin, Cout, k, stride, pad, ipad -- every layer type of ResNet-18, ragged tiles, Wo % 4 != 0, split-K, the row-mode stem
    (2, 64, 64, 64, 3, 1, 1, 1), (2, 64, 64, 128, 3, 2, 1, 1), (2, 64, 64, 128, 1, 2, 0, 1), (64, 8, 512, 512, 3, 1, 1, 1),
    (3, 16, 256, 256, 3, 1, 1, 1), (1, 8, 512, 512, 3, 1, 1, 1), (2, 9, 128, 256, 3, 2, 1, 1), (3, 14, 64, 64, 3, 1, 1, 2),
    (2, 30, 18, 64, 7, 2, 3, 3), (1, 256, 18, 64, 7, 2, 3, 3), (2, 12, 4, 64, 7, 2, 3, 3)])
def test_padded_conv_kernel(cfg, dev):
    """csrc/conv_pad.hip against torch's convolution and, bit for bit, against the conv.hip kernel it replaces
    (same K order and summation order; the row-mode stem has its own K order and is held to the tolerance only)."""
    B, H, Cin, Cout, k, s, p, ipad = cfg
    torch.manual_seed(sum(cfg))
    conv = torch.nn.Conv2d(Cin, Cout, k, s, p, bias=False)
    bn = torch.nn.BatchNorm2d(Cout).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    x = torch.randn(B, Cin, H, H)
    Ho = (H + 2 * p - k) // s + 1
    res = torch.randn(B, Cout, Ho, Ho)
    with torch.no_grad():
        want = F.relu(bn(conv(x)) + res)
        want_nores = bn(conv(x))
    tol = 1e-4 * max(1.0, float(want.abs().max()))
    cb = _ConvBN(conv.to(dev), bn.to(dev))
    cb.use_winograd = False                     # this test is about the direct kernel (test_winograd_conv_kernel covers the other)
    xh = x.to(dev).permute(0, 2, 3, 1).contiguous()
    resh = res.to(dev).permute(0, 2, 3, 1).contiguous()
    xp = _frame(xh, ipad)
    for opad in (0, 1):
        out = torch.full((B, Ho + 2 * opad, Ho + 2 * opad, Cout), 7.0, device=dev)
        cb.padded(xp, ipad, out, opad, residual=_frame(resh, opad), relu=True)
        inner = out[:, opad:opad + Ho, opad:opad + Ho]
        assert maxerr(inner.permute(0, 3, 1, 2), want) <= tol
        if opad:        # the halo is never written
            assert float((out[:, 0] - 7).abs().max()) == 0 and float((out[:, :, -1] - 7).abs().max()) == 0
        out2 = torch.zeros_like(out)
        cb.padded(xp, ipad, out2, opad, relu=False)
        assert maxerr(out2[:, opad:opad + Ho, opad:opad + Ho].permute(0, 3, 1, 2), want_nores) <= tol
    if Cin % 32 == 0:
        for v in (0, 1, 2, 3, 4):
            if v == 4 and Cout != 64:
                continue
            cb.variant = v
            plain = cb(xh, residual=resh, relu=True)        # same tile / split-K rule on both sides
            out = torch.zeros(B, Ho + 2, Ho + 2, Cout, device=dev)
            cb.padded(xp, ipad, out, 1, residual=_frame(resh, 1), relu=True)
            assert torch.equal(out[:, 1:-1, 1:-1], plain), v
        cb.variant = 0
        if Cout % 128 == 0:
            chunks = k * k * Cin // 32
            for ks in (2, 4):
                if chunks % ks == 0:
                    cb.ksplit = ks
                    plain_k = cb(xh, residual=resh, relu=True)
                    out = torch.zeros(B, Ho + 2, Ho + 2, Cout, device=dev)
                    cb.padded(xp, ipad, out, 1, residual=_frame(resh, 1), relu=True)
                    assert torch.equal(out[:, 1:-1, 1:-1], plain_k), ("split-K", ks)


def test_padded_and_plain_encoders_agree(dev, net_gpu, golden, golden_input):
    enc = net_gpu.image_encoder
    x = golden_input.to(dev)
    try:
        enc.layout = "plain"
        plain = enc(x)
    finally:
        enc.layout = "padded"
    padded = enc(x)
    again = enc(x)
    assert torch.equal(padded, again)                       # frames are reused: nothing stale leaks between calls
    ref = golden["net_feats"]
    assert maxerr(plain, ref) <= 1e-4 * float(ref.abs().max()) and maxerr(padded, ref) <= 1e-4 * float(ref.abs().max())
    assert maxerr(padded, plain) <= 2e-5 * float(ref.abs().max())
    # a different batch in between does not disturb the cached frames
    other = enc(torch.rand(3, 18, 256, 256, device=dev))
    assert torch.isfinite(other).all() and torch.equal(enc(x), padded)


@pytest.mark.parametrize("svd_mode", ["device", "host"])
def test_composite_calls_issue_the_same_work(svd_mode, dev, net_gpu, golden_input):
    """hps_encoder_run / hps_head_pose_levels against the one-launch-per-call Python loops: bit-identical outputs."""
    x = torch.cat([golden_input, torch.rand(3, 18, 256, 256, generator=torch.Generator().manual_seed(5))]).to(dev)
    enc = net_gpu.image_encoder
    net_gpu.svd_mode = svd_mode
    try:
        want = net_gpu(x)
        enc.composite = False
        net_gpu.composite_head = False
        got = net_gpu(x)
    finally:
        enc.composite = True
        net_gpu.composite_head = True
        net_gpu.svd_mode = "device"
    for a, b in zip(want, got):
        if isinstance(a, torch.distributions.Normal):
            assert torch.equal(a.loc, b.loc) and torch.equal(a.scale, b.scale)
        else:
            assert torch.equal(a, b)


def test_padded_pooling_and_layout_kernels(dev):
    P = _capi.ptr
    x = torch.randn(2, 18, 12, 10, generator=torch.Generator().manual_seed(0))
    xp = torch.zeros(2, 18, 16, 18, device=dev)
    _capi.call("hps_nchw_to_padded_nhwc", P(x.to(dev)), P(xp), 2, 18, 12, 10, 3, _capi.stream())
    assert maxerr(xp[:, 3:-3, 3:-3].permute(0, 3, 1, 2), x) == 0.0
    xp[:, 3:-3, 3:-3] = 0
    assert float(xp.abs().max()) == 0.0
    y = torch.randn(2, 64, 13, 11)
    yh = y.to(dev).permute(0, 2, 3, 1).contiguous()
    out = torch.zeros(2, 9, 8, 64, device=dev)
    _capi.call("hps_maxpool3x3s2_pad", P(yh), P(out), 2, 13, 11, 64, 1, _capi.stream())
    assert maxerr(out[:, 1:-1, 1:-1].permute(0, 3, 1, 2), F.max_pool2d(y, 3, 2, 1)) == 0.0
    assert float(out[:, 0].abs().max()) == 0.0 and float(out[:, :, 0].abs().max()) == 0.0
    avg = torch.empty(2, 64, device=dev)
    _capi.call("hps_global_avgpool_pad", P(_frame(yh, 1)), P(avg), 2, 13, 11, 64, 1, _capi.stream())
    plain = torch.empty(2, 64, device=dev)
    with _capi.dev_library():                       # the un-padded generation lives in libhps_dev.so
        _capi.call("hps_global_avgpool", P(yh), P(plain), 2, 13 * 11, 64, _capi.stream())
    assert torch.equal(avg, plain) and maxerr(avg, y.mean(dim=(2, 3))) <= 1e-6


def test_pooling_and_layout_kernels(dev):
    """The un-padded generation's relayout / pooling kernels (libhps_dev.so; the cross-check of the product's padded ones)."""
    with _capi.dev_library():
        _pooling_and_layout_kernels(dev)


def _pooling_and_layout_kernels(dev):
    P = _capi.ptr
    x = torch.randn(2, 18, 12, 10, generator=torch.Generator().manual_seed(0))
    xh = torch.empty(2, 12, 10, 20, device=dev)
    _capi.call("hps_nchw_to_nhwc", P(x.to(dev)), P(xh), 2, 18, 12, 10, 20, _capi.stream())
    assert maxerr(xh[..., :18].permute(0, 3, 1, 2), x) == 0.0 and float(xh[..., 18:].abs().max()) == 0.0
    y = torch.randn(2, 64, 13, 11)
    yh = y.to(dev).permute(0, 2, 3, 1).contiguous()
    out = torch.empty(2, 7, 6, 64, device=dev)
    _capi.call("hps_maxpool3x3s2", P(yh), P(out), 2, 13, 11, 64, _capi.stream())
    assert maxerr(out.permute(0, 3, 1, 2), F.max_pool2d(y, 3, 2, 1)) == 0.0
    avg = torch.empty(2, 64, device=dev)
    _capi.call("hps_global_avgpool", P(yh), P(avg), 2, 13 * 11, 64, _capi.stream())
    assert maxerr(avg, y.mean(dim=(2, 3))) <= 1e-6


def test_training_mode_is_refused(dev, net_gpu):
    net_gpu.train()
    try:
        with pytest.raises(RuntimeError):
            net_gpu(torch.zeros(1, 18, 256, 256, device=dev))
    finally:
        net_gpu.eval()


def test_load_state_dict_invalidates_prepared_weights(dev, net_gpu, net_cpu, golden, golden_input):
    import copy
    net = copy.deepcopy(net_gpu)
    sd = {k: v.clone() for k, v in net_cpu[1].items()}
    sd["fc_cam.bias"] = sd["fc_cam.bias"] + 1.0
    net.load_state_dict(sd)
    out = net(None, input_feats=golden["net_feats"].to(dev))
    assert maxerr(out[7], golden["net_cam"] + 1.0) <= 1e-4


def test_reload_through_the_parent_resets_the_encoder_and_copies_own_their_weights(dev, net_gpu, net_cpu, golden, golden_input):
    """nn.Module.load_state_dict recurses with _load_from_state_dict, so a child's load_state_dict override never
    runs -- the encoder's folded filters must be reset by a post hook; and a deepcopy of a PREPARED net must not keep the
    original's device addresses in its pointer tables."""
    import copy
    x = golden_input.to(dev)
    net = copy.deepcopy(net_gpu)
    feats0 = net.image_encoder(x).clone()                          # prepares encoder (and frames)
    net(x)                                                          # prepares the head
    assert net._prepared is not None and net.image_encoder._prepared is not None
    sd = {k: v.clone() for k, v in net_cpu[1].items()}
    sd["image_encoder.bn1.weight"] = sd["image_encoder.bn1.weight"] * 0.5
    net.load_state_dict(sd)                                         # through the PARENT
    assert net._prepared is None and net.image_encoder._prepared is None
    feats1 = net.image_encoder(x)
    assert maxerr(feats1, feats0) > 1e-3                            # the new BatchNorm scale is in effect
    net.load_state_dict(net_cpu[1])
    assert maxerr(net.image_encoder(x), golden["net_feats"]) <= 1e-4 * float(golden["net_feats"].abs().max())
    # a copy of a prepared net rebuilds its own tables; deleting the original must not matter
    net(x)
    clone = copy.deepcopy(net)
    assert clone._prepared is None and clone.image_encoder._prepared is None and clone._pinned_bufs == {}
    del net
    torch.cuda.empty_cache()
    out = clone(x)
    assert maxerr(out[0], golden["net_F"]) <= 1e-4


def test_device_svd_kernel_is_the_host_emulation_and_tracks_lapack(dev, golden):
    """hps_svd3_packed (device) against the same algorithm compiled for the host (bit for bit: contraction is off in
    svd3_gesdd.h) and against torch.svd; then the head in both SVD modes on random features: S and mode agree to 1e-5 /
    1e-4, and U / V wherever the signs agree (the rare (2,3) double flips are counted, <= 0.2 % of joints)."""
    import ctypes
    torch.manual_seed(3)
    F = torch.cat([golden["net_F"].reshape(-1, 3, 3), torch.eye(3)[None] + 0.7 * torch.randn(50000, 3, 3)]).contiguous()
    n = F.shape[0]
    usv = torch.empty(n, 21, device=dev)
    _capi.call("hps_svd3_packed", _capi.ptr(F.to(dev)), _capi.ptr(usv), n, _capi.stream())
    host = torch.empty(n, 21)
    assert _capi.load().hps_host_svd3_emulated(ctypes.c_void_p(F.data_ptr()), ctypes.c_void_p(host.data_ptr()), n) == 0
    assert torch.equal(usv.cpu(), host)
    U, S, V = torch.svd(F)
    assert maxerr(usv[:, 9:12], S) <= 2e-6 * float(S.max())
    agree = ((U * usv[:, :9].cpu().reshape(n, 3, 3)).sum(1) > 0).all(1)
    assert float((~agree).float().mean()) <= 5e-4 and bool(agree[:46].all())


def test_head_svd_modes_agree(dev, net_gpu):
    feats = (torch.rand(64, 512, generator=torch.Generator().manual_seed(8)) * 2).to(dev)
    net_gpu.svd_mode = "host"
    try:
        h = net_gpu(None, input_feats=feats)
    finally:
        net_gpu.svd_mode = "device"
    d = net_gpu(None, input_feats=feats)
    # joints whose singular vectors came out with LAPACK's signs (descendants of a flipped joint see other inputs: excluded too)
    same = ((h[1] * d[1]).sum(2) > 0).all(2) & ((h[3] * d[3]).sum(2) > 0).all(2)            # (B, 23)
    assert float((~same).float().mean()) <= 2e-3
    clean = same.all(1)                                                                       # images without any flip
    assert int(clean.sum()) >= 60
    assert maxerr(d[0][clean], h[0][clean]) <= 1e-4 and maxerr(d[4][clean], h[4][clean]) <= 1e-4
    assert maxerr(d[2][clean], h[2][clean]) <= 1e-5 * max(1.0, float(h[2].max()))
    assert maxerr(d[1][clean], h[1][clean]) <= 1e-3 and maxerr(d[3][clean], h[3][clean]) <= 1e-3


@pytest.mark.parametrize("cfg", [
    # B, H, Cin, Cout -- the stride-1 3x3 layers of layer1 / layer2 / layer3 at the 256x256 input, plus an odd batch
    (2, 64, 64, 64), (2, 32, 128, 128), (3, 16, 256, 256), (1, 16, 64, 128), (5, 48, 8, 64), (300, 16, 64, 64),
    # layer4's 8 x 8 maps: four images per item (full and partial quads), K in four slices (512, 256) or one (64)
    (4, 8, 512, 512), (5, 8, 512, 512), (1, 8, 256, 128), (7, 8, 64, 64), (70, 8, 256, 64)])
def test_winograd_conv_kernel(cfg, dev):
    """csrc/conv_wino.hip (Winograd F(2x2, 3x3) + BatchNorm + residual + ReLU) against torch's convolution and against the direct
    implicit-GEMM kernel: same results up to fp32 rounding of a different summation order (<= 1e-5 of the output scale)."""
    B, H, Cin, Cout = cfg
    torch.manual_seed(sum(cfg))
    conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(Cout).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    x = torch.randn(B, Cin, H, H)
    res = torch.randn(B, Cout, H, H)
    with torch.no_grad():
        want = F.relu(bn(conv(x)) + res)
        want_nores = bn(conv(x))
    scale_ref = max(1.0, float(want.abs().max()))
    cb = _ConvBN(conv.to(dev), bn.to(dev))
    assert cb.wino_u is not None and cb.winograd_ok(H, H, 1)
    xp = _frame(x.to(dev).permute(0, 2, 3, 1).contiguous(), 1)
    resh = res.to(dev).permute(0, 2, 3, 1).contiguous()
    for opad in (0, 1):
        out = torch.full((B, H + 2 * opad, H + 2 * opad, Cout), 7.0, device=dev)
        cb.padded(xp, 1, out, opad, residual=_frame(resh, opad), relu=True)
        inner = out[:, opad:opad + H, opad:opad + H]
        assert maxerr(inner.permute(0, 3, 1, 2), want) <= 1e-5 * scale_ref
        if opad:        # the halo is never written
            assert float((out[:, 0] - 7).abs().max()) == 0 and float((out[:, :, -1] - 7).abs().max()) == 0
        out2 = torch.zeros_like(out)
        cb.padded(xp, 1, out2, opad, relu=False)
        assert maxerr(out2[:, opad:opad + H, opad:opad + H].permute(0, 3, 1, 2), want_nores) <= 1e-5 * scale_ref
        if Cin % 32 == 0:                            # the direct kernel's non-row mode needs 32-channel chunks
            cb.use_winograd = False
            direct = torch.zeros_like(out)
            cb.padded(xp, 1, direct, opad, relu=False)
            cb.use_winograd = True
            assert maxerr(out2, direct) <= 1e-5 * scale_ref
    # maps that are neither 16 x 16-pixel blocks nor 8 x 8 stay on the direct kernel, for every batch size
    assert cb.winograd_ok(8, 8, 1) and not cb.winograd_ok(30, 30, 1) and not cb.winograd_ok(24, 24, 1) and not cb.winograd_ok(4, 4, 1)
    if H == 8:          # per-image results do not depend on the quad an image sits in (bit for bit)
        one = torch.zeros(1, H + 2, H + 2, Cout, device=dev)
        cb.padded(xp[B - 1:B].contiguous(), 1, one, 1, relu=False)
        assert torch.equal(one[0], out2[B - 1])


def test_winograd_and_direct_encoders_agree_and_are_batch_invariant(dev, net_gpu, golden, golden_input):
    enc = net_gpu.image_encoder
    x = golden_input.to(dev)
    ref = golden["net_feats"]
    wino = enc(x).clone()
    try:
        enc.set_winograd(False)
        direct = enc(x).clone()
    finally:
        enc.set_winograd(True)
    scale_ref = float(ref.abs().max())
    assert maxerr(wino, ref) <= 1e-4 * scale_ref and maxerr(direct, ref) <= 1e-4 * scale_ref
    assert maxerr(wino, direct) <= 2e-5 * scale_ref
    # per-image features do not depend on the batch they were computed in (bit for bit)
    big = torch.cat([x, torch.rand(5, 18, 256, 256, generator=torch.Generator().manual_seed(9)).to(dev)])
    assert torch.equal(enc(big)[:2], wino) and torch.equal(enc(x[1:2]), wino[1:2])

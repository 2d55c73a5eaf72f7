"""Developer bring-up / micro-benchmark script for a GPU box (test infrastructure: it imports the oracle; not part of the product, not collected by pytest).

    python tests/dev/gpu_bringup.py <section> [...]      sections: rot smpl smpl_perf sampler head encoder encoder_perf e2e
Each section compares the HIP path with the CPU oracle and prints max-abs errors and timings.
"""
import os
import sys
import time

import numpy as np
import torch
import os as _os, sys as _sys
_sys.path[:0] = [_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))]          # tests/ (devlib) and the repository root
import devlib
devlib.enable_plain_call()          # cb(x) = the un-padded kernel generation of libhps_dev.so (tests/devlib.py)


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd import rigid_transform_utils as rtu  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd import sampling_utils as su  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import infer  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402

dev = torch.device("cuda:0")


def err(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def make_smpl():
    model = smpl_data.synthetic_smpl_model(0)
    ex = smpl_data.load_extra_joint_regressors(None)
    params = O.SMPLParams(model, ex, configs.SMPLX_EXTRA_VERTEX_IDS)
    smpl = SMPL(model).to(dev)
    return model, params, smpl


def sec_rot():
    g = torch.randn(5, 6)
    print("rot6d", err(rtu.rot6d_to_rotmat(g.to(dev)), O.rot6d_to_rotmat(g)))
    g3 = torch.randn(3, 6)
    print("rot6d B=3", err(rtu.rot6d_to_rotmat(g3.to(dev)), O.rot6d_to_rotmat(g3)))
    q = torch.randn(7, 4)
    print("quat", err(rtu.quat_to_rotmat(q.to(dev)), O.quat_to_rotmat(q)))
    a = torch.randn(9, 3)
    a[0] = 0
    print("rodrigues", err(rtu.batch_rodrigues(a.to(dev)), O.batch_rodrigues(a)))


def sec_smpl():
    model, params, smpl = make_smpl()
    smpl.keep_intermediates = True
    for M in (1, 37, 130):
        g = torch.Generator().manual_seed(M)
        betas = torch.randn(M, 10, generator=g)
        aa = torch.randn(M, 24, 3, generator=g) * 0.5
        R = O.batch_rodrigues(aa.view(-1, 3)).view(M, 24, 3, 3)
        transl = torch.randn(M, 3, generator=g)
        ref = O.smpl_forward(params, betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False,
                             transl=transl, return_intermediates=True)
        out = smpl(betas=betas.to(dev), body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), pose2rot=False,
                   transl=transl.to(dev))
        L = smpl._last
        print("M=%d rotmat: v_posed %.2e A %.2e J_posed %.2e verts %.2e joints %.2e" % (
            M, err(L["v_posed"], ref["v_posed"]), err(L["a"].view(M, 24, 3, 4), ref["A"][:, :, :3, :]),
            err(L["j_posed"], ref["J_posed"]), err(out.vertices, ref["vertices"]), err(out.joints, ref["joints"])))
        ref2 = O.smpl_forward(params, betas=betas, body_pose=aa[:, 1:].reshape(M, 69), global_orient=aa[:, 0],
                              pose2rot=True)
        out2 = smpl(betas=betas.to(dev), body_pose=aa[:, 1:].reshape(M, 69).to(dev), global_orient=aa[:, 0].to(dev))
        print("M=%d axis-angle: verts %.2e joints %.2e" % (M, err(out2.vertices, ref2["vertices"]),
                                                            err(out2.joints, ref2["joints"])))
    # defaults: zero pose via module parameters
    b1 = torch.randn(1, 10)
    print("tpose", err(smpl(betas=b1.to(dev)).vertices, O.smpl_forward(params, betas=b1)["vertices"]))


def sec_smpl_perf():
    model, params, smpl = make_smpl()
    from hierarchicalprobabilistic3dhuman_amd import _capi
    P = _capi.ptr
    for M in (6528, 16000):
        betas = torch.randn(M, 10, device=dev)
        aa = torch.randn(M * 24, 3, device=dev) * 0.5
        R = rtu.batch_rodrigues(aa).view(M, 24, 3, 3)
        gl, bo = R[:, :1].contiguous(), R[:, 1:].contiguous()
        smpl.keep_intermediates = True
        smpl(betas=betas, body_pose=bo, global_orient=gl, pose2rot=False)
        L = smpl._last
        V, J = 6890, 24
        verts = torch.empty(M, V, 3, device=dev)
        joints = torch.empty(M, 90, 3, device=dev)
        s = _capi.stream()
        t_all = timeit(lambda: smpl(betas=betas, body_pose=bo, global_orient=gl, pose2rot=False), 5, 2)
        t_prep = timeit(lambda: _capi.call("hps_smpl_pose_prep", P(gl), P(bo), 1, P(betas), 10, P(smpl._j_template),
                                           P(smpl._j_shapedirs), _capi.iptr(smpl._parents_i32), _capi.iptr(smpl._depth_i32),
                                           J, P(L["xt"]), smpl._kp, L["xt"].shape[1], P(L["a"]), P(L["j_posed"]), None, M, s))
        t_blend = timeit(lambda: _capi.call("hps_smpl_blend", P(L["xt"]), P(smpl._bmat), P(smpl._v_template_flat),
                                            P(L["v_posed_raw"]), M, 3 * V, smpl._kp, L["xt"].shape[1], smpl._np, L["ldv"], s))
        t_lbs = timeit(lambda: _capi.call("hps_smpl_lbs", P(L["v_posed_raw"]), L["ldv"], P(L["a"]), _capi.iptr(smpl._w_idx),
                                          P(smpl._w_val), smpl._lbs_k, J, None, P(verts), M, V, s), 20, 5)
        t_j = timeit(lambda: _capi.call("hps_smpl_joints", P(verts), P(L["j_posed"]), _capi.iptr(smpl._csr_ptr),
                                        _capi.iptr(smpl._csr_col), P(smpl._csr_val), smpl._n_joint_rows, J, None,
                                        P(joints), M, V, s))
        gb = 166896.0 * M / 1e9
        print("M=%d: forward %.3f ms | prep %.3f blend %.3f (%.1f TF) lbs %.3f (%.2f TB/s alg, %.1f%% of 8) joints %.3f" % (
            M, t_all, t_prep, t_blend, 2.0 * M * 217 * 20670 / t_blend / 1e9, t_lbs, gb / t_lbs, gb / t_lbs / 8 * 100, t_j))
        if M == 6528:
            B, N = 64, 102
            vs = verts[:B * N].view(B, N, V, 3)
            t_u = timeit(lambda: su.vertex_uncertainty(vs))
            print("uncertainty B=64 N=102: %.3f ms (%.2f TB/s single-read)" % (t_u, B * N * V * 12 / t_u / 1e9))
            ref_u = O.vertex_uncertainty(vs[3].cpu())
            print("unc err", err(su.vertex_uncertainty(vs)[3], ref_u))


def sec_mesh_fused():
    """Fused blend + skinning kernel against the unfused pair: bits and time, at the bench sizes."""
    model, params, smpl = make_smpl()
    for M in (6528, 16032):
        g = torch.Generator().manual_seed(5)
        betas = torch.randn(M, 10, generator=g).to(dev)
        pose = (torch.randn(M, 72, generator=g) * 0.5).to(dev)
        run = lambda: smpl(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3])
        smpl.fused_mesh = True
        vf = run().vertices.clone()
        t_f = timeit(run, iters=20)
        smpl.fused_mesh = False
        vu = run().vertices.clone()
        t_u = timeit(run, iters=20)
        smpl.fused_mesh = True
        flop = 2.0 * 224 * 3 * 6890 * M
        print("mesh_fused M=%d: equal=%s  whole SMPL call fused %.3f ms, unfused %.3f ms  (blend FLOP at fused-call time: %.1f TF/s)"
              % (M, torch.equal(vf, vu), t_f, t_u, flop / (t_f * 1e-3) / 1e12), flush=True)
        # the kernel alone
        smpl.lbs_events = []
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for (_, a, b) in smpl.lbs_events][2:]
        smpl.lbs_events = None
        t = sum(ms) / len(ms)
        print("   hps_smpl_mesh_fused alone: %.3f ms = %.1f TF/s fp32 MFMA (of 157.3), LBS-definition %.0f GB/s"
              % (t, flop / (t * 1e-3) / 1e12, 166896.0 * M / (t * 1e-3) / 1e9), flush=True)
        # ablations (libhps_dev.so): what each phase of the kernel costs
        from hierarchicalprobabilistic3dhuman_amd import _capi
        smpl.keep_intermediates = True
        run()
        L = smpl._last
        smpl.keep_intermediates = False
        P = _capi.ptr
        verts = torch.empty(M, 6890, 3, device=dev)
        mp = L["xt"].shape[1]
        for ab, what in ((0, "product"), (0, "product (again)"), (1, "no skinning (stores v_posed)"), (2, "no MFMA"), (3, "K loop only"),
                         (4, "no operand DMA")):
            fn = lambda: _capi.call("hps_dev_mesh_fused", P(L["xt"]), P(smpl._bmat_p), P(smpl._v_template_flat), P(L["a"]),
                                    _capi.iptr(smpl._w_idx), P(smpl._w_val), 4, 24, None, P(verts), M, 6890, smpl._kp, mp,
                                    smpl._np_fused, ab, _capi.stream())
            t = timeit(fn, iters=20)
            print("   ablate %d %-30s %.3f ms  (%.1f TF/s)" % (ab, what, t, flop / (t * 1e-3) / 1e12), flush=True)


def sec_wino():
    """Winograd convolution: time and ablations per layer shape (B = 64)."""
    from hierarchicalprobabilistic3dhuman_amd import _capi
    from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN
    import torch.nn.functional as F
    P = _capi.ptr
    for (H, C) in ((64, 64), (32, 128), (16, 256)):
        conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(C).eval().to(dev)
        cb = _ConvBN(conv, bn)
        x = F.pad(torch.relu(torch.randn(64, H, H, C, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
        out = torch.zeros(64, H + 2, H + 2, C, device=dev)
        res = torch.randn(64, H + 2, H + 2, C, device=dev)
        gflop = 2.0 * 64 * H * H * C * C * 9 / 1e9
        for ab, what in ((0, "product"), (0, "product + residual"), (1, "no patch loads / transform"), (2, "no MFMA"), (3, "no filter DMA"),
                         (4, "no epilogue"), (5, "raw DMA but no transform"), (6, "transform but no raw DMA"), (7, "raw DMA from one line"), (8, "raw DMA of one hot window"), (9, "DMAs issued in one burst"), (10, "no barrier per chunk (races)")):
            r = P(res) if "residual" in what else None
            fn = lambda: _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), r, P(out), 64, H, H, 1, C, C,
                                    1, 1, None, ab, _capi.stream())
            t = timeit(fn, iters=20)
            print("wino %3dx%-3d C=%3d ablate %d %-28s %.4f ms  (direct-conv-equivalent %.1f TF/s, MFMA %.1f TF/s)"
                  % (H, H, C, ab, what, t, gflop / t, gflop / 2.25 / t), flush=True)
        cb.use_winograd = False
        t = timeit(lambda: cb.padded(x, 1, out, 1, relu=True), iters=20)
        print("     direct kernel                                   %.4f ms  (%.1f TF/s)" % (t, gflop / t), flush=True)
    # layer4: 8 x 8 maps, four images per item, K in four slices
    H, C = 8, 512
    conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(C).eval().to(dev)
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.normal_(); bn.bias.data.normal_()
    cb = _ConvBN(conv, bn)
    gflop1 = 2.0 * H * H * C * C * 9 / 1e9
    for B in (64, 16, 5, 1):
        x = F.pad(torch.relu(torch.randn(B, H, H, C, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
        res = F.pad(torch.randn(B, H, H, C, device=dev), (0, 0, 1, 1, 1, 1)).contiguous()
        outs = {}
        for wino in (True, False):
            cb.use_winograd = wino
            out = torch.zeros(B, H + 2, H + 2, C, device=dev)
            t = timeit(lambda: cb.padded(x, 1, out, 1, residual=res, relu=True), iters=20)
            outs[wino] = out.clone()
            print("layer4 8x8 C=512 B=%-3d %-9s %.4f ms  (direct-conv-equivalent %.1f TF/s)" % (B, "winograd" if wino else "direct", t, gflop1 * B / t), flush=True)
        d = (outs[True] - outs[False]).abs().max().item()
        print("     max |winograd - direct| = %.3g of scale %.3g; halo untouched: %s" % (
            d, outs[False].abs().max().item(), bool((outs[True][:, 0].abs().max() == 0) and (outs[True][:, :, 0].abs().max() == 0))), flush=True)


def sec_blend_modes():
    """hps_smpl_blend: tiled kernel vs stationary-A kernel, time and bit equality, at the bench size and others."""
    model, params, smpl = make_smpl()
    from hierarchicalprobabilistic3dhuman_amd import _capi
    P = _capi.ptr
    s = _capi.stream()
    V = 6890
    for M in (6528, 16032, 130, 1):
        betas = torch.randn(M, 10, device=dev)
        aa = torch.randn(M * 24, 3, device=dev) * 0.5
        R = rtu.batch_rodrigues(aa).view(M, 24, 3, 3)
        smpl.keep_intermediates = True
        smpl(betas=betas, body_pose=R[:, 1:].contiguous(), global_orient=R[:, :1].contiguous(), pose2rot=False)
        L = smpl._last
        outs, line = {}, "blend M=%d:" % M
        for mode, name in ((1, "tiled"), (2, "stationary-A"), (0, "auto")):      # auto = tiled
            _capi.call("hps_dev_blend_mode", mode)
            buf = torch.full_like(L["v_posed_raw"], float("nan"))
            fn = lambda: _capi.call("hps_smpl_blend", P(L["xt"]), P(smpl._bmat), P(smpl._v_template_flat), P(buf), M, 3 * V,
                                    smpl._kp, L["xt"].shape[1], smpl._np, L["ldv"], s)
            fn()
            outs[name] = buf[:, :3 * V].clone()
            t = timeit(fn, 10, 3)
            line += "  %s %.3f ms (%.1f TF)" % (name, t, 2.0 * M * 217 * 20670 / t / 1e9)
        _capi.call("hps_dev_blend_mode", 0)
        print(line, " identical:", bool(torch.equal(outs["tiled"], outs["stationary-A"])), bool(torch.equal(outs["auto"], outs["tiled"])),
              "finite:", bool(torch.isfinite(outs["stationary-A"]).all()))


def sec_blend_conv():
    """The blend GEMM run by the halo-padded convolution kernel as a 1x1 'convolution' (row-major operands)."""
    model, params, smpl = make_smpl()
    from hierarchicalprobabilistic3dhuman_amd import _capi
    P = _capi.ptr
    M = 6528
    betas = torch.randn(M, 10, device=dev)
    aa = torch.randn(M * 24, 3, device=dev) * 0.5
    R = rtu.batch_rodrigues(aa).view(M, 24, 3, 3)
    smpl.keep_intermediates = True
    smpl(betas=betas, body_pose=R[:, 1:].contiguous(), global_orient=R[:, :1].contiguous(), pose2rot=False)
    L = smpl._last
    s = _capi.stream()
    V = 6890
    t_blend = timeit(lambda: _capi.call("hps_smpl_blend", P(L["xt"]), P(smpl._bmat), P(smpl._v_template_flat),
                                        P(L["v_posed_raw"]), M, 3 * V, smpl._kp, L["xt"].shape[1], smpl._np, L["ldv"], s))
    ref = L["v_posed_raw"][:, :3 * V].clone()
    x_rm = L["xt"][:, :M].t().contiguous()                    # (M, kp)
    w_rm = smpl._bmat.t().contiguous()                        # (np, kp)
    npad = w_rm.shape[0]
    ones = torch.ones(npad, device=dev)
    shift = torch.zeros(npad, device=dev)
    shift[:3 * V] = smpl._v_template_flat
    out = torch.empty(M, npad, device=dev)
    for v, ab in ((1, 0), (2, 0), (3, 0)):
        fn = lambda: _capi.call("hps_conv2d_bn_act_pad", P(x_rm), P(w_rm), P(ones), P(shift), None, P(out), M, 1, 1, 0, smpl._kp, npad,
                                1, 1, 1, 0, 0, 0, 0, v, 1, None, s)
        _capi.call("hps_dev_conv_pad_ablate", ab)
        fn()
        t = timeit(fn)
        _capi.call("hps_dev_conv_pad_ablate", 0)
        print("blend: GEMM kernel %.3f ms (%.1f TF) | conv_pad variant %d dephase %d: %.3f ms (%.1f TF)  max diff %.2e" % (
            t_blend, 2.0 * M * 217 * 20670 / t_blend / 1e9, v, ab, t, 2.0 * M * 217 * 20670 / t / 1e9, err(out[:, :3 * V], ref)))


def sec_unc_modes():
    """hps_vertex_uncertainty kernel generations on the bench shape and a few others: time and bit equality."""
    from hierarchicalprobabilistic3dhuman_amd import _capi
    for (B, N) in ((64, 100), (64, 8), (16, 128), (3, 37)):
        v = torch.randn(B, N, 6890, 3, device=dev)
        outs = {}
        line = "unc B=%d N=%d:" % (B, N)
        for mode, name in ((1, "two-sweep"), (2, "lds128"), (3, "lds64"), (4, "registers"), (0, "auto")):
            _capi.call("hps_dev_unc_mode", mode)
            outs[name] = su.vertex_uncertainty(v)
            t = timeit(lambda: su.vertex_uncertainty(v), 10, 3)
            line += "  %s %.1f us (%.2f TB/s)" % (name, t * 1e3, v.numel() * 4 / t / 1e9)
        _capi.call("hps_dev_unc_mode", 0)
        print(line)
        print("    registers == lds128: %s, == lds64: %s, auto == registers: %s, max |two-sweep - registers| %.2e" % (
            bool(torch.equal(outs["registers"], outs["lds128"])), bool(torch.equal(outs["registers"], outs["lds64"])),
            bool(torch.equal(outs["auto"], outs["registers"])), float((outs["two-sweep"] - outs["registers"]).abs().max())))


def sec_lbs_tune():
    model, params, smpl = make_smpl()
    from hierarchicalprobabilistic3dhuman_amd import _capi
    P = _capi.ptr
    M, V, J = 6528, 6890, 24
    betas = torch.randn(M, 10, device=dev)
    aa = torch.randn(M * 24, 3, device=dev) * 0.5
    R = rtu.batch_rodrigues(aa).view(M, 24, 3, 3)
    smpl.keep_intermediates = True
    out = smpl(betas=betas, body_pose=R[:, 1:].contiguous(), global_orient=R[:, :1].contiguous(), pose2rot=False)
    L = smpl._last
    ref = out.vertices.clone()
    verts = torch.empty(M, V, 3, device=dev)
    s = _capi.stream()
    gb = 166896.0 * M / 1e9
    names = {0: "G4 VPT1", 1: "G8 VPT1", 2: "G4 VPT2", 3: "G2 VPT2", 4: "G2 VPT1", 5: "G8 VPT2", 6: "G16 VPT1"}
    for variant in (0, 1, 2, 5, 6):
        for tb in (4096, 6144, 8192, 12288, 16384, 24576, 49152):
            fn = lambda: _capi.call("hps_dev_lbs_variant", P(L["v_posed_raw"]), L["ldv"], P(L["a"]), _capi.iptr(smpl._w_idx),
                                    P(smpl._w_val), smpl._lbs_k, J, None, P(verts), M, V, variant, tb, s)
            verts.zero_()
            fn()
            ok = bool(torch.equal(verts, ref))
            t = timeit(fn, 20, 5)
            print("lbs variant %d (%s) target_blocks %5d: %.1f us  %.2f TB/s (%.1f%% of 8)  bitwise==default %s" % (
                variant, names[variant], tb, t * 1e3, gb / t, gb / t / 8 * 100, ok))


def sec_sampler():
    g = torch.Generator().manual_seed(1)
    for (B, N) in ((2, 4), (2, 100), (64, 1), (3, 1000)):
        F = torch.randn(B, 23, 3, 3, generator=g) * (3.0 if N != 1 else 0.3) + torch.eye(3)
        U, S, Vh = torch.linalg.svd(F)
        V = Vh.transpose(-1, -2).contiguous()
        torch.manual_seed(5)
        Rref, (eps, w, disc) = O.pose_matrix_fisher_sampling(U, S, V, N, return_noise=True)
        torch.manual_seed(5)
        t0 = time.time()
        R = su.pose_matrix_fisher_sampling_torch(U.to(dev), S.to(dev), V.to(dev), N, sample_on_cpu=True)
        torch.cuda.synchronize()
        d = (R.cpu() - Rref).abs().amax(dim=(-1, -2))
        print("sampler host-stream B=%d N=%d: max err %.2e, entries>1e-4: %d of %d, discarded rounds %d, %.1f ms" % (
            B, N, float(d.max()), int((d > 1e-4).sum()), d.numel(), int(disc.sum()), (time.time() - t0) * 1e3))
    # philox route
    B, N = 64, 100
    F = torch.randn(B, 23, 3, 3, generator=g) * 2 + torch.eye(3)
    U, S, Vh = torch.linalg.svd(F)
    V = Vh.transpose(-1, -2).contiguous()
    Ud, Sd, Vd = U.to(dev), S.to(dev), V.to(dev)
    R = su.pose_matrix_fisher_sampling_torch(Ud, Sd, Vd, N, seed=123)
    RtR = torch.matmul(R.transpose(-1, -2), R) - torch.eye(3, device=dev)
    print("philox: orth err %.2e det min %.6f" % (float(RtR.abs().max()), float(torch.det(R.cpu()).min())))
    R2a = su.pose_matrix_fisher_sampling_torch(Ud[:32], Sd[:32], Vd[:32], N, seed=123, image_offset=0)
    R2b = su.pose_matrix_fisher_sampling_torch(Ud[32:], Sd[32:], Vd[32:], N, seed=123, image_offset=32)
    print("philox sharding invariance (bitwise):", bool(torch.equal(R, torch.cat([R2a, R2b]))))
    print("philox B=64 N=100: %.3f ms" % timeit(lambda: su.pose_matrix_fisher_sampling_torch(Ud, Sd, Vd, N, seed=123)))
    # first-moment check against the mode direction: E[R] ~ U diag(d) V^T with d in (0,1): check U^T E[R] V is ~diagonal
    Nbig = 4000
    Rb = su.pose_matrix_fisher_sampling_torch(Ud[:2], Sd[:2], Vd[:2], Nbig, seed=7)
    Up, Sp, Vp = O.proper_svd(U[:2], S[:2], V[:2])
    Em = Rb.mean(dim=1).cpu()
    D = torch.matmul(Up.transpose(-1, -2), torch.matmul(Em, Vp))
    off = D - torch.diag_embed(torch.diagonal(D, dim1=-2, dim2=-1))
    print("philox E[R] off-diagonal max %.3f (MC sigma ~%.3f); diag sample" % (float(off.abs().max()), 1 / np.sqrt(Nbig)),
          D[0, 0].diagonal().tolist(), "S", Sp[0, 0].tolist())


def make_net():
    cfg = configs.get_cfg_defaults()
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, cfg).eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    return net.to(dev), sd


def sec_head():
    net, sd = make_net()
    g = torch.Generator().manual_seed(3)
    for B in (2, 64):
        feats = torch.rand(B, 512, generator=g) * 2
        ref = O.head_forward(sd, feats, configs.SMPL_PARENTS)
        out = net(None, input_feats=feats.to(dev))
        names = "F U S V mode".split()
        print("head B=%d:" % B, " ".join("%s %.2e" % (n, err(a, b)) for n, a, b in zip(names, out[:5], ref[:5])),
              "loc %.2e scale %.2e glob %.2e cam %.2e" % (err(out[5].loc, ref[5][0]), err(out[5].scale, ref[5][1]),
                                                           err(out[6], ref[6]), err(out[7], ref[7])))
        fd = feats.to(dev)
        print("head B=%d time %.3f ms" % (B, timeit(lambda: net(None, input_feats=fd), 5, 2)))


def sec_encoder():
    net, sd = make_net()
    import torch.nn.functional as F
    enc = net.image_encoder
    prep = enc.prepare()
    from hierarchicalprobabilistic3dhuman_amd import _capi
    P = _capi.ptr
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 18, 256, 256, generator=g)
    # stem only
    xh = torch.empty(2, 256, 256, 20, device=dev)
    _capi.call("hps_nchw_to_nhwc", P(x.to(dev)), P(xh), 2, 18, 256, 256, 20, _capi.stream())
    print("nhwc", err(xh[..., :18].permute(0, 3, 1, 2), x), float(xh[..., 18:].abs().max()))
    y = prep["stem"](xh, relu=True)
    ref = F.relu(O._bn(F.conv2d(x, sd["image_encoder.conv1.weight"], stride=2, padding=3), sd, "image_encoder.bn1"))
    print("stem", err(y.permute(0, 3, 1, 2), ref), "ref max", float(ref.abs().max()))
    yp = torch.empty(2, 64, 64, 64, device=dev)
    _capi.call("hps_maxpool3x3s2", P(y), P(yp), 2, 128, 128, 64, _capi.stream())
    refp = F.max_pool2d(ref, 3, 2, 1)
    print("maxpool", err(yp.permute(0, 3, 1, 2), refp))
    # random conv checks for each tile config / stride / 1x1
    for (B, H, Cin, Cout, k, s, p) in ((2, 64, 64, 64, 3, 1, 1), (2, 64, 64, 128, 3, 2, 1), (2, 64, 64, 128, 1, 2, 0),
                                       (64, 8, 512, 512, 3, 1, 1), (3, 16, 256, 256, 3, 1, 1), (1, 8, 512, 512, 3, 1, 1)):
        conv = torch.nn.Conv2d(Cin, Cout, k, s, p, bias=False)
        bn = torch.nn.BatchNorm2d(Cout).eval()
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
        xin = torch.randn(B, Cin, H, H)
        res = torch.randn(B, Cout, (H + 2 * p - k) // s + 1, (H + 2 * p - k) // s + 1)
        with torch.no_grad():
            r = F.relu(bn(conv(xin)) + res)
        from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN
        cb = _ConvBN(conv.to(dev), bn.to(dev))
        yo = cb(xin.to(dev).permute(0, 2, 3, 1).contiguous(), residual=res.to(dev).permute(0, 2, 3, 1).contiguous())
        print("conv B%d H%d %d->%d k%d s%d: err %.2e (ref max %.1f)" % (B, H, Cin, Cout, k, s, err(yo.permute(0, 3, 1, 2), r), float(r.abs().max())))
    feats_ref = O.resnet18_forward(sd, x)
    feats = enc(x.to(dev))
    print("encoder feats err %.2e rel %.2e" % (err(feats, feats_ref), err(feats, feats_ref) / float(feats_ref.abs().max())))


def sec_encoder_perf():
    net, sd = make_net()
    enc = net.image_encoder
    x = torch.rand(64, 18, 256, 256, device=dev)
    t = timeit(lambda: enc(x), 5, 2)
    print("encoder B=64: %.3f ms  (%.1f TFLOP/s on 6.279 GFLOP/img)" % (t, 64 * 6.279 / t))
    # per-layer timing
    prep = enc._prepared
    xh = torch.empty(64, 256, 256, 20, device=dev)
    from hierarchicalprobabilistic3dhuman_amd import _capi
    P = _capi.ptr
    t = timeit(lambda: _capi.call("hps_nchw_to_nhwc", P(x), P(xh), 64, 18, 256, 256, 20, _capi.stream()))
    print("  nchw->nhwc %.3f ms" % t)
    t = timeit(lambda: prep["stem"](xh), 5, 2)
    print("  stem conv7x7 %.3f ms (%.1f TF)" % (t, 64 * 2 * 0.925 / t))
    y = prep["stem"](xh)
    yp = torch.empty(64, 64, 64, 64, device=dev)
    t = timeit(lambda: _capi.call("hps_maxpool3x3s2", P(y), P(yp), 64, 128, 128, 64, _capi.stream()))
    print("  maxpool %.3f ms" % t)
    cur = yp
    for bi, (c1, c2, down) in enumerate(prep["blocks"]):
        t1 = timeit(lambda: c1(cur), 5, 2)
        o = c1(cur)
        idn = down(cur, relu=False) if down is not None else cur
        t2 = timeit(lambda: c2(o, residual=idn), 5, 2)
        fl1 = 2.0 * o.numel() * c1.kh * c1.kw * c1.cin_p / 1e9
        o2 = c2(o, residual=idn)
        fl2 = 2.0 * o2.numel() * c2.kh * c2.kw * c2.cin_p / 1e9
        td = timeit(lambda: down(cur, relu=False), 5, 2) if down is not None else 0.0
        print("  block %d: conv1 %.3f ms (%.1f TF) conv2 %.3f ms (%.1f TF) down %.3f ms" % (bi, t1, fl1 / t1, t2, fl2 / t2, td))
        cur = o2


def sec_encoder_pad():
    """Halo-padded generation (csrc/conv_pad.hip): whole encoder against the plain layout, then per layer."""
    net, sd = make_net()
    enc = net.image_encoder
    x = torch.rand(64, 18, 256, 256, device=dev)
    f0 = devlib.plain_forward(enc, x)
    t0 = timeit(lambda: devlib.plain_forward(enc, x), 5, 2)
    f1 = enc(x)
    t1 = timeit(lambda: enc(x), 5, 2)
    print("encoder B=64: plain %.3f ms, padded %.3f ms (%.1f TFLOP/s on 6.279 GFLOP/img); feats diff %.2e (max |f| %.2e)" % (
        t0, t1, 64 * 6.279 / t1, err(f0, f1), float(f0.abs().max())))
    prep = enc._prepared
    # this section times the DIRECT kernels layer by layer on a plain padded NHWC input frame and a materialised stem output: select the
    # direct stem + separate max pool (with the product defaults -- Winograd stem, pool in its epilogue -- fs["stem"] is None and fs["in"]
    # holds phase frames or nothing: ADVICE r5)
    enc.fused_pool = False
    enc.set_winograd(False)
    prep = enc._prepared
    fs = enc._frame_set(prep, 64, 18, 256, 256, x.device)
    from hierarchicalprobabilistic3dhuman_amd import _capi
    P = _capi.ptr
    t = timeit(lambda: _capi.call("hps_nchw_to_padded_nhwc", P(x), P(fs["in"]), 64, 18, 256, 256, 3, _capi.stream()))
    print("  nchw->padded nhwc %.3f ms" % t)
    stem = prep["stem"]
    for v in (0, 1, 2, 3, 4):
        stem.variant = v
        t = timeit(lambda: stem.padded(fs["in"], 3, fs["stem"], 0), 5, 2)
        print("  stem row-mode variant %d: %.3f ms (%.1f TF real, %.1f TF issued)" % (v, t, 64 * 2 * 0.925 / t, 64 * 2 * 0.925 * 896 / 882 / t))
    stem.variant = 0
    t = timeit(lambda: _capi.call("hps_maxpool3x3s2_pad", P(fs["stem"]), P(fs["pool"]), 64, 128, 128, 64, 1, _capi.stream()))
    print("  maxpool %.3f ms" % t)
    cur = fs["pool"]
    for bi, ((c1, c2, down), ent) in enumerate(zip(prep["blocks"], fs["blocks"])):
        t1 = timeit(lambda: c1.padded(cur, 1, ent["c1"], 1, ws=ent["ws"]), 5, 2)
        idn = down.padded(cur, 1, ent["down"], 1, relu=False) if down is not None else cur
        td = timeit(lambda: down.padded(cur, 1, ent["down"], 1, relu=False), 5, 2) if down is not None else 0.0
        t2 = timeit(lambda: c2.padded(ent["c1"], 1, ent["c2"], 1, residual=idn, ws=ent["ws"]), 5, 2)
        Ho = ent["c1"].shape[1] - 2
        fl1 = 2.0 * 64 * Ho * Ho * c1.cout * c1.kh * c1.kw * c1.cin_p / 1e9
        fl2 = 2.0 * 64 * Ho * Ho * c2.cout * c2.kh * c2.kw * c2.cin_p / 1e9
        print("  block %d: conv1 %.3f ms (%.1f TF) conv2 %.3f ms (%.1f TF) down %.3f ms" % (bi, t1, fl1 / t1, t2, fl2 / t2, td))
        cur = ent["c2"]


def sec_pad_ablate():
    """conv_pad kernel per layer shape: full vs no-epilogue, per tile variant."""
    from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN
    from hierarchicalprobabilistic3dhuman_amd import _capi
    import torch.nn.functional as F
    B = 64
    for (H, Cin, Cout, k, st, pd) in [(64, 64, 64, 3, 1, 1), (64, 64, 128, 3, 2, 1), (32, 128, 128, 3, 1, 1), (16, 256, 256, 3, 1, 1),
                                      (8, 512, 512, 3, 1, 1)]:
        conv = torch.nn.Conv2d(Cin, Cout, k, st, pd, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
        cb = _ConvBN(conv, bn)
        xp = F.pad(torch.relu(torch.randn(B, H, H, Cin, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
        Ho = (H + 2 * pd - k) // st + 1
        out = torch.zeros(B, Ho + 2, Ho + 2, Cout, device=dev)
        res = torch.randn(B, Ho + 2, Ho + 2, Cout, device=dev)
        fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin / 1e9
        line = "pad H%d %d->%d s%d:" % (H, Cin, Cout, st)
        for v in (1, 2, 3, 4):
            if (v == 1 and Cout % 128) or (v == 4 and Cout != 64):
                continue
            cb.variant = v
            ts = []
            for ab in (0, 1):
                _capi.call("hps_dev_conv_pad_ablate", ab)
                ts.append(timeit(lambda: cb.padded(xp, 1, out, 1, residual=res), 10, 3))
            _capi.call("hps_dev_conv_pad_ablate", 0)
            line += "  v%d %.0f us (%.0f TF) no-epi %.0f us" % (v, ts[0] * 1e3, fl / ts[0], ts[1] * 1e3)
        print(line)


def sec_pad_splitk():
    """conv_pad kernel: split-K factors per ResNet-18 layer shape at B=64 (tile variant automatic)."""
    from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN
    import torch.nn.functional as F
    B = 64
    for (H, Cin, Cout, k, st, pd) in [(64, 64, 128, 3, 2, 1), (32, 128, 128, 3, 1, 1), (32, 128, 256, 3, 2, 1), (16, 256, 256, 3, 1, 1),
                                      (16, 256, 512, 3, 2, 1), (8, 512, 512, 3, 1, 1)]:
        conv = torch.nn.Conv2d(Cin, Cout, k, st, pd, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
        cb = _ConvBN(conv, bn)
        xp = F.pad(torch.relu(torch.randn(B, H, H, Cin, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
        Ho = (H + 2 * pd - k) // st + 1
        out = torch.zeros(B, Ho + 2, Ho + 2, Cout, device=dev)
        res = torch.randn(B, Ho + 2, Ho + 2, Cout, device=dev)
        fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin / 1e9
        line = "pad H%d %d->%d s%d (%d px/img):" % (H, Cin, Cout, st, Ho * Ho)
        for ks in (1, 2, 3, 4, 6):
            if (k * k * Cin // 32) % ks:
                continue
            cb.ksplit = ks
            ws = torch.empty(ks, B * Ho * Ho, Cout, device=dev) if ks > 1 else None
            t = timeit(lambda: cb.padded(xp, 1, out, 1, residual=res, ws=ws), 10, 3)
            line += "  K/%d %.0f us (%.0f TF)" % (ks, t * 1e3, fl / t)
        print(line)


def sec_conv_tune():
    """Every distinct convolution of ResNet-18 at B=64 under the v1 kernel and the three v2 tile shapes."""
    from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN
    shapes = [  # H, Cin, Cout, k, stride, pad
        (64, 64, 64, 3, 1, 1), (64, 64, 128, 3, 2, 1), (64, 64, 128, 1, 2, 0), (32, 128, 128, 3, 1, 1),
        (32, 128, 256, 3, 2, 1), (32, 128, 256, 1, 2, 0), (16, 256, 256, 3, 1, 1), (16, 256, 512, 3, 2, 1),
        (16, 256, 512, 1, 2, 0), (8, 512, 512, 3, 1, 1)]
    B = 64
    for (H, Cin, Cout, k, st, pd) in shapes:
        conv = torch.nn.Conv2d(Cin, Cout, k, st, pd, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
        cb = _ConvBN(conv, bn)
        x = torch.relu(torch.randn(B, H, H, Cin, device=dev))     # post-ReLU statistics like the real activations
        Ho = (H + 2 * pd - k) // st + 1
        res = torch.randn(B, Ho, Ho, Cout, device=dev)
        fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin / 1e9
        line = "conv H%d %d->%d k%d s%d (%.1f GFLOP):" % (H, Cin, Cout, k, st, fl)
        ref = None
        for v in (11, 13, 14, 102, 104):
            if v % 10 == 1 and Cout % 128:
                continue
            if x.numel() >= 2 ** 31:
                continue
            if v >= 100:
                if Cout % 128 or (k * k * Cin // 32) % (v - 100) or (B * Ho * Ho) % 128:
                    continue
                cb.kernel, cb.variant, cb.ksplit = "v3", 0, v - 100
            else:
                cb.ksplit = 1
                cb.kernel, cb.variant = ("v1", 0) if v < 0 else (("v2", v) if v < 10 else ("v3", v - 10))
            y = cb(x, residual=res)
            if ref is None:
                ref = y
            e = float((y - ref).abs().max())
            t = timeit(lambda: cb(x, residual=res), 10, 3)
            line += "  %s %.0f us (%.0f TF, d=%.1e)" % ("v1" if v < 0 else ("v2/%d" % v if v < 10 else ("v3/%d" % (v - 10) if v < 100 else "splitK%d" % (v - 100))), t * 1e3, fl / t, e)
        print(line)


def sec_conv_ablate():
    """v3 kernel with the DMA (variant 2x) or the MFMAs (variant 3x) removed: where does the time go?"""
    from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN
    B = 64
    for (H, Cin, Cout, k, st, pd) in [(32, 128, 128, 3, 1, 1), (64, 64, 64, 3, 1, 1), (8, 512, 512, 3, 1, 1)]:
        conv = torch.nn.Conv2d(Cin, Cout, k, st, pd, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
        cb = _ConvBN(conv, bn)
        x = torch.relu(torch.randn(B, H, H, Cin, device=dev))
        Ho = (H + 2 * pd - k) // st + 1
        fl = 2.0 * B * Ho * Ho * Cout * k * k * Cin / 1e9
        line = "ablate H%d %d->%d:" % (H, Cin, Cout)
        cb.kernel, cb.variant = "v3", 3
        ref = cb(x)
        for v in (1, 21, 31, 41, 51, 3, 23, 33, 43, 4, 44):
            if v % 10 == 1 and Cout % 128:
                continue
            if v % 10 == 4 and Cout != 64:
                continue
            cb.kernel, cb.variant = "v3", v
            d = float((cb(x) - ref).abs().max()) if v in (1, 3, 4, 41, 43, 44) else -1.0
            t = timeit(lambda: cb(x), 10, 3)
            line += "  v3/%d %.0f us (%.0f TF-equiv, d=%.0e)" % (v, t * 1e3, fl / t, d)
        print(line)


def sec_e2e():
    model, params, smpl = make_smpl()
    net, sd = make_net()
    g = torch.Generator().manual_seed(0)
    B, N = 2, 4
    x = torch.rand(B, 18, 256, 256, generator=g)
    torch.manual_seed(11)
    ref = O.infer(sd, params, configs.SMPL_PARENTS, x, N)
    torch.manual_seed(11)
    out = infer(net, smpl, x.to(dev), num_samples=N, sample_on_cpu=True)
    for k in ("pose_F", "pose_S", "pose_rotmats_mode", "shape_loc", "glob_rotmats", "verts_mode", "joints_mode",
              "verts_tpose", "R_samples", "verts_samples", "joints_samples", "unc"):
        print("e2e %s: %.2e" % (k, err(out[k], ref[k])))
    xb = torch.rand(64, 18, 256, 256, device=dev)
    for Ns in (100,):
        t = timeit(lambda: infer(net, smpl, xb, num_samples=Ns, seed=1), 3, 1)
        print("e2e B=64 N=%d: %.2f ms -> %.0f img/s" % (Ns, t, 64 / t * 1e3))


if __name__ == "__main__":
    from hierarchicalprobabilistic3dhuman_amd import _capi as _capi_main
    with _capi_main.dev_library():          # every section runs on libhps_dev.so (tuning switches, earlier generations, ablations)
        for name in sys.argv[1:]:
            print("==== %s ====" % name, flush=True)
            t0 = time.time()
            globals()["sec_" + name]()
            print("---- %s done in %.1f s" % (name, time.time() - t0), flush=True)

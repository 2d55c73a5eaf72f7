"""GPU parity: SMPL forward (pose prep / blend GEMM / LBS / joints / uncertainty kernels behind the C ABI)
against the CPU oracle, the float64 twin, analytic known answers, and size-independent properties at the
full BASELINE configs[1] size (6528 meshes).

Stated fp32 tolerances: vertices and joints <= 2e-5 m given identical (R, beta)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from oracle.smpl_np64 import smpl_forward64, rodrigues64
from hierarchicalprobabilistic3dhuman_amd import configs, _capi
from hierarchicalprobabilistic3dhuman_amd import sampling_utils as su
from conftest import maxerr, smplx_golden_models

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _pose(M, seed, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(M, 10, generator=g), torch.randn(M, 24, 3, generator=g) * scale, torch.randn(M, 3, generator=g)


@pytest.mark.parametrize("M", [1, 2, 37, 128, 129, 300])
def test_rotmat_route_matches_oracle(M, dev, smpl_gpu, smpl_assets):
    p = smpl_assets[2]
    betas, aa, transl = _pose(M, M)
    R = O.batch_rodrigues(aa.view(-1, 3)).view(M, 24, 3, 3)
    ref = O.smpl_forward(p, betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False, transl=transl,
                         return_intermediates=True)
    args = dict(betas=betas.to(dev), body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), pose2rot=False,
                transl=transl.to(dev))
    out = smpl_gpu(**args)                                    # product path: fused blend + skinning kernel
    smpl_gpu.keep_intermediates, smpl_gpu.fused_mesh = True, False
    try:
        out_unfused = smpl_gpu(**args)                        # unfused pair: exposes v_posed
        L = smpl_gpu._last
    finally:
        smpl_gpu.keep_intermediates, smpl_gpu.fused_mesh = False, True
    assert torch.equal(out.vertices, out_unfused.vertices) and torch.equal(out.joints, out_unfused.joints)
    assert maxerr(L["v_posed"], ref["v_posed"]) <= TOL
    assert maxerr(L["a"].view(M, 24, 3, 4), ref["A"][:, :, :3, :]) <= TOL
    assert maxerr(L["j_posed"], ref["J_posed"]) <= TOL
    assert maxerr(out.vertices, ref["vertices"]) <= TOL
    assert maxerr(out.joints, ref["joints"]) <= TOL
    assert out.joints.shape == (M, 90, 3) and out.vertices.shape == (M, 6890, 3)


@pytest.mark.parametrize("M", [1, 5, 130])
def test_axis_angle_route_matches_oracle_and_float64(M, dev, smpl_gpu, smpl_assets):
    model, extra, p = smpl_assets
    betas, aa, _ = _pose(M, 100 + M, scale=0.8)
    ref = O.smpl_forward(p, betas=betas, body_pose=aa[:, 1:].reshape(M, 69), global_orient=aa[:, 0])
    out = smpl_gpu(betas=betas.to(dev), body_pose=aa[:, 1:].reshape(M, 69).to(dev), global_orient=aa[:, 0].to(dev))
    assert maxerr(out.vertices, ref["vertices"]) <= TOL and maxerr(out.joints, ref["joints"]) <= TOL
    R64 = rodrigues64(aa.reshape(-1, 3).double().numpy()).reshape(M, 24, 3, 3)
    v64, j64 = smpl_forward64(model, extra, configs.SMPLX_EXTRA_VERTEX_IDS, betas.double().numpy(), R64)
    assert np.abs(out.vertices.cpu().numpy() - v64).max() <= TOL
    assert np.abs(out.joints.cpu().numpy() - j64).max() <= TOL


def test_default_arguments_and_beta_expansion(dev, smpl_gpu, smpl_assets):
    p = smpl_assets[2]
    b1 = torch.randn(1, 10, generator=torch.Generator().manual_seed(9))
    # smpl(betas=...) with the zero module pose (predict/...:136)
    assert maxerr(smpl_gpu(betas=b1.to(dev)).vertices, O.smpl_forward(p, betas=b1)["vertices"]) <= TOL
    # zero pose, zero betas -> template
    assert maxerr(smpl_gpu().vertices[0], p.v_template) <= 1e-6
    # one row of betas expanded to the pose batch
    _, aa, _ = _pose(3, 5)
    ref = O.smpl_forward(p, betas=b1, body_pose=aa[:, 1:].reshape(3, 69), global_orient=aa[:, 0])
    out = smpl_gpu(betas=b1.to(dev), body_pose=aa[:, 1:].reshape(3, 69).to(dev), global_orient=aa[:, 0].to(dev))
    assert maxerr(out.vertices, ref["vertices"]) <= TOL
    assert out.betas.shape[0] == 3


def test_argument_errors(dev, smpl_gpu):
    with pytest.raises(ValueError):
        smpl_gpu(betas=torch.zeros(2, 10, device=dev), body_pose=torch.zeros(2, 23, 3, 3, device=dev),
                 global_orient=torch.zeros(2, 3, device=dev), pose2rot=False)
    with pytest.raises(ValueError):
        smpl_gpu(betas=torch.zeros(2, 9, device=dev), body_pose=torch.zeros(2, 69, device=dev),
                 global_orient=torch.zeros(2, 3, device=dev))


def test_dense_skin_weights_use_wider_kernel(dev, smpl_assets):
    """A model with more than 4 influences per vertex takes the K=8 / K=24 instantiations."""
    from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
    model, extra, _ = smpl_assets
    rs = np.random.RandomState(1)
    for nnz, K in ((7, 8), (24, 24)):
        m2 = dict(model)
        W = np.zeros((6890, 24))
        for v in range(6890):
            idx = rs.choice(24, nnz, replace=False)
            w = rs.uniform(0.1, 1, nnz)
            W[v, idx] = w / w.sum()
        m2["weights"] = W
        smpl = SMPL(m2).to(dev)
        assert smpl._lbs_k == K
        p2 = O.SMPLParams(m2, extra, configs.SMPLX_EXTRA_VERTEX_IDS)
        betas, aa, _ = _pose(3, 77)
        ref = O.smpl_forward(p2, betas=betas, body_pose=aa[:, 1:].reshape(3, 69), global_orient=aa[:, 0])
        out = smpl(betas=betas.to(dev), body_pose=aa[:, 1:].reshape(3, 69).to(dev), global_orient=aa[:, 0].to(dev))
        assert maxerr(out.vertices, ref["vertices"]) <= TOL
        smpl.fused_mesh = False                                  # the unfused pair's K = 8 / 24 instantiations: same bits
        out_u = smpl(betas=betas.to(dev), body_pose=aa[:, 1:].reshape(3, 69).to(dev), global_orient=aa[:, 0].to(dev))
        assert torch.equal(out.vertices, out_u.vertices)


def test_full_size_properties_6528_meshes(dev, smpl_gpu, smpl_assets):
    """BASELINE configs[1] mesh count (64 x (100 + 2)); the oracle is too slow here, so check properties:
    rigid global rotation, translation equivariance, batch-position independence, one-hot consistency."""
    M = 6528
    g = torch.Generator().manual_seed(42)
    betas = torch.randn(M, 10, generator=g).to(dev)
    aa = (torch.randn(M, 24, 3, generator=g) * 0.5).to(dev)
    from hierarchicalprobabilistic3dhuman_amd.rigid_transform_utils import batch_rodrigues
    R = batch_rodrigues(aa.view(-1, 3)).view(M, 24, 3, 3)
    out = smpl_gpu(betas=betas, body_pose=R[:, 1:].contiguous(), global_orient=R[:, :1].contiguous(), pose2rot=False)
    assert torch.isfinite(out.vertices).all() and torch.isfinite(out.joints).all()
    # (a) a random subset recomputed in a small batch gives the same meshes (no cross-mesh coupling, tile edges)
    idx = torch.tensor([0, 1, 127, 128, 129, 3000, 6400, 6527])
    sub = smpl_gpu(betas=betas[idx], body_pose=R[idx, 1:].contiguous(), global_orient=R[idx, :1].contiguous(), pose2rot=False)
    assert maxerr(sub.vertices, out.vertices[idx]) == 0.0 and maxerr(sub.joints, out.joints[idx]) == 0.0
    # (b) those meshes against the oracle
    ref = O.smpl_forward(smpl_assets[2], betas=betas[idx].cpu(), body_pose=R[idx, 1:].cpu(), global_orient=R[idx, :1].cpu(), pose2rot=False)
    assert maxerr(sub.vertices, ref["vertices"]) <= TOL
    # (c) pre-multiplying the global rotation by Q rotates the mesh rigidly about the root joint
    Q = batch_rodrigues(torch.tensor([[0.3, -1.1, 0.4]], device=dev))[0]
    out_q = smpl_gpu(betas=betas[:256], body_pose=R[:256, 1:].contiguous(),
                     global_orient=torch.matmul(Q, R[:256, :1]).contiguous(), pose2rot=False)
    J0 = out.joints[:256, :1]
    want = torch.einsum("ij,bvj->bvi", Q, out.vertices[:256] - J0) + J0
    assert maxerr(out_q.vertices, want) <= TOL
    # (d) translation equivariance
    t = torch.randn(256, 3, generator=g).to(dev)
    out_t = smpl_gpu(betas=betas[:256], body_pose=R[:256, 1:].contiguous(), global_orient=R[:256, :1].contiguous(),
                     pose2rot=False, transl=t)
    assert maxerr(out_t.vertices, out.vertices[:256] + t[:, None]) <= 1e-6
    assert maxerr(out_t.joints[:, :45], out.joints[:256, :45] + t[:, None]) <= 2e-6
    # the extra / cocoplus / h36m regressor rows do not sum to exactly 1 (0.9998..1.0) and, like the reference
    # (models/smpl_official.py:30-32), they are applied to the translated vertices: check the definition instead
    reg = torch.cat([smpl_gpu.J_regressor_extra, smpl_gpu.J_regressor_cocoplus, smpl_gpu.J_regressor_h36m])
    assert maxerr(out_t.joints[:, 45:], torch.einsum("jv,bvk->bjk", reg, out_t.vertices)) <= TOL


def test_blend_kernels_agree_bitwise(dev, smpl_gpu):
    """The opt-in stationary-A blend GEMM (mesh fragments resident in registers, bmat streamed) gives the bits of the
    default tiled kernel, for ragged mesh counts too."""
    g = torch.Generator().manual_seed(11)
    for M in (1, 130, 700):
        betas = torch.randn(M, 10, generator=g).to(dev)
        pose = (torch.randn(M, 72, generator=g) * 0.4).to(dev)
        smpl_gpu.fused_mesh = False
        try:
            ref = smpl_gpu(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3]).vertices.clone()
            with _capi.dev_library():            # alternate kernels and their switches live in libhps_dev.so only
                try:
                    _capi.call("hps_dev_blend_mode", 2)
                    got = smpl_gpu(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3]).vertices
                finally:
                    _capi.call("hps_dev_blend_mode", 0)
        finally:
            smpl_gpu.fused_mesh = True
        assert torch.equal(got, ref), M


@pytest.mark.parametrize("M", [1, 63, 64, 65, 130, 700, 6528])
def test_fused_mesh_kernel_gives_the_bits_of_blend_plus_lbs(M, dev, smpl_gpu):
    """hps_smpl_mesh_fused (GEMM tile skinned in the MFMA epilogue, no v_posed in HBM) against hps_smpl_blend followed by
    hps_smpl_lbs -- the unfused definition SURVEY 8(d) counts the LBS bytes on: same MFMA k order and the same
    skin_vertex arithmetic, so vertices and joints are bit-identical, for ragged mesh tiles and with a translation."""
    g = torch.Generator().manual_seed(1000 + M)
    betas = torch.randn(M, 10, generator=g).to(dev)
    pose = (torch.randn(M, 72, generator=g) * 0.5).to(dev)
    transl = torch.randn(M, 3, generator=g).to(dev) if M % 2 else None
    args = dict(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3], transl=transl)
    fused = smpl_gpu(**args)
    smpl_gpu.fused_mesh = False
    try:
        unfused = smpl_gpu(**args)
    finally:
        smpl_gpu.fused_mesh = True
    assert torch.isfinite(fused.vertices).all()
    assert torch.equal(fused.vertices, unfused.vertices), float((fused.vertices - unfused.vertices).abs().max())
    assert torch.equal(fused.joints, unfused.joints)


def test_vertex_uncertainty_kernel(dev):
    g = torch.Generator().manual_seed(3)
    v = torch.randn(3, 17, 6890, 3, generator=g)
    got = su.vertex_uncertainty(v.to(dev))
    want = torch.stack([O.vertex_uncertainty(v[i]) for i in range(3)])
    assert maxerr(got, want) <= 1e-5
    assert maxerr(su.vertex_uncertainty(v[:, :1].contiguous().to(dev)), torch.zeros(3, 6890)) == 0.0     # N = 1: zero spread
    # N <= 128 takes the register-resident single pass (2, 4, 8, 13 or 16 samples per lane), larger N the two-sweep kernel;
    # the LDS-resident forms it replaced give the same bits (same summation order)
    for n in (8, 9, 31, 64, 100, 102, 120, 128, 130):
        vv = torch.randn(2, n, 6890, 3, generator=g)
        want_n = torch.stack([O.vertex_uncertainty(vv[i]) for i in range(2)])
        got_n = su.vertex_uncertainty(vv.to(dev))
        assert maxerr(got_n, want_n) <= 1e-5, n
        if n <= 100:
            with _capi.dev_library():
                try:
                    for mode in (2, 3):
                        _capi.call("hps_dev_unc_mode", mode)
                        assert torch.equal(su.vertex_uncertainty(vv.to(dev)), got_n), (n, mode)
                finally:
                    _capi.call("hps_dev_unc_mode", 0)


def test_vertex_uncertainty_one_sweep_kernel(dev):
    """128 < N <= 1024 (BASELINE configs[4]: N = 1000): the one-sweep kernel (rows in registers, the first row blocks DMA'd into
    LDS) against the oracle and against the two-sweep kernel it replaces (dev library, mode 1; another summation order, so
    rounding-level agreement, not bits).  Sizes around every dispatch and masking boundary: full and partial last row blocks,
    rows-per-lane 8 / 16 / 24 / 32, a vertex count that is not a multiple of the 16-vertex chunk, several images per launch."""
    g = torch.Generator().manual_seed(5)
    for n, nv, b in ((129, 6890, 2), (160, 6890, 1), (255, 200, 3), (256, 6890, 1), (257, 77, 2), (511, 100, 2), (513, 6890, 1),
                     (768, 33, 2), (769, 50, 1), (1000, 6890, 2), (1023, 40, 2), (1024, 16, 3)):
        vv = torch.randn(b, n, nv, 3, generator=g) * 0.3 + torch.randn(b, 1, nv, 3, generator=g)
        want = torch.stack([O.vertex_uncertainty(vv[i]) for i in range(b)])
        x = vv.to(dev)
        got = su.vertex_uncertainty(x)
        assert maxerr(got, want) <= 1e-5, (n, nv)
        assert torch.equal(su.vertex_uncertainty(x), got)                       # deterministic
        # per-image results do not depend on what else is in the launch
        assert torch.equal(su.vertex_uncertainty(x[b - 1:].contiguous())[0], got[b - 1])
        with _capi.dev_library():
            try:
                _capi.call("hps_dev_unc_mode", 1)
                two = su.vertex_uncertainty(x)
                if n in (1000, 1024):
                    _capi.call("hps_dev_unc_mode", 5)                            # 32-vertex chunks (another reduction tree)
                    assert maxerr(su.vertex_uncertainty(x), got) <= 2e-6, n
            finally:
                _capi.call("hps_dev_unc_mode", 0)
        assert maxerr(two, got) <= 2e-6, (n, nv)
    big = torch.randn(1, 1025, 64, 3, generator=g)                              # beyond the register-resident range: two sweeps
    assert maxerr(su.vertex_uncertainty(big.to(dev)), O.vertex_uncertainty(big[0])[None]) <= 1e-5


@pytest.mark.parametrize("J", [22, 24, 7, 32])
def test_fused_mesh_kernel_with_other_joint_counts(J, dev, smpl_gpu):
    """hps_smpl_mesh_fused through the C ABI with a run-time joint count -- J = 22 and 7 are not multiples of 4, so
    the 768 J byte halves of the skinning transforms are not whole 1 KiB DMA pieces and the last piece is cut by the
    per-lane mask -- against hps_smpl_blend + hps_smpl_lbs on the same operands, bit for bit, with and without translation,
    for mesh counts that leave ragged tiles."""
    g = torch.Generator().manual_seed(J)
    V, kp, K = smpl_gpu.num_verts, smpl_gpu._kp, 4
    P = _capi.ptr
    w_idx = (torch.randint(0, J, (V, K), generator=g, dtype=torch.int32)).to(dev)
    w_val = torch.rand(V, K, generator=g)
    w_val = (w_val / w_val.sum(1, keepdim=True)).to(dev)
    for M in (1, 33, 64, 200):
        mp = _capi.query_workspace(_capi.WS_SMPL_MP, M)
        xt = (torch.randn(kp, mp, generator=g) * 0.1).to(dev)
        a = torch.randn(M, J, 12, generator=g).to(dev)
        transl = torch.randn(M, 3, generator=g).to(dev)
        for tr in (None, transl):
            fused = torch.zeros(M, V, 3, device=dev)
            _capi.call("hps_smpl_mesh_fused", P(xt), P(smpl_gpu._bmat_p), P(smpl_gpu._v_template_flat), P(a), _capi.iptr(w_idx), P(w_val),
                       K, J, P(tr) if tr is not None else None, P(fused), M, V, kp, mp, smpl_gpu._np_fused, _capi.stream())
            ldv = smpl_gpu._np
            v_posed = torch.empty(M, ldv, device=dev)
            _capi.call("hps_smpl_blend", P(xt), P(smpl_gpu._bmat), P(smpl_gpu._v_template_flat), P(v_posed), M, smpl_gpu._N, kp, mp,
                       smpl_gpu._np, ldv, _capi.stream())
            want = torch.zeros(M, V, 3, device=dev)
            _capi.call("hps_smpl_lbs", P(v_posed), ldv, P(a), _capi.iptr(w_idx), P(w_val), K, J, P(tr) if tr is not None else None,
                       P(want), M, V, _capi.stream())
            assert torch.isfinite(fused).all()
            assert torch.equal(fused, want), (J, M, tr is not None, float((fused - want).abs().max()))
    # combinations that are not instantiated (they would need scratch memory) are refused, not silently slow
    with pytest.raises(_capi.HpsError):
        _capi.call("hps_smpl_mesh_fused", P(xt), P(smpl_gpu._bmat_p), P(smpl_gpu._v_template_flat), P(a), _capi.iptr(w_idx), P(w_val),
                   24, 24, None, P(fused), M, V, kp, mp, smpl_gpu._np_fused, _capi.stream())
    if J != 24:
        with pytest.raises(_capi.HpsError):
            _capi.call("hps_smpl_mesh_fused", P(xt), P(smpl_gpu._bmat_p), P(smpl_gpu._v_template_flat), P(a), _capi.iptr(w_idx), P(w_val),
                       8, J, None, P(fused), M, V, kp, mp, smpl_gpu._np_fused, _capi.stream())


def test_c_abi_rejects_unsupported_k(dev, smpl_gpu):
    z = torch.zeros(16, device=dev)
    zi = torch.zeros(16, device=dev, dtype=torch.int32)
    with pytest.raises(_capi.HpsError):
        _capi.call("hps_smpl_lbs", _capi.ptr(z), 3, _capi.ptr(z), _capi.iptr(zi), _capi.ptr(z), 5, 24, None, _capi.ptr(z), 1, 1,
                   _capi.stream())
    with pytest.raises(_capi.HpsError):
        _capi.call("hps_smpl_mesh_fused", _capi.ptr(z), _capi.ptr(z), _capi.ptr(z), _capi.ptr(z), _capi.iptr(zi), _capi.ptr(z),
                   5, 24, None, _capi.ptr(z), 1, 1, 16, 64, 384, _capi.stream())


def _pose_prep(name, smpl, g, b, is_rotmat, be, M, dev, J=None, parents=None, depth=None):
    """hps_smpl_pose_prep (or its first generation in the dev library) through the C ABI -> (xt, a, j_posed, rot)."""
    P = _capi.ptr
    J = smpl.NUM_JOINTS if J is None else J
    mp = _capi.query_workspace(_capi.WS_SMPL_MP, M)
    kp = smpl._kp
    xt = torch.full((kp, mp), -3.0, device=dev)
    a = torch.full((M, J, 12), -3.0, device=dev)
    jp = torch.full((M, J, 3), -3.0, device=dev)
    rot = torch.full((M, J, 9), -3.0, device=dev)
    _capi.call(name, P(g), P(b), is_rotmat, P(be), smpl.num_betas, P(smpl._j_template), P(smpl._j_shapedirs),
               _capi.iptr(smpl._parents_i32 if parents is None else parents), _capi.iptr(smpl._depth_i32 if depth is None else depth), J,
               P(xt), kp, mp, P(a), P(jp), P(rot), M, _capi.stream())
    return xt, a, jp, rot


@pytest.mark.parametrize("M", [1, 15, 16, 17, 130, 6528])
def test_pose_prep_second_generation_gives_the_first_ones_bits(M, dev, smpl_gpu):
    """the round-5 pose-prep kernel (depth by a shuffle reduction, a lane's joint-regressor coefficients requested
    all at once) against the first generation kept in the dev library: every output bit for bit -- both input routes, mesh counts
    that leave ragged workgroups, and the rows of the operand it must not touch (columns of meshes that do not exist) left alone."""
    betas, aa, _ = _pose(M, 40 + M, scale=0.9)
    R = O.batch_rodrigues(aa.view(-1, 3)).view(M, 24, 9)
    for is_rotmat, g, b in ((1, R[:, 0].contiguous(), R[:, 1:].reshape(M, -1).contiguous()),
                            (0, aa[:, 0].contiguous(), aa[:, 1:].reshape(M, -1).contiguous())):
        args = (smpl_gpu, g.to(dev), b.to(dev), is_rotmat, betas.to(dev), M, dev)
        new = _pose_prep("hps_smpl_pose_prep", *args)
        with _capi.dev_library():
            old = _pose_prep("hps_dev_smpl_pose_prep_v1", *args)
        for name, x, y in zip(("xt", "a", "j_posed", "rot"), new, old):
            assert torch.equal(x, y), (M, is_rotmat, name, float((x - y).abs().max()))
        assert bool((new[0][:, M:] == -3.0).all())                       # padding columns of the operand are not written


def test_pose_prep_with_other_kinematic_trees(dev, smpl_gpu):
    """Joint counts other than 24 and a chain-shaped tree (depth J - 1: more levels than the body's 8): same bits as the first
    generation."""
    for J, chain in ((7, False), (22, False), (32, True), (5, True)):
        g0 = torch.Generator().manual_seed(J)
        M = 37
        parents = torch.tensor([-1] + ([i for i in range(J - 1)] if chain else [max(0, (i - 1) // 2) for i in range(1, J)]), dtype=torch.int32)
        depth = torch.zeros(J, dtype=torch.int32)
        for j in range(1, J):
            depth[j] = depth[parents[j]] + 1
        jt = torch.randn(J, 3, generator=g0).to(dev)
        jsd = (0.1 * torch.randn(J, 3, 10, generator=g0)).to(dev)
        R = O.batch_rodrigues(torch.randn(M * J, 3, generator=g0)).view(M, J, 9)
        betas = torch.randn(M, 10, generator=g0).to(dev)

        class Shim:
            NUM_JOINTS, num_betas, _kp = J, 10, ((10 + 9 * (J - 1) + 15) // 16) * 16
            _j_template, _j_shapedirs = jt, jsd
        args = (Shim, R[:, 0].contiguous().to(dev), R[:, 1:].reshape(M, -1).contiguous().to(dev), 1, betas, M, dev, J, parents.to(dev), depth.to(dev))
        new = _pose_prep("hps_smpl_pose_prep", *args)
        with _capi.dev_library():
            old = _pose_prep("hps_dev_smpl_pose_prep_v1", *args)
        for name, x, y in zip(("xt", "a", "j_posed", "rot"), new, old):
            assert torch.equal(x, y), (J, chain, name)


@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 130, 6528])
def test_joints_second_generation(M, dev, smpl_gpu):
    """the round-5 joint kernel (the first generation's shape, the row sum pinned as ONE explicit fmaf chain in
    row order) on the reference's 90 joints and on a synthetic regressor with rows of 0, 1, 12, 13 and 40 entries and more than
    128 output rows, with and without translation:
      * the kinematic joints and the vertex picks (one entry of weight 1) bit for bit against the first generation;
      * regressed joints within 2 units in the last place of the first generation (whose compiled sum rounded two of every four
        products separately) and <= 2e-6 of a float64 evaluation;
      * a mesh's joints do not depend on the batch it is in: every mesh alone gives the same bits."""
    P = _capi.ptr
    g = torch.Generator().manual_seed(900 + M)
    V, J = smpl_gpu.num_verts, 24
    verts = torch.randn(M, V, 3, generator=g).to(dev)
    jp = torch.randn(M, J, 3, generator=g).to(dev)
    transl = torch.randn(M, 3, generator=g).to(dev)
    lens = [0, 1, 12, 13, 40, 5, 7] + [int(x) for x in torch.randint(0, 15, (150,), generator=g)]
    ptr = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    col = torch.randint(0, V, (int(ptr[-1]),), generator=g, dtype=torch.int32)
    val = torch.randn(int(ptr[-1]), generator=g)
    cases = [(smpl_gpu._csr_ptr, smpl_gpu._csr_col, smpl_gpu._csr_val, smpl_gpu._n_joint_rows),
             (ptr.to(dev), col.to(dev), val.to(dev), len(lens))]

    def run(name, cp, cc, cv, n_rows, tr, vs, js, dev_lib):
        m = vs.shape[0]
        o = torch.full((m, J + n_rows, 3), -5.0, device=dev)
        if dev_lib:
            with _capi.dev_library():
                _capi.call(name, P(vs), P(js), _capi.iptr(cp), _capi.iptr(cc), P(cv), n_rows, J, P(tr) if tr is not None else None, P(o), m, V, _capi.stream())
        else:
            _capi.call(name, P(vs), P(js), _capi.iptr(cp), _capi.iptr(cc), P(cv), n_rows, J, P(tr) if tr is not None else None, P(o), m, V, _capi.stream())
        return o

    for cp, cc, cv, n_rows in cases:
        dense = torch.zeros(n_rows, V, dtype=torch.float64)
        pc, cl, vl = cp.cpu(), cc.cpu().long(), cv.cpu().double()
        for r in range(n_rows):
            for e in range(int(pc[r]), int(pc[r + 1])):
                dense[r, cl[e]] += vl[e]
        ref = torch.einsum("rv,mvc->mrc", dense, verts.cpu().double())
        row_len = (pc[1:] - pc[:-1])
        for tr in (None, transl):
            new = run("hps_smpl_joints", cp, cc, cv, n_rows, tr, verts, jp, False)
            old = run("hps_dev_smpl_joints_v1", cp, cc, cv, n_rows, tr, verts, jp, True)
            assert torch.equal(new[:, :J], old[:, :J])                                        # kinematic joints
            single = (row_len <= 1).nonzero().flatten() + J                                   # picks / empty rows: nothing to round
            assert torch.equal(new[:, single], old[:, single])
            scale = max(1.0, float(ref.abs().max()))
            assert float((new - old).abs().max()) <= 4e-7 * scale
            assert float((new[:, J:].cpu().double() - ref).abs().max()) <= 2e-6 * scale
            # batch invariance: mesh i alone (another position in its workgroup's groups of four / pairs)
            for i in sorted({0, M // 2, M - 1}):
                alone = run("hps_smpl_joints", cp, cc, cv, n_rows, tr[i:i + 1].contiguous() if tr is not None else None,
                            verts[i:i + 1].contiguous(), jp[i:i + 1].contiguous(), False)
                assert torch.equal(alone[0], new[i]), (M, i)


@pytest.mark.parametrize("M", [1, 52, 63, 64, 65, 130, 700, 6528])
def test_mesh_kernel_side_output_feeds_the_joint_regression(M, dev, smpl_gpu):
    """Round 5: the fused mesh kernel also writes the 198 vertices the joint regressors read to a compact (M, 198, 3) array
    (hps_smpl_mesh_fused_picks) and hps_smpl_joints runs on that array through the entries' slots.  Same vertices (the compact array
    is exactly verts[:, picked vertices]), same joints bit for bit as the gather from the whole mesh, vertices untouched -- with and
    without translation, for mesh counts that leave ragged tiles and both K-loop forms (<= 2 tiles: four stages)."""
    betas, aa, transl = _pose(M, 500 + M, scale=0.7)
    R = O.batch_rodrigues(aa.view(-1, 3)).view(M, 24, 3, 3)
    for tr in (None, transl.to(dev)):
        args = dict(betas=betas.to(dev), body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), pose2rot=False, transl=tr)
        assert smpl_gpu.picked_joints
        with_picks = smpl_gpu(**args)
        try:
            smpl_gpu.picked_joints = False
            gathered = smpl_gpu(**args)
        finally:
            smpl_gpu.picked_joints = True
        assert torch.equal(with_picks.vertices, gathered.vertices)
        assert torch.equal(with_picks.joints, gathered.joints), float((with_picks.joints - gathered.joints).abs().max())
    # the side output itself, through the C ABI
    P = _capi.ptr
    mp = _capi.query_workspace(_capi.WS_SMPL_MP, M)
    g = torch.Generator().manual_seed(M)
    xt = (torch.randn(smpl_gpu._kp, mp, generator=g) * 0.1).to(dev)
    a = torch.randn(M, 24, 12, generator=g).to(dev)
    verts = torch.zeros(M, smpl_gpu.num_verts, 3, device=dev)
    picked = torch.full((M, smpl_gpu._n_picked, 3), -7.0, device=dev)
    _capi.call("hps_smpl_mesh_fused_picks", P(xt), P(smpl_gpu._bmat_p), P(smpl_gpu._v_template_flat), P(a), _capi.iptr(smpl_gpu._w_idx),
               P(smpl_gpu._w_val), 4, 24, None, P(verts), M, smpl_gpu.num_verts, smpl_gpu._k_used, mp, smpl_gpu._np_fused,
               _capi.iptr(smpl_gpu._pick_slot), P(picked), smpl_gpu._n_picked, _capi.stream())
    uniq = (smpl_gpu._pick_slot >= 0).nonzero().flatten()
    assert uniq.numel() == smpl_gpu._n_picked == 198
    assert torch.equal(picked, verts[:, uniq])
    with pytest.raises(_capi.HpsError):          # configurations without the side output say so
        _capi.call("hps_smpl_mesh_fused_picks", P(xt), P(smpl_gpu._bmat_p), P(smpl_gpu._v_template_flat), P(a), _capi.iptr(smpl_gpu._w_idx),
                   P(smpl_gpu._w_val), 8, 24, None, P(verts), M, smpl_gpu.num_verts, smpl_gpu._k_used, mp, smpl_gpu._np_fused,
                   _capi.iptr(smpl_gpu._pick_slot), P(picked), smpl_gpu._n_picked, _capi.stream())


def test_hip_smpl_matches_the_reference_class_on_smplx(dev, smplx_golden):
    """A10 / A11 pinned on the device: SMPL.forward (pose prep, fused mesh kernel, joints) against models/smpl_official.py:27-41
    running on the installed smplx (tests/golden/make_smpl_golden.py).  Skips where the fixture does not exist."""
    from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
    fix = smplx_golden
    for tag, gender, model in smplx_golden_models(fix):
        key = "%s_%s_" % (tag, gender)
        smpl = SMPL(model, batch_size=1, gender=gender, num_betas=10).to(dev)
        t = lambda name: torch.from_numpy(fix[key + name]).to(dev)
        betas, aa, transl, R = t("betas"), t("aa"), t("transl"), t("rotmats")
        M = betas.shape[0]
        sel = torch.from_numpy(fix[key + "vertex_ids"]).to(dev) if key + "vertex_ids" in fix else None
        outs = {
            "rotmat": smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, pose2rot=False),
            "aa": smpl(body_pose=aa[:, 1:].reshape(M, 69), global_orient=aa[:, 0], betas=betas),
            "tpose": smpl(betas=betas[:1]),
            "transl": smpl(body_pose=R[:, 1:], global_orient=R[:, :1], betas=betas, transl=transl, pose2rot=False),
        }
        for name, o in outs.items():
            v = o.vertices if sel is None else o.vertices[:, sel]
            assert maxerr(v, t(name + "_verts")) <= TOL, (key, name)
            assert maxerr(o.joints, t(name + "_joints")) <= TOL, (key, name)


@pytest.mark.parametrize("B,N", [(1, 1), (1, 50), (3, 7), (2, 100), (17, 33), (64, 100)])
def test_shared_shape_form_of_the_mesh_kernel(B, N, dev, smpl_gpu, smpl_assets):
    """hps_smpl_mesh_fused_shared_shape (K = 207: the shape blend once per image, smplx's v_posed = v_shaped + pose_offsets) on infer()'s
    mesh layout [mode | T-pose | samples]: vertices and joints within the stated 2e-5 m of the oracle, within rounding of the K = 217 form
    (one MFMA chain over template, shape and pose terms), and the group table covers tiles with one, two and many images."""
    p = smpl_assets[2]
    g = torch.Generator().manual_seed(1000 * B + N)
    loc = torch.randn(B, 10, generator=g)
    rows = list(range(B)) + list(range(B)) + [b for b in range(B) for _ in range(N)]
    M = len(rows)
    aa = torch.randn(M, 24, 3, generator=g) * 0.5
    aa[B:2 * B] = 0.0                                            # the T-pose meshes
    R = O.batch_rodrigues(aa.view(-1, 3)).view(M, 24, 3, 3)
    betas = loc[torch.tensor(rows)]
    mesh_row, group_rows = smpl_gpu.shared_shape_tables(rows)
    gr = group_rows.view(-1, 3).cpu()
    assert mesh_row.shape[0] % 64 == 0 and mesh_row[:M].cpu().tolist() == rows and gr.shape[0] == mesh_row.shape[0] // 32
    for gi in range(gr.shape[0]):                                # the table says what the rows are
        r = mesh_row[32 * gi:32 * gi + 32].cpu()
        a, b_, split = gr[gi].tolist()
        if split >= 0:
            assert (r[:split] == a).all() and (r[split:] == b_).all()
        else:
            assert int((r[1:] != r[:-1]).sum()) > 1                 # more than one change of row inside the group
    args = dict(betas=betas.to(dev), body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), pose2rot=False)
    shared = smpl_gpu(_shared_shapes=(loc.to(dev), mesh_row, group_rows), **args)
    plain = smpl_gpu(**args)                                     # K = 217: template + shape + pose in one chain
    assert maxerr(shared.vertices, plain.vertices) <= 4e-6 and maxerr(shared.joints, plain.joints) <= 4e-6
    if M <= 700:
        ref = O.smpl_forward(p, betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False)
        assert maxerr(shared.vertices, ref["vertices"]) <= TOL and maxerr(shared.joints, ref["joints"]) <= TOL
    else:                                                        # full size: the oracle on a slice of every region
        pick = torch.cat([torch.arange(0, 3), torch.arange(B, B + 3), torch.arange(2 * B, 2 * B + 70), torch.arange(M - 40, M)])
        ref = O.smpl_forward(p, betas=betas[pick], body_pose=R[pick, 1:], global_orient=R[pick, :1], pose2rot=False)
        assert maxerr(shared.vertices[pick.to(dev)], ref["vertices"]) <= TOL and maxerr(shared.joints[pick.to(dev)], ref["joints"]) <= TOL
    # the T-pose meshes are the shaped template itself (identity rotations: A = identity up to the rest-pose subtraction)
    smpl_gpu.shared_shape = False
    try:
        off = smpl_gpu(_shared_shapes=(loc.to(dev), mesh_row, group_rows), **args)      # the switch: the K = 217 form again
    finally:
        smpl_gpu.shared_shape = True
    assert torch.equal(off.vertices, plain.vertices)


def _split_reference(x):
    """x = x1 + x2 + x3 in bf16 pieces, round-to-nearest-even each (torch's own conversion)."""
    x1 = x.bfloat16()
    r1 = x - x1.float()
    x2 = r1.bfloat16()
    x3 = (r1 - x2.float()).bfloat16()
    return x1, x2, x3


@pytest.mark.parametrize("rows,cols,W", [(207, 128, 64), (16, 64, 64), (5, 192, 192), (207, 384, 192), (33, 320, 64), (207, 256, 128)])
def test_split_bf16x3_operand_layout_and_exactness(rows, cols, W, dev):
    """hps_smpl_split_bf16x3: the three pieces are torch's round-to-nearest-even bf16 conversions of x, x - x1, x - x1 - x2, they lie in
    [tile][16-row chunk][piece][k half][column][8] order, rows behind `rows` are zero, and x1 + x2 + x3 == x EXACTLY."""
    g = torch.Generator().manual_seed(rows * 1000 + cols)
    ld = cols + 64
    src = (torch.randn(rows, ld, generator=g) * torch.logspace(-6, 1, ld)[None]).to(dev)
    lib = _capi.load()
    nbytes = lib.hps_smpl_split_bf16x3_bytes(rows, cols)
    nchunks = -(-rows // 16)
    assert nbytes == nchunks * 16 * cols * 6
    dst = torch.full((nbytes,), 0x55, dtype=torch.uint8, device=dev)
    _capi.call("hps_smpl_split_bf16x3", _capi.ptr(src), rows, ld, cols, W, _capi._P(dst.data_ptr()), _capi.stream())
    got = dst.view(torch.bfloat16).view(cols // W, nchunks, 3, 2, W, 8)
    x = torch.zeros(nchunks * 16, cols, device=dev)
    x[:rows] = src[:, :cols]
    pieces = _split_reference(x)
    want = torch.stack([p.view(nchunks, 2, 8, cols // W, W).permute(3, 0, 1, 4, 2) for p in pieces], dim=2)   # (tile, chunk, piece, kl, w, j)
    assert torch.equal(got.view(torch.int16), want.contiguous().view(torch.int16))
    total = got[:, :, 0].double() + got[:, :, 1].double() + got[:, :, 2].double()
    back = total.permute(1, 2, 4, 0, 3).reshape(nchunks * 16, cols)
    assert torch.equal(back, x.double())
    with pytest.raises(_capi.HpsError):
        _capi.call("hps_smpl_split_bf16x3", _capi.ptr(src), rows, ld, cols, 100, _capi._P(dst.data_ptr()), _capi.stream())


@pytest.mark.parametrize("B,N", [(1, 1), (1, 50), (3, 7), (2, 100), (17, 33), (64, 100)])
def test_split_bf16_form_of_the_mesh_kernel(B, N, dev, smpl_gpu, smpl_assets):
    """SMPL.mesh_arith = "bf16x3" (hps_smpl_mesh_fused_shared_shape_bf16x3: the pose blend GEMM as six exact bf16 piece products per
    product, fp32 accumulation) on infer()'s mesh layout: within rounding of the fp32-MFMA form (4e-6 m), within the stated 2e-5 m of the
    oracle, and NO FURTHER from the float64 twin than the fp32-MFMA form is (fp32 accuracy, not fp32 bits); the joints follow from the
    side output.  Without shared shapes the same kernel runs over all 217 rows (shape blend inside the GEMM); a call with a translation
    keeps the fp32 kernel whatever the switch says."""
    model, extra, p = smpl_assets
    g = torch.Generator().manual_seed(7000 * B + N)
    loc = torch.randn(B, 10, generator=g)
    rows = list(range(B)) + list(range(B)) + [b for b in range(B) for _ in range(N)]
    M = len(rows)
    aa = torch.randn(M, 24, 3, generator=g) * 0.5
    aa[B:2 * B] = 0.0
    R = O.batch_rodrigues(aa.view(-1, 3)).view(M, 24, 3, 3)
    betas = loc[torch.tensor(rows)]
    mesh_row, group_rows = smpl_gpu.shared_shape_tables(rows)
    args = dict(betas=betas.to(dev), body_pose=R[:, 1:].to(dev), global_orient=R[:, :1].to(dev), pose2rot=False)
    f32 = smpl_gpu(_shared_shapes=(loc.to(dev), mesh_row, group_rows), **args)
    plain32 = smpl_gpu(**args)
    tr = torch.randn(M, 3, generator=g).to(dev)
    moved32 = smpl_gpu(transl=tr, **args)
    smpl_gpu.mesh_arith = "bf16x3"
    try:
        sp = smpl_gpu(_shared_shapes=(loc.to(dev), mesh_row, group_rows), **args)
        sp2 = smpl_gpu(_shared_shapes=(loc.to(dev), mesh_row, group_rows), **args)
        plain = smpl_gpu(**args)                                  # no shared shapes: the same kernel over all 217 rows
        moved = smpl_gpu(transl=tr, **args)                       # a translation: the fp32 kernel
    finally:
        smpl_gpu.mesh_arith = "f32"
    assert torch.equal(sp.vertices, sp2.vertices) and torch.equal(sp.joints, sp2.joints)           # deterministic
    assert maxerr(sp.vertices, f32.vertices) <= 4e-6 and maxerr(sp.joints, f32.joints) <= 4e-6
    assert maxerr(plain.vertices, plain32.vertices) <= 4e-6 and maxerr(plain.joints, plain32.joints) <= 4e-6
    assert not torch.equal(plain.vertices, plain32.vertices) and not torch.equal(sp.vertices, f32.vertices)     # (the other kernel did run)
    assert torch.equal(moved.vertices, moved32.vertices) and torch.equal(moved.joints, moved32.joints)
    pick = torch.arange(M) if M <= 300 else torch.cat([torch.arange(0, 3), torch.arange(B, B + 3), torch.arange(2 * B, 2 * B + 70),
                                                         torch.arange(M - 40, M)])
    ref = O.smpl_forward(p, betas=betas[pick], body_pose=R[pick, 1:], global_orient=R[pick, :1], pose2rot=False)
    assert maxerr(sp.vertices[pick.to(dev)], ref["vertices"]) <= TOL and maxerr(sp.joints[pick.to(dev)], ref["joints"]) <= TOL
    v64, j64 = smpl_forward64(model, extra, configs.SMPLX_EXTRA_VERTEX_IDS, betas[pick].double().numpy(), R[pick].double().numpy())
    for got, base in ((sp, f32), (plain, plain32)):
        e_sp = np.abs(got.vertices[pick.to(dev)].cpu().numpy() - v64)
        e_f32 = np.abs(base.vertices[pick.to(dev)].cpu().numpy() - v64)
        assert e_sp.max() <= max(1.25 * e_f32.max(), 1e-6) and e_sp.mean() <= 1.1 * e_f32.mean() + 1e-9, (e_sp.max(), e_f32.max(), e_sp.mean(), e_f32.mean())
    with pytest.raises(ValueError):
        smpl_gpu.mesh_arith = "bf16"
        try:
            smpl_gpu(_shared_shapes=(loc.to(dev), mesh_row, group_rows), **args)
        finally:
            smpl_gpu.mesh_arith = "f32"

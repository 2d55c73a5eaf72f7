"""CPU: host-side logic of the boundary -- kinematic levels, state-dict layout, derived SMPL constants,
synthetic-model determinism, sharding arithmetic."""
import hashlib

import numpy as np
import pytest
import torch

from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data, sharding
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import immediate_parents_to_all_parents
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
from oracle import ref_cpu as O


def test_ancestor_lists_match_reference_probe():
    anc = immediate_parents_to_all_parents(configs.SMPL_PARENTS)
    assert len(anc) == 23
    # values printed by the reference's immediate_parents_to_all_parents (models/poseMF_shapeGaussian_net.py:14-21)
    assert anc[0] == [] and anc[3] == [0] and anc[6] == [3, 0] and anc[14] == [11, 8, 5, 2]
    assert anc[22] == [20, 18, 16, 13, 8, 5, 2] and anc[21] == [19, 17, 15, 12, 8, 5, 2]
    assert [anc[j] for j in range(23)] == O.all_ancestors(configs.SMPL_PARENTS)


def test_levels_and_state_dict_layout(net_cpu):
    net, sd = net_cpu
    assert net.levels == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11, 12, 13], [14, 15, 16], [17, 18], [19, 20], [21, 22]]
    assert len(sd) == 224 and sum(p.numel() for p in net.parameters()) == 12616684
    assert sd["fc_pose.22.0.weight"].shape == (128, 256 + 21 * 7) and sd["fc_pose.0.2.weight"].shape == (9, 128)
    assert sd["init_glob"].tolist() == [[1.0, 0.0, 0.0, 1.0, 0.0, 0.0]]
    assert sd["image_encoder.conv1.weight"].shape == (64, 18, 7, 7)


def test_synthetic_smpl_is_deterministic_and_smpl_shaped():
    m = smpl_data.synthetic_smpl_model(0)
    assert m["v_template"].shape == (6890, 3) and m["shapedirs"].shape == (6890, 3, 10)
    assert m["posedirs"].shape == (6890, 3, 207) and m["J_regressor"].shape == (24, 6890) and m["weights"].shape == (6890, 24)
    assert ((m["weights"] != 0).sum(1) <= 4).all() and np.allclose(m["weights"].sum(1), 1) and np.allclose(m["J_regressor"].sum(1), 1)
    h = hashlib.sha256(m["v_template"].tobytes() + m["weights"].tobytes() + m["posedirs"].tobytes()).hexdigest()
    assert h == hashlib.sha256(smpl_data.synthetic_smpl_model(0)["v_template"].tobytes() + m["weights"].tobytes() + m["posedirs"].tobytes()).hexdigest()
    assert smpl_data.parents_from_kintree(m["kintree_table"]).tolist() == configs.SMPL_PARENTS


def test_extra_regressors_packaged_copy():
    extra, coco, h36m = smpl_data.load_extra_joint_regressors(None)
    assert extra.shape == (9, 6890) and coco.shape == (19, 6890) and h36m.shape == (17, 6890)
    assert (extra != 0).sum() == 62 and (coco != 0).sum() == 86 and (h36m != 0).sum() == 107          # SURVEY.md probe


def test_smpl_derived_constants(smpl_assets):
    model, extra, p = smpl_assets
    smpl = SMPL(model)
    V = 6890
    # compressed skin weights reproduce the dense matrix exactly
    dense = torch.zeros(V, 24)
    dense.scatter_add_(1, smpl._w_idx.long(), smpl._w_val)
    assert torch.equal(dense, smpl.lbs_weights)
    # blend matrix rows: shapedirs then posedirs, zero padded
    assert smpl._bmat.shape == (224, 20736) and torch.equal(smpl._bmat[10:217, :3 * V], smpl.posedirs)
    assert float(smpl._bmat[217:].abs().max()) == 0 and float(smpl._bmat[:, 3 * V:].abs().max()) == 0
    # folded joint regression equals regressing the shaped template
    betas = torch.randn(3, 10)
    J = smpl._j_template + torch.einsum("jcl,bl->bjc", smpl._j_shapedirs, betas)
    ref = O.smpl_forward(p, betas=betas, return_intermediates=True, body_pose=torch.zeros(3, 69), global_orient=torch.zeros(3, 3))
    assert float((J - ref["J"]).abs().max()) <= 1e-6
    # CSR joint rows: 21 picks + 9 + 19 + 17
    assert smpl._n_joint_rows == 66 and int(smpl._csr_ptr[-1]) == 21 + 62 + 86 + 107
    assert smpl._depth_i32.tolist() == [0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7, 7, 8, 8]
    assert smpl.parents.tolist() == configs.SMPL_PARENTS and len(smpl.state_dict()) == 12


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 4, 8):
            r = [sharding.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(512, 3, 8) == (192, 256)                      # BASELINE configs[2]: 64 images per GPU


def _emulated_svd(F, flavor=None):
    import ctypes
    from hierarchicalprobabilistic3dhuman_amd import _capi
    F = F.contiguous().float()
    out = torch.empty(F.shape[0], 21)
    rc = _capi.load().hps_host_svd3_emulated(ctypes.c_void_p(F.data_ptr()), ctypes.c_void_p(out.data_ptr()), F.shape[0],
                                             _capi.svd_flavor() if flavor is None else flavor)
    return out[:, :9].reshape(-1, 3, 3), out[:, 9:12], out[:, 12:].reshape(-1, 3, 3), rc


def _svd_families(n):
    """Matrix families the bit-level agreement is asserted on: the head's regime (I + noise), generic, ill-conditioned,
    rank-deficient, exact ties (orthogonal matrices: sigma = 1, 1, 1), exact and signed zeros, structured, tiny / huge
    (sgesdd's rescaling path)."""
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    eye = torch.eye(3)[None]
    fam = [("I + 0.05 N", eye + 0.05 * rn(n, 3, 3)), ("I + 0.5 N", eye + 0.5 * rn(n, 3, 3)), ("I + 2 N", eye + 2.0 * rn(n, 3, 3)),
           ("30 N", 30.0 * rn(n, 3, 3)), ("100 I + 20 N", 100.0 * eye + 20.0 * rn(n, 3, 3)), ("I + 1e-3 N", eye + 1e-3 * rn(n, 3, 3)),
           ("diag(500,400,300) + N", torch.diag(torch.tensor([500.0, 400.0, 300.0]))[None] + rn(n, 3, 3)),
           ("columns scaled 1e3, 1, 1e-3", rn(n, 3, 3) * torch.tensor([1e3, 1.0, 1e-3]))]
    q = torch.linalg.qr(rn(n // 4, 3, 3))[0]
    fam += [("orthogonal (exact ties)", q), ("orthogonal + 1e-6 N", q + 1e-6 * rn(n // 4, 3, 3)),
            ("rank 2", rn(n // 4, 3, 2) @ rn(n // 4, 2, 3)), ("rank 1", rn(n // 4, 3, 1) @ rn(n // 4, 1, 3)),
            ("integers", torch.randint(-3, 4, (n // 4, 3, 3), generator=g).float()),
            ("half zeros (signed)", rn(n // 4, 3, 3) * (torch.rand(n // 4, 3, 3, generator=g) > 0.5)),
            ("half zeros (+0)", rn(n // 4, 3, 3) * (torch.rand(n // 4, 3, 3, generator=g) > 0.5) + 0.0),
            ("diagonal", torch.diag_embed(rn(n // 4, 3))), ("upper triangular", torch.triu(rn(n // 4, 3, 3))),
            ("lower triangular", torch.tril(rn(n // 4, 3, 3))), ("symmetric", (lambda a: a + a.transpose(1, 2))(rn(n // 4, 3, 3))),
            ("1e-20 N", 1e-20 * rn(n // 4, 3, 3)), ("1e20 N", 1e20 * rn(n // 4, 3, 3)),
            ("1e-30 I + 1e-31 N", 1e-30 * eye + 1e-31 * rn(n // 4, 3, 3))]
    return [(name, f.contiguous().float()) for name, f in fam]


def test_gesdd_faithful_svd_agrees_with_lapack(golden):
    """the device SVD's algorithm (csrc/svd3_gesdd.h: LAPACK sgesdd followed step by step with MKL's roundings),
    compiled for the host, against torch.svd = MKL sgesdd, the reference's routine (models/poseMF_shapeGaussian_net.py:137).
    The singular-vector SIGNS are load-bearing (:126-130).  MKL picks its kernels by the host CPU (fused multiply-adds on Intel
    hosts, reference-BLAS rounding elsewhere); hps_host_svd_flavor calibrates which flavour of the header matches this host's,
    and with it U, S and V must be BIT-IDENTICAL to torch.svd on every matrix of 22 families (1.15 x 10^6 matrices).  The other
    flavour is the same algorithm with other last bits: no factor further than 1e-3 / 2e-6 away, and a differently signed
    singular-vector pair in about one matrix of 10^4 -- which is how far the reference itself is reproducible between an Intel
    and an AMD host."""
    from hierarchicalprobabilistic3dhuman_amd import _capi
    native = _capi.load().hps_host_svd_flavor()
    assert native in (0, 1), "neither rounding flavour reproduces this host's LAPACK sgesdd (another MKL build?)"
    assert _capi.svd_flavor() == native
    cases = [("golden net_F", golden["net_F"].reshape(-1, 3, 3).contiguous())] + _svd_families(100000)
    total = other_flips = well_total = 0
    for name, F in cases:
        U, S, V = torch.svd(F)
        u, s, v, rc = _emulated_svd(F, native)
        assert rc == 0, name
        assert torch.equal(s, S) and torch.equal(u, U) and torch.equal(v, V), name          # bit for bit (as values: -0 == +0)
        total += F.shape[0]
        # the other flavour
        u2, s2, v2, rc = _emulated_svd(F, 1 - native)
        assert rc == 0, name
        scale = S.abs().max(1, keepdim=True)[0].clamp_min(1e-30)
        assert float(((S - s2).abs() / scale).max()) <= 2e-6, name
        # singular vectors are comparable where the values are well separated and none is (numerically) zero
        well = (torch.minimum(S[:, 0] - S[:, 1], S[:, 1] - S[:, 2]) / scale[:, 0] > 1e-3) & (S[:, 2] / scale[:, 0] > 1e-4)
        flipped = (((U * u2).sum(1) < 0) | ((V * v2).sum(1) < 0)).any(1) & well
        other_flips += int(flipped.sum())
        well_total += int(well.sum())
        gap_ok = well & ~flipped
        if gap_ok.any():
            assert float(torch.maximum((U - u2)[gap_ok].abs().amax(), (V - v2)[gap_ok].abs().amax())) <= 1e-3, name
        rec = torch.matmul(u2 * s2[:, None, :], v2.transpose(1, 2))
        assert float(((rec - F).abs().amax((1, 2)) / F.abs().amax((1, 2)).clamp_min(1e-30)).max()) <= 1e-5, name
        if name == "golden net_F":
            assert int(flipped.sum()) == 0                       # the golden matrices come out alike in both flavours
    print("svd3_gesdd flavour %d vs torch.svd: %d of %d matrices bit-identical in U, S, V; the other flavour: %d of %d "
          "well-conditioned matrices with a differently signed vector pair (%.1e)"
          % (native, total, total, other_flips, well_total, other_flips / max(1, well_total)))
    assert other_flips <= 5e-4 * well_total


def test_gesdd_faithful_svd_edge_cases():
    nan = float("nan")
    F = torch.stack([torch.eye(3), torch.zeros(3, 3), torch.diag(torch.tensor([1.0, 2.0, 3.0])), torch.full((3, 3), nan),
                     torch.eye(3) * 1e-20, torch.eye(3) * 1e20])
    u, s, v, rc = _emulated_svd(F)
    assert rc != 0                                                     # the NaN matrix is reported (LAPACK: INFO != 0)
    assert torch.isnan(u[3]).all() and torch.isnan(s[3]).all()
    assert s[0].tolist() == [1.0, 1.0, 1.0] and s[1].tolist() == [0.0, 0.0, 0.0] and s[2].tolist() == [3.0, 2.0, 1.0]
    for i in (0, 1, 2, 4, 5):                                          # tiny / huge matrices go through sgesdd's rescaling
        rec = torch.matmul(u[i] * s[i][None], v[i].T)
        assert float((rec - F[i]).abs().max()) <= 1e-5 * max(1e-30, float(F[i].abs().max()))


def test_stem_winograd_filters_reproduce_the_convolution():
    """resnet._stem_winograd_filters (the 81 transformed filter positions of csrc/stem_wino.hip, in the kernel's packed layout)
    together with the B^T / A^T matrices the kernel hard-codes is the 7x7 / 2 / 3 convolution: evaluated here in fp64."""
    import numpy as np
    from hierarchicalprobabilistic3dhuman_amd.resnet import _stem_winograd_filters
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 18, 7, 7, generator=g, dtype=torch.float64)
    x = torch.randn(2, 18, 32, 64, generator=g, dtype=torch.float64)
    packed = _stem_winograd_filters(w)
    assert packed.dtype == torch.float32 and packed.numel() == 81 * 1152 + 256 and float(packed[-256:].abs().max()) == 0
    # unpack: [position][co half][k-block][c parity][co % 32][(c % 8) / 2] for c < 16, then [c - 16][co % 32]
    pk = packed[:81 * 1152].double().view(81, 2, 576)
    lo = pk[:, :, :512].reshape(81, 2, 2, 2, 32, 4)                # p, half, kb, parity, co32, e
    U = torch.zeros(81, 18, 64, dtype=torch.float64)
    for kb in range(2):
        for par in range(2):
            for e in range(4):
                U[:, 8 * kb + 2 * e + par, :] = lo[:, :, kb, par, :, e].reshape(81, 64)
    hi = pk[:, :, 512:].reshape(81, 2, 2, 32)                      # p, half, c - 16, co32
    for c in range(2):
        U[:, 16 + c, :] = hi[:, :, c, :].reshape(81, 64)
    U = U.numpy()
    BT = {5: np.array([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]], float),
          4: np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, -1, 0, 1]], float)}
    AT = {5: np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 1]], float), 4: np.array([[1, 1, 1, 0], [0, 1, -1, 1]], float)}
    B, C, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    xp = np.zeros((B, C, H + 8, W + 8))
    xp[:, :, 3:3 + H, 3:3 + W] = x.numpy()
    y = np.zeros((B, 64, Ho, Wo))
    pos = 0
    for ry in (0, 1):
        for rx in (0, 1):
            ny, nx = 5 - ry, 5 - rx
            ph = xp[:, :, ry::2, rx::2]
            iy = (2 * np.arange(Ho // 2))[:, None] + np.arange(ny)[None]
            ix = (2 * np.arange(Wo // 2))[:, None] + np.arange(nx)[None]
            d = ph[:, :, iy][:, :, :, :, ix]                       # (B, C, ty, ny, tx, nx)
            V = np.einsum("ia,jb,ncpaqb->ijnpqc", BT[ny], BT[nx], d)
            Uph = U[pos:pos + ny * nx].reshape(ny, nx, 18, 64)
            M = np.einsum("ijnpqc,ijco->ijnpqo", V, Uph)
            Y = np.einsum("ai,bj,ijnpqo->nopaqb", AT[ny], AT[nx], M)
            y += Y.reshape(B, 64, Ho, Wo)
            pos += ny * nx
    assert pos == 81
    want = torch.nn.functional.conv2d(x, w, stride=2, padding=3).numpy()
    # U went through fp32 once (2^-24 relative per entry)
    assert np.abs(y - want).max() <= 3e-6 * np.abs(want).max()


def test_smpl_pkl_loader_reads_the_licensed_file_format(tmp_path):
    """BASELINE configs[3] needs the licensed SMPL_{NEUTRAL,MALE,FEMALE}.pkl (absent here; the reference reads them at
    models/smpl_official.py:15-16 through smplx).  The guarded loader is exercised on files WRITTEN in that format from the
    synthetic model: scipy-sparse J_regressor, 300 shape directions of which the first num_betas are kept, an unsigned
    kintree_table whose root parent is 2^32 - 1, extra keys ignored; SMPL(model_dir, gender=...) resolves the reference's
    directory + gender convention and ends with the constants of the in-memory model, bit for bit."""
    import pickle
    import numpy as np
    import scipy.sparse
    from hierarchicalprobabilistic3dhuman_amd import smpl_data
    from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
    for gender, seed in (("neutral", 0), ("male", 1), ("female", 2)):
        m = smpl_data.synthetic_smpl_model(seed)
        shapedirs300 = np.concatenate([np.asarray(m["shapedirs"]), np.zeros((6890, 3, 290))], axis=2)
        kt = np.asarray(m["kintree_table"]).astype(np.int64).copy()
        kt[0, 0] = 2 ** 32 - 1
        raw = {"v_template": np.asarray(m["v_template"]), "shapedirs": shapedirs300, "posedirs": np.asarray(m["posedirs"]),
               "J_regressor": scipy.sparse.csc_matrix(np.asarray(m["J_regressor"])), "weights": np.asarray(m["weights"]),
               "kintree_table": kt.astype(np.uint32), "f": np.zeros((13776, 3), np.uint32), "bs_style": "lbs", "bs_type": "lrotmin"}
        with open(tmp_path / ("SMPL_%s.pkl" % gender.upper()), "wb") as f:
            pickle.dump(raw, f, protocol=2)
    with pytest.raises(FileNotFoundError):
        smpl_data.resolve_smpl_model(str(tmp_path / "missing"), gender="neutral")
    for gender, seed in (("neutral", 0), ("male", 1), ("female", 2)):
        loaded = smpl_data.resolve_smpl_model(str(tmp_path), gender=gender, num_betas=10)
        want = smpl_data.synthetic_smpl_model(seed)
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):
            assert np.array_equal(np.asarray(loaded[k]), np.asarray(want[k])), (gender, k)
        assert smpl_data.parents_from_kintree(loaded["kintree_table"]).tolist() == \
            [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
        from_file, from_dict = SMPL(str(tmp_path), batch_size=1, gender=gender, num_betas=10), SMPL(want, gender=gender)
        assert from_file.parents.tolist() == from_dict.parents.tolist()
        a, b = from_file.state_dict(), from_dict.state_dict()
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_kernel_selection_modes_survive_invalidation():
    """latency mode / the Winograd switch are properties of the MODEL -- .to(), load_state_dict() or invalidate()
    rebuild the prepared layers and must re-apply them; leaving latency mode restores the earlier Winograd choice."""
    from hierarchicalprobabilistic3dhuman_amd.resnet import resnet18
    torch.manual_seed(0)
    enc = resnet18(in_channels=18).eval()
    convs = lambda: [enc._prepared["stem"]] + [c for blk in enc._prepared["blocks"] for c in blk if c is not None]
    enc.prepare()
    assert all(c.use_winograd and not c.latency for c in convs())
    enc.set_latency_mode(True)
    assert all(c.latency and not c.use_winograd for c in convs())
    enc.invalidate()                                   # what .to() / load_state_dict trigger
    assert enc._prepared is None
    enc.prepare()
    assert all(c.latency and not c.use_winograd for c in convs()), "latency mode lost by the rebuild"
    enc.load_state_dict(enc.state_dict())
    enc.prepare()
    assert all(c.latency for c in convs())
    enc.set_latency_mode(False)
    assert all(c.use_winograd and not c.latency for c in convs())
    enc.set_winograd(False)
    enc.set_latency_mode(True)
    enc.set_latency_mode(False)
    assert all(not c.use_winograd and not c.latency for c in convs()), "set_latency_mode(False) must not force Winograd back on"


def test_bf16x3_split_is_exact_and_the_dropped_products_are_below_one_fp32_rounding():
    """The arithmetic behind SMPL.mesh_arith = "bf16x3" (csrc/mesh_split.hip), stated on the host with torch's own conversions: an fp32
    number is EXACTLY the sum of three round-to-nearest bf16 pieces; the six piece products the kernel issues reproduce the fp32 product to
    within 2^-23 |a b| (the three dropped ones are a2 b3, a3 b2, a3 b3), i.e. to one fp32 rounding; the product of two pieces is exact in fp32.  (Range: |x| below bf16's largest finite
    value, 3.39e38 -- beyond it the first piece rounds to infinity -- and at least 2^-110, so that the third piece, 2^-16 of the first, is still
    a normal number; the operands here are pose features and blend shapes: |x| < 10, and a pose-blend entry below 2^-110 m moves nothing.)"""
    g = torch.Generator().manual_seed(0)
    mags = torch.logspace(-30, 30, 4096, base=2.0)
    x = torch.cat([torch.randn(4096, generator=g) * mags, torch.tensor([1.0, -1.0, 3.0, 1 / 3, 2.0 ** -100, 1.9999999, 16777215.0, 3.3e38,
                                                                       -2.0 ** -126, 0.0])])
    y = torch.cat([torch.randn(4096, generator=g), torch.tensor([0.1, 7.0, -1 / 3, 3.0, 2.0 ** 90, 1.0000001, 1e-3, 1.2345678 * 2.0 ** -100, 5.0, 1.0])])

    def split(v):
        v1 = v.bfloat16()
        r1 = v - v1.float()
        v2 = r1.bfloat16()
        v3 = (r1 - v2.float()).bfloat16()
        return v1.float(), v2.float(), v3.float()

    xs, ys = split(x), split(y)
    for v, vs in ((x, xs), (y, ys)):
        assert torch.equal(vs[0].double() + vs[1].double() + vs[2].double(), v.double())             # exact, piece by piece
        assert torch.equal((vs[0] + vs[1]) + vs[2], v)
    kept = [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]
    total = torch.zeros_like(x, dtype=torch.float64)
    for i, j in kept:
        p32 = xs[i] * ys[j]                                                                          # fp32 product of two bf16 pieces
        assert torch.equal(p32.double(), xs[i].double() * ys[j].double())                            # ... is exact (8 x 8 significand bits)
        total += p32.double()
    exact = x.double() * y.double()
    finite = torch.isfinite(exact) & (exact.abs() > 2.0 ** -100) & (exact.abs() < 2.0 ** 100)
    rel = ((total - exact).abs() / exact.abs())[finite]
    assert float(rel.max()) <= 2.0 ** -23, float(rel.max())

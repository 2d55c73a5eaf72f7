"""CPU: analytic known-answer tests and the float64 twin for the SMPL restatement (no importable oracle
exists for smplx: SURVEY.md section 4.2 / section 8(c); parity for this part is unpinned and anchored here)."""
import numpy as np
import torch

from oracle import ref_cpu as O
from oracle.smpl_np64 import smpl_forward64, rodrigues64
from hierarchicalprobabilistic3dhuman_amd import configs
from conftest import maxerr, smplx_golden_models


def _rand_pose(M, seed, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    betas = torch.randn(M, 10, generator=g)
    aa = torch.randn(M, 24, 3, generator=g) * scale
    return betas, aa


def test_zero_pose_zero_betas_gives_template(smpl_assets):
    _, _, p = smpl_assets
    out = O.smpl_forward(p)
    assert maxerr(out["vertices"][0], p.v_template) <= 3e-7   # skin weights sum to 1 within an ulp
    assert out["joints"].shape == (1, 90, 3)


def test_identity_rotmats_equal_zero_axis_angle(smpl_assets):
    _, _, p = smpl_assets
    betas, _ = _rand_pose(3, 0)
    eye = torch.eye(3).expand(3, 24, 3, 3)
    a = O.smpl_forward(p, betas=betas, body_pose=eye[:, 1:], global_orient=eye[:, :1], pose2rot=False)
    b = O.smpl_forward(p, betas=betas, body_pose=torch.zeros(3, 69), global_orient=torch.zeros(3, 3))
    assert maxerr(a["vertices"], b["vertices"]) == 0.0


def test_global_rotation_is_rigid_about_root(smpl_assets):
    _, _, p = smpl_assets
    betas, aa = _rand_pose(2, 1)
    R0 = O.batch_rodrigues(aa[:, 0])
    eye = torch.eye(3).expand(2, 23, 3, 3)
    rest = O.smpl_forward(p, betas=betas, return_intermediates=True, body_pose=torch.zeros(2, 69), global_orient=torch.zeros(2, 3))
    out = O.smpl_forward(p, betas=betas, body_pose=eye, global_orient=R0[:, None], pose2rot=False)
    # pose blend shapes ignore the root, so v = R (v_rest - J0) + J0
    J0 = rest["J"][:, :1]
    want = torch.einsum("bij,bvj->bvi", R0, rest["vertices"] - J0) + J0
    assert maxerr(out["vertices"], want) <= 2e-6


def test_translation_equivariance_and_joint_layout(smpl_assets):
    _, _, p = smpl_assets
    betas, aa = _rand_pose(2, 2)
    t = torch.tensor([[0.3, -0.2, 1.0], [0.0, 0.5, -0.7]])
    a = O.smpl_forward(p, betas=betas, body_pose=aa[:, 1:].reshape(2, 69), global_orient=aa[:, 0], return_intermediates=True)
    b = O.smpl_forward(p, betas=betas, body_pose=aa[:, 1:].reshape(2, 69), global_orient=aa[:, 0], transl=t)
    assert maxerr(b["vertices"], a["vertices"] + t[:, None]) <= 1e-6
    # joint layout: 24 FK joints, 21 vertex picks, 9 + 19 regressed, h36m block at 73..89 (label_conversions.py:18)
    assert maxerr(a["joints"][:, :24], a["J_posed"]) == 0.0
    assert maxerr(a["joints"][:, 24:45], a["vertices"][:, configs.SMPLX_EXTRA_VERTEX_IDS]) == 0.0
    h36m = torch.einsum("jv,bvk->bjk", p.J_regressor_h36m, a["vertices"])
    assert maxerr(a["joints"][:, 73:90], h36m) <= 1e-6


def test_fp32_restatement_agrees_with_float64_twin(smpl_assets):
    model, extra, p = smpl_assets
    betas, aa = _rand_pose(4, 3, scale=0.8)
    out = O.smpl_forward(p, betas=betas, body_pose=aa[:, 1:].reshape(4, 69), global_orient=aa[:, 0])
    R64 = rodrigues64(aa.reshape(-1, 3).double().numpy()).reshape(4, 24, 3, 3)
    v64, j64 = smpl_forward64(model, extra, configs.SMPLX_EXTRA_VERTEX_IDS, betas.double().numpy(), R64)
    assert np.abs(out["vertices"].numpy() - v64).max() <= 2e-6
    assert np.abs(out["joints"].numpy() - j64).max() <= 2e-6


def test_vertex_uncertainty_definition():
    v = torch.randn(7, 11, 3)
    want = torch.stack([(v[:, i] - v[:, i].mean(0)).norm(dim=-1).mean() for i in range(11)])
    assert maxerr(O.vertex_uncertainty(v), want) <= 1e-6


def test_batched_infer_equals_looping_single_images(smpl_assets, net_cpu, golden):
    """The batched oracle path must equal looping the reference's B=1 calls (same RNG order)."""
    _, _, p = smpl_assets
    feats = golden["net_feats"]
    torch.manual_seed(4)
    both = O.infer(net_cpu[1], p, configs.SMPL_PARENTS, None, 3, feats=feats)
    torch.manual_seed(4)
    singles = [O.infer(net_cpu[1], p, configs.SMPL_PARENTS, None, 3, feats=feats[i:i + 1]) for i in range(2)]
    for k in ("verts_mode", "R_samples", "verts_samples", "unc"):
        assert maxerr(both[k], torch.cat([s[k] for s in singles])) <= 5e-6, k   # BLAS blocking differs with batch size


def test_model_pkl_writer_round_trips_through_the_loader(tmp_path):
    """tests/golden/make_smpl_golden.py hands smplx the seeded synthetic models as SMPL_<GENDER>.pkl; the file it writes must be what
    smpl_data.load_smpl_pkl (the mirror of smplx's own reader) turns back into the same arrays."""
    import importlib.util
    import os
    from hierarchicalprobabilistic3dhuman_amd import smpl_data
    spec = importlib.util.spec_from_file_location("make_smpl_golden", os.path.join(os.path.dirname(__file__), "golden", "make_smpl_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    model = smpl_data.synthetic_smpl_model(1)
    path = gen.write_model_pkl(model, str(tmp_path / "SMPL_MALE.pkl"))
    back = smpl_data.resolve_smpl_model(str(tmp_path), gender="male")
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):
        assert np.array_equal(np.asarray(model[k], np.float64), back[k]), k
    assert smpl_data.parents_from_kintree(back["kintree_table"]).tolist() == configs.SMPL_PARENTS
    betas, aa, transl = gen.seeded_inputs(0)
    assert betas.shape == (8, 10) and aa.shape == (8, 24, 3) and transl.shape == (8, 3) and not aa[0].any()
    assert os.path.getsize(path) > 0


def test_oracle_smpl_matches_the_reference_class_on_smplx(smplx_golden):
    """A10 / A11 pinned: oracle/ref_cpu.smpl_forward against models/smpl_official.py:27-41 running on the installed smplx
    (fixture: tests/golden/make_smpl_golden.py).  Skips where the fixture does not exist."""
    fix = smplx_golden
    models = smplx_golden_models(fix)
    assert models
    for tag, gender, model in models:
        key = "%s_%s_" % (tag, gender)
        from hierarchicalprobabilistic3dhuman_amd import smpl_data
        p = O.SMPLParams(model, smpl_data.load_extra_joint_regressors(None), configs.SMPLX_EXTRA_VERTEX_IDS)
        t = lambda name: torch.from_numpy(fix[key + name])
        betas, aa, transl, R = t("betas"), t("aa"), t("transl"), t("rotmats")
        M = betas.shape[0]
        sel = torch.from_numpy(fix[key + "vertex_ids"]) if key + "vertex_ids" in fix else None
        assert fix[key + "parents"].tolist()[1:] == configs.SMPL_PARENTS[1:]
        assert maxerr(O.batch_rodrigues(aa.reshape(-1, 3)).reshape(M, 24, 3, 3), R) <= 1e-6
        outs = {
            "rotmat": O.smpl_forward(p, betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False),
            "aa": O.smpl_forward(p, betas=betas, body_pose=aa[:, 1:].reshape(M, 69), global_orient=aa[:, 0]),
            "tpose": O.smpl_forward(p, betas=betas[:1]),
            "transl": O.smpl_forward(p, betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False, transl=transl),
        }
        for name, o in outs.items():
            v = o["vertices"] if sel is None else o["vertices"][:, sel]
            assert maxerr(v, t(name + "_verts")) <= 2e-5, (key, name)
            assert maxerr(o["joints"], t(name + "_joints")) <= 2e-5, (key, name)
            s64 = o["vertices"].double().abs().sum(dim=(1, 2))
            assert maxerr(s64, t(name + "_verts_sum64")) <= 1e-5 * float(s64.max()), (key, name)

"""Generates the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (only possible in the build
container, where /root/reference exists).  The reference's source never leaves that container: what is
committed is data -- seeded inputs' recipes and the reference's outputs.

    python tests/golden/make_golden.py

Recipes (so that tests can rebuild the inputs without the reference):
  net weights : default initialisation of PoseMFShapeGaussianNet under torch.manual_seed(0)
                (asserted here to be identical, tensor for tensor, between the reference class and ours)
  net input   : torch.rand(2,18,256,256, generator=torch.Generator().manual_seed(0))
  sampler     : torch.manual_seed(0) then pose_matrix_fisher_sampling_torch(U,S,V, N, sample_on_cpu=True)
  A9          : torch.manual_seed(5) then the reference's compute_vertex_uncertainties_by_poseMF_shapeGaussian_sampling
                (utils/sampling_utils.py:146-192) on image 0 of the net outputs, N = 4, both use_mean_shape values, with
                the oracle's SMPL (synthetic model, seed 0) injected as ``smpl_model`` -- smplx itself is absent, so this
                pins the function's own logic (sampling call, shape expand / sample, global_orient expand, mean, norm)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.modules.setdefault("cv2", types.ModuleType("cv2"))     # utils/rigid_transform_utils.py:1 imports cv2; unused on the path
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

from models.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet as RefNet  # noqa: E402
from utils.sampling_utils import pose_matrix_fisher_sampling_torch as ref_sampling  # noqa: E402
from utils.sampling_utils import bingham_sampling_for_matrix_fisher_torch as ref_bingham  # noqa: E402
from utils.sampling_utils import compute_vertex_uncertainties_by_poseMF_shapeGaussian_sampling as ref_vertex_unc  # noqa: E402
import utils.rigid_transform_utils as ref_rtu  # noqa: E402
from losses.matrix_fisher_loss import LogMFNormConstant  # noqa: E402

from hierarchicalprobabilistic3dhuman_amd import configs  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet as OurNet  # noqa: E402


def main():
    cfg = configs.get_cfg_defaults()
    parents = configs.SMPL_PARENTS
    torch.manual_seed(0)
    ref = RefNet(parents, cfg).eval()
    torch.manual_seed(0)
    ours = OurNet(parents, cfg).eval()
    a, b = ref.state_dict(), ours.state_dict()
    assert list(a.keys()) == list(b.keys()) and all(torch.equal(a[k], b[k]) for k in a), "weight recipe broken"
    assert {int(k): list(v) for k, v in ref.parents_dict.items()} == {int(k): list(v) for k, v in ours.parents_dict.items()}

    out = {}
    # ---- net: encoder features + 8-tuple on a seeded (2,18,256,256) input; plus the B=1 plumbing case ----
    x = torch.rand(2, 18, 256, 256, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        feats = ref.image_encoder(x)
        F_, U, S, V, mode, sd, glob, cam = ref(x)
    out.update(net_feats=feats, net_F=F_, net_U=U, net_S=S, net_V=V, net_mode=mode, net_shape_loc=sd.loc,
               net_shape_scale=sd.scale, net_glob=glob, net_cam=cam)
    # ---- sampler given those U,S,V, for N = 1 (config #1), 4 and 100 ----
    for N in (1, 4, 100):
        torch.manual_seed(0)
        out["sampler_R_N%d" % N] = ref_sampling(U, S, V, N, sample_on_cpu=True)
    # ---- A9: per-vertex uncertainty by sampling (utils/sampling_utils.py:146-192), oracle SMPL injected ----
    from types import SimpleNamespace
    from oracle import ref_cpu as O
    from hierarchicalprobabilistic3dhuman_amd import smpl_data
    smpl_params = O.SMPLParams(smpl_data.synthetic_smpl_model(0), smpl_data.load_extra_joint_regressors(None),
                               configs.SMPLX_EXTRA_VERTEX_IDS)

    def oracle_smpl(body_pose, global_orient, betas, pose2rot=True):
        o = O.smpl_forward(smpl_params, betas=betas, body_pose=body_pose, global_orient=global_orient, pose2rot=pose2rot)
        return SimpleNamespace(vertices=o["vertices"], joints=o["joints"])

    glob_R = ref_rtu.rot6d_to_rotmat(glob[:1])
    out["a9_glob_rotmats"] = glob_R
    for tag, mean_shape in (("mean", True), ("samp", False)):
        torch.manual_seed(5)
        with torch.no_grad():
            unc, vs, js = ref_vertex_unc(U[:1], S[:1], V[:1], torch.distributions.Normal(sd.loc[:1], sd.scale[:1]), glob_R, 4,
                                         oracle_smpl, use_mean_shape=mean_shape)
        assert unc.shape == (6890,) and vs.shape == (4, 6890, 3) and js.shape == (4, 90, 3)
        out["a9_%s_unc" % tag], out["a9_%s_joints" % tag] = unc, js
        out["a9_%s_verts_sub" % tag] = vs[:, ::10].contiguous()          # every 10th vertex keeps the fixture small
    # a concentration sweep with random proper U, V (SURVEY.md section 8(d) stress set)
    S_sweep = torch.tensor([[0., 0., 0.], [1.9, .78, .6], [5., 5., 5.], [20., 15., 10.], [100., 80., 50.],
                            [500., 400., 300.], [50., 1., .1]])
    g = torch.Generator().manual_seed(7)
    Q1, _ = torch.linalg.qr(torch.randn(7, 3, 3, generator=g))
    Q2, _ = torch.linalg.qr(torch.randn(7, 3, 3, generator=g))
    Q1[:, :, 2] *= torch.det(Q1)[:, None]
    Q2[:, :, 2] *= torch.det(Q2)[:, None]
    torch.manual_seed(1)
    out["sweep_U"], out["sweep_S"], out["sweep_V"] = Q1[None], S_sweep[None], Q2[None]
    out["sweep_R_N50"] = ref_sampling(Q1[None], S_sweep[None], Q2[None], 50, sample_on_cpu=True)
    # E[R] = U diag(d log c / d s) V^T: the reference's own normalising-constant gradient (losses/matrix_fisher_loss.py:172-192)
    Sg = S_sweep.clone().requires_grad_(True)
    LogMFNormConstant.apply(Sg).sum().backward()
    out["sweep_dlogc_dS"] = Sg.grad.detach()
    # bingham entry point
    A = torch.tensor([0., 2.76, 5., 5.36])
    torch.manual_seed(2)
    q, _ = ref_bingham(A, 16)
    out["bingham_A"], out["bingham_q_N16"] = A, q
    # ---- rotation conversions ----
    g = torch.Generator().manual_seed(3)
    x6 = torch.randn(5, 6, generator=g)
    quat = torch.randn(7, 4, generator=g)
    out.update(rot6d_in=x6, rot6d_out=ref_rtu.rot6d_to_rotmat(x6), quat_in=quat, quat_out=ref_rtu.quat_to_rotmat(quat),
               rotmat_to_rot6d_out=ref_rtu.rotmat_to_rot6d(ref_rtu.quat_to_rotmat(quat)))
    # ---- proxy-representation front end: Canny + heat-maps on a seeded 2x3x64x64 crop ----
    from models.canny_edge_detector import CannyEdgeDetector
    from utils.label_conversions import convert_2Djoints_to_gaussian_heatmaps_torch
    g = torch.Generator().manual_seed(11)
    rgb = torch.nn.functional.interpolate(torch.rand(2, 3, 24, 24, generator=g), size=(64, 64), mode="bilinear",
                                          align_corners=False) + 0.05 * torch.rand(2, 3, 64, 64, generator=g)
    j2d = torch.rand(2, 17, 2, generator=g) * 64
    with torch.no_grad():
        for tag, (nms, thr) in (("nms0", (True, 0.0)), ("nms2", (True, 0.2)), ("plain", (False, 0.1))):
            res = CannyEdgeDetector(non_max_suppression=nms, gaussian_filter_std=1.0, gaussian_filter_size=5, threshold=thr)(rgb)
            for k, v in res.items():
                if tag == "nms0" or k in ("thresholded_thin_edges", "thresholded_grad_magnitude"):
                    out["canny_%s_%s" % (tag, k)] = v
    out["canny_rgb"], out["heat_joints"] = rgb, j2d
    out["heat_out"] = convert_2Djoints_to_gaussian_heatmaps_torch(j2d, 64, 4.0)
    # ---- bounding-box crops of the predict front end (utils/image_utils.py:234-372): person box -> HRNet input (1.2x box,
    #      aspect fix), then HRNet input -> proxy-representation size with the 2D joints mapped along (predict/...:78-87) ----
    import utils.image_utils as ref_iu
    g = torch.Generator().manual_seed(21)
    img = torch.rand(1, 3, 120, 160, generator=g)
    c1 = ref_iu.batch_crop_pytorch_affine(input_wh=(160, 120), output_wh=(72, 96), num_to_crop=1, device="cpu", rgb=img,
                                          bbox_centres=torch.tensor([[55.0, 83.0]]), bbox_heights=torch.tensor([90.0]),
                                          bbox_widths=torch.tensor([40.0]), orig_scale_factor=1.2)["rgb"]
    jc = torch.rand(1, 17, 2, generator=g) * torch.tensor([72.0, 96.0])
    side = torch.tensor([96.0])
    c2 = ref_iu.batch_crop_pytorch_affine(input_wh=(72, 96), output_wh=(64, 64), num_to_crop=1, device="cpu", joints2D=jc, rgb=c1,
                                          bbox_centres=torch.tensor([[48.0, 36.0]]), bbox_heights=side, bbox_widths=side,
                                          orig_scale_factor=1.0)
    out.update(crop_img=img, crop1_rgb=c1, crop_joints_in=jc, crop2_rgb=c2["rgb"], crop2_joints=c2["joints2D"])
    # ---- heat-map arg-max (utils/label_conversions.py:127-155) on the case tests/test_frontend.py builds ----
    from utils.label_conversions import convert_heatmaps_to_2Djoints_coordinates_torch
    g4 = torch.Generator().manual_seed(4)
    jj = torch.rand(1, 17, 2, generator=g4) * 200 + 20
    hh = convert_2Djoints_to_gaussian_heatmaps_torch(jj.round(), 256, 4.0)
    hh[:, [7, 9]] = 0.0
    hh[:, 3, 10, 10] = hh[:, 3].max()
    out["argmax_joints"] = convert_heatmaps_to_2Djoints_coordinates_torch(hh)[0]
    # ---- sample ranking: the reference's joints2D_error_sorted_verts_sampling
    #      (utils/sampling_utils.py:195-233) itself, on the same generator stream (seed 4, continued): N = 12 samples, the two
    #      invisible joints and the arg-max tie of the heat-maps above, and one TIE between samples (2 and 7 share their joints
    #      but not their vertices).  pytorch3d is absent, so utils.rigid_transform_utils.so3_exponential_map (guarded import,
    #      :5-8) is set to the Rodrigues closed form pytorch3d publishes (fac1 = sin(t)/t, fac2 = (1 - cos t)/t^2,
    #      R = I + fac1 K + fac2 K^2, t = sqrt(clamp(|r|^2, 1e-4))) -- the same accommodation A9 uses for SMPL ----
    from utils.sampling_utils import joints2D_error_sorted_verts_sampling as ref_rank

    def so3_exponential_map(log_rot, eps=0.0001):
        nrms = (log_rot * log_rot).sum(1)
        ang = torch.clamp(nrms, eps).sqrt()
        inv = 1.0 / ang
        fac1 = inv * ang.sin()
        fac2 = inv * inv * (1.0 - ang.cos())
        K = torch.zeros(log_rot.shape[0], 3, 3, dtype=log_rot.dtype)
        x, y, z = log_rot.unbind(1)
        K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -z, y, z, -x, -y, x
        return fac1[:, None, None] * K + fac2[:, None, None] * torch.bmm(K, K) + torch.eye(3, dtype=log_rot.dtype)[None]

    ref_rtu.so3_exponential_map = so3_exponential_map
    assert torch.equal(hh, O.joints2d_to_gaussian_heatmaps(jj.round(), 256, 4.0).index_put_(
        (torch.tensor([0, 0]), torch.tensor([7, 9])), torch.tensor(0.0)).index_put_(
        (torch.tensor([0]), torch.tensor([3]), torch.tensor([10]), torch.tensor([10])), hh[0, 3].max())), "heat-map recipe"
    Nr = 12
    rj = torch.randn(Nr, 90, 3, generator=g4) * 0.4
    rv = torch.randn(Nr, 6890, 3, generator=g4)
    rj[7] = rj[2]
    rcam = torch.tensor([[0.9, 0.05, -0.1]])
    with torch.no_grad():
        rsorted = ref_rank(rv, rj, hh, rcam)
    order = torch.tensor([int((rv == rsorted[i]).all(dim=-1).all(dim=-1).nonzero()[0, 0]) for i in range(Nr)])
    assert sorted(order.tolist()) == list(range(Nr)) and torch.equal(rv[order], rsorted)
    out["rank_order"], out["rank_sorted_verts_sub"] = order, rsorted[:, ::53].contiguous()
    # ---- evaluation metrics: the reference tracker on a seeded synthetic scenario ----
    from metrics.eval_metrics_tracker import EvalMetricsTracker
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from metric_scenario import METRICS, make_frames
    tracker = EvalMetricsTracker(METRICS)
    tracker.initialise_metric_sums()
    tracker.initialise_per_frame_metric_lists()
    for pred, target in make_frames():
        tracker.update_per_batch(pred, target, 1)
    tracker.compute_final_metrics()                      # prints only (metrics/eval_metrics_tracker.py:332-362)
    per = lambda m: 6890 if "PVE" in m else 14
    out["metrics_final"] = torch.tensor([tracker.metric_sums[m] / (tracker.total_samples * per(m)) for m in METRICS],
                                        dtype=torch.float64)
    out["metrics_sums"] = torch.tensor([tracker.metric_sums[m] for m in METRICS], dtype=torch.float64)
    out["metrics_pve_pa_per_frame"] = torch.tensor(np.concatenate(tracker.per_frame_metrics["PVE-PA"]), dtype=torch.float64)
    # ---- sampler sizes that take SEVERAL wavefronts per (image, joint) call in hps_mf_sample (waves = ceil(2N / 256):
    #      N = 129 -> 2, 300 -> 3, 1000 -> 8, the BASELINE configs[4] size): "the first N accepted, in proposal order"
    #      (utils/sampling_utils.py:61-66) across wavefront boundaries.  N = 129 is kept whole; of N = 300 / 1000 every
    #      3rd / 8th sample and the last one are kept, plus the float64 sum over ALL samples (any ordering or selection
    #      error moves it).  Then a starved case -- oversampling_ratio = 2, N = 200 (2 wavefronts) on broad concentrations:
    #      rounds with fewer than N accepts are discarded and redrawn (:68-69), 23 of them here, up to 9 for one call ----
    for N, step in ((129, 1), (300, 3), (1000, 8)):
        torch.manual_seed(40 + N)
        Rn = ref_sampling(U, S, V, N, sample_on_cpu=True)
        keep = sorted(set(range(0, N, step)) | {N - 1})
        out["sampler_R_N%d_keep" % N] = torch.tensor(keep)
        out["sampler_R_N%d_sub" % N] = Rn[:, keep].contiguous()
        out["sampler_R_N%d_sum" % N] = Rn.double().sum(1)
    g = torch.Generator().manual_seed(5)
    Fs = torch.randn(2, 23, 3, 3, generator=g) * 2.0 + torch.eye(3)
    Us, Ss, Vhs = torch.linalg.svd(Fs)
    Vs = Vhs.transpose(-1, -2).contiguous()
    torch.manual_seed(9)
    _, (_, _, disc) = O.pose_matrix_fisher_sampling(Us, Ss, Vs, 200, oversampling_ratio=2, return_noise=True)
    assert int(disc.sum()) >= 10 and int(disc.max()) >= 3, "starved case no longer discards rounds"
    torch.manual_seed(9)
    out["starved_U"], out["starved_S"], out["starved_V"] = Us, Ss, Vs
    out["starved_R_N200"] = ref_sampling(Us, Ss, Vs, 200, oversampling_ratio=2, sample_on_cpu=True)
    out["starved_next_rand"] = torch.rand(1)          # the host generator's state after the reference loop
    out["starved_discarded"] = disc
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"),
                        **{k: v.detach().numpy() for k, v in out.items()})
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), {k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()

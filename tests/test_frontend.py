"""Proxy-representation front end: Canny edge detector and 2D-joint heat-maps.
CPU: oracle vs golden vectors from the imported reference.  GPU: HIP kernels vs golden vectors and oracle.

Stated tolerance: 2e-6 on blurred image / gradient magnitude / heat-maps.  Orientation bins and the non-max
suppression decision are discontinuous: a pixel may differ only where the reference's own decision sits on an fp32
rounding tie (orientation exactly on a 22.5 degree boundary, or equal neighbouring magnitudes); such pixels are
bounded to <= 0.1 % and must have matching magnitudes."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from conftest import maxerr


def _golden_cases(golden):
    return (("nms0", True, 0.0), ("nms2", True, 0.2), ("plain", False, 0.1))


def test_oracle_canny_matches_reference(golden):
    rgb = golden["canny_rgb"]
    for tag, nms, thr in _golden_cases(golden):
        out = O.canny_edge_detector(rgb, nms, 1.0, 5, thr)
        for k, v in out.items():
            key = "canny_%s_%s" % (tag, k)
            if key in golden:
                assert maxerr(v, golden[key]) == 0.0, key


def test_oracle_heatmaps_and_proxy_rep_match_reference(golden):
    heat = O.joints2d_to_gaussian_heatmaps(golden["heat_joints"], 64, 4.0)
    assert maxerr(heat, golden["heat_out"]) == 0.0
    vis = torch.ones(2, 17)
    vis[:, [7, 9]] = 0
    proxy = O.proxy_representation(golden["canny_rgb"], golden["heat_joints"], vis, img_wh=64)
    assert proxy.shape == (2, 18, 64, 64)
    assert maxerr(proxy[:, :1], golden["canny_nms0_thresholded_thin_edges"]) == 0.0
    assert float(proxy[:, [8, 10]].abs().max()) == 0.0 and maxerr(proxy[:, 1], golden["heat_out"][:, 0]) == 0.0


def _fraction_differing(a, b, tol):
    return float(((a.cpu() - b).abs() > tol).float().mean())


@pytest.mark.gpu
def test_canny_kernel_matches_reference(dev, golden):
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    rgb = golden["canny_rgb"].to(dev)
    for tag, nms, thr in _golden_cases(golden):
        det = CannyEdgeDetector(non_max_suppression=nms, gaussian_filter_std=1.0, gaussian_filter_size=5, threshold=thr).to(dev)
        out = det(rgb)
        assert set(out) == ({"blurred_img", "grad_magnitude", "grad_orientation", "thresholded_grad_magnitude"}
                            | ({"thin_edges", "thresholded_thin_edges"} if nms else set()))
        if tag == "nms0":
            assert maxerr(out["blurred_img"], golden["canny_nms0_blurred_img"]) <= 2e-6
            assert maxerr(out["grad_magnitude"], golden["canny_nms0_grad_magnitude"]) <= 2e-6
            assert _fraction_differing(out["grad_orientation"], golden["canny_nms0_grad_orientation"], 1e-3) <= 1e-3
            assert _fraction_differing(out["thin_edges"], golden["canny_nms0_thin_edges"], 2e-6) <= 1e-3
        key = "canny_%s_%s" % (tag, "thresholded_thin_edges" if nms else "thresholded_grad_magnitude")
        got = out["thresholded_thin_edges" if nms else "thresholded_grad_magnitude"]
        assert _fraction_differing(got, golden[key], 2e-6) <= 1e-3, key


@pytest.mark.gpu
@pytest.mark.parametrize("nms,thr", [(True, 0.0), (True, 0.2), (False, 0.1)])
def test_edge_map_entry_point_equals_the_detectors_dict_entry(nms, thr, dev, golden):
    """hps_canny_edge_map writes the one output the predict front end uses (predict/...:92-93) straight into channel 0 of the
    proxy representation: bit-identical to that entry of the full detector's dict, other channels untouched; the heat-map
    kernel then fills channels 1..17 in place -- the whole tensor equals the two-step construction bit for bit."""
    from hierarchicalprobabilistic3dhuman_amd import configs
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    from hierarchicalprobabilistic3dhuman_amd.label_conversions import make_proxy_representation
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import proxy_representation
    rgb = torch.cat([golden["canny_rgb"], golden["canny_rgb"].flip(-1)]).to(dev)                 # (4,3,64,64)
    det = CannyEdgeDetector(nms, 1.0, 5, thr).to(dev)
    full = det(rgb)
    want_edge = full["thresholded_thin_edges" if nms else "thresholded_grad_magnitude"]
    out = torch.full((4, 18, 64, 64), -7.0, device=dev)
    det.edge_map_into(rgb, out)
    assert torch.equal(out[:, :1], want_edge) and bool((out[:, 1:] == -7.0).all())
    j = torch.cat([golden["heat_joints"], golden["heat_joints"].flip(0)]).to(dev)
    vis = torch.ones(4, 17, device=dev)
    vis[:, [7, 9]] = 0
    cfg = configs.get_cfg_defaults()
    cfg.DATA.PROXY_REP_SIZE, cfg.DATA.EDGE_NMS, cfg.DATA.EDGE_THRESHOLD = 64, nms, thr
    fused = proxy_representation(rgb, j, vis, det, cfg)
    assert torch.equal(fused, make_proxy_representation(want_edge, j, vis, 64, cfg.DATA.HEATMAP_GAUSSIAN_STD))
    # the CONFIG selects the dict entry (predict/...:93), not the detector: a detector built with NMS and
    # cfg.DATA.EDGE_NMS = False yields 'thresholded_grad_magnitude'; the converse is the reference's KeyError
    cfg.DATA.EDGE_NMS = not nms
    if nms:
        other = proxy_representation(rgb, j, vis, det, cfg)
        assert torch.equal(other[:, :1], full["thresholded_grad_magnitude"]) and torch.equal(other[:, 1:], fused[:, 1:])
    else:
        with pytest.raises(KeyError):
            proxy_representation(rgb, j, vis, det, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 256, 256), (3, 3, 70, 45), (2, 1, 33, 32), (1, 3, 40, 300), (2, 1, 21, 522), (1, 2, 37, 50),
                                   (1, 3, 9, 252), (1, 3, 3, 8), (2, 1, 65, 64), (3, 3, 17, 260), (1, 3, 1, 4)])
def test_canny_kernel_matches_oracle_at_borders_and_odd_sizes(shape, dev):
    """Shapes wider than 256 run the row-marching kernel in column blocks of 248 valid columns (300 -> 2 blocks, 522 -> 3), W % 4 != 0
    takes its scalar-load instantiation, two channels the tile kernel, 9 rows a single ragged strip.  Strips of 1, 3 and 8 + 1 rows
    (65 rows in strips of 8: the last one has a single row) leave the row loop by each of its early exits; the warm-up steps run
    above the image there (rows -4 .. -1 of the first strip)."""
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    g = torch.Generator().manual_seed(sum(shape))
    B, C, H, W = shape
    img = torch.nn.functional.interpolate(torch.rand(B, C, max(H // 4, 2), max(W // 4, 2), generator=g), size=(H, W),
                                          mode="bilinear", align_corners=False) + 0.05 * torch.rand(B, C, H, W, generator=g)
    ref = O.canny_edge_detector(img, True, 1.0, 5, 0.05)
    out = CannyEdgeDetector(True, 1.0, 5, 0.05).to(dev)(img.to(dev))
    assert maxerr(out["blurred_img"], ref["blurred_img"]) <= 2e-6
    assert maxerr(out["grad_magnitude"], ref["grad_magnitude"]) <= 2e-6
    for k in ("thin_edges", "thresholded_thin_edges", "thresholded_grad_magnitude"):
        assert _fraction_differing(out[k], ref[k], 2e-6) <= 2e-3, k


@pytest.mark.gpu
def test_canny_rows_kernel_on_random_shapes(dev):
    """Sixteen seeded random shapes (1 or 3 channels, 1-5 images, 1-140 rows, 4-600 columns, with and without NMS): every strip height
    the launcher can choose, ragged last strips, one to three column blocks, both load forms -- against the oracle, and the edge-map
    entry point against the full detector bit for bit."""
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    rng = np.random.RandomState(4)
    for _ in range(16):
        B, C = int(rng.randint(1, 6)), int(rng.choice([1, 3]))
        H = int(rng.randint(1, 141))
        W = 4 * int(rng.randint(1, 151)) if rng.randint(0, 2) else int(rng.randint(4, 601))      # half of them with 16-byte rows
        nms, thr = bool(rng.randint(0, 2)), float(rng.choice([0.0, 0.05, 0.2]))
        g = torch.Generator().manual_seed(B * 1000003 + H * 1009 + W)
        img = torch.nn.functional.interpolate(torch.rand(B, C, max(H // 4, 2), max(W // 4, 2), generator=g), size=(H, W),
                                              mode="bilinear", align_corners=False) + 0.05 * torch.rand(B, C, H, W, generator=g)
        ref = O.canny_edge_detector(img, nms, 1.0, 5, thr)
        det = CannyEdgeDetector(nms, 1.0, 5, thr).to(dev)
        out = det(img.to(dev))
        tag = (B, C, H, W, nms, thr)
        assert maxerr(out["blurred_img"], ref["blurred_img"]) <= 2e-6, tag
        assert maxerr(out["grad_magnitude"], ref["grad_magnitude"]) <= 2e-6, tag
        keys = ("thin_edges", "thresholded_thin_edges", "thresholded_grad_magnitude") if nms else ("thresholded_grad_magnitude",)
        for k in keys:
            assert _fraction_differing(out[k], ref[k], 2e-6) <= max(2e-3, 2.0 / (B * H * W)), (k,) + tag
        edge = torch.full((B, 2, H, W), -1.0, device=dev)
        det.edge_map_into(img.to(dev), edge)
        assert torch.equal(edge[:, 0], out["thresholded_thin_edges" if nms else "thresholded_grad_magnitude"][:, 0]), tag
        assert float(edge[:, 1].max()) == -1.0 and float(edge[:, 1].min()) == -1.0, tag


@pytest.mark.gpu
def test_heatmaps_and_proxy_representation(dev, golden):
    from hierarchicalprobabilistic3dhuman_amd import configs
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    from hierarchicalprobabilistic3dhuman_amd.label_conversions import convert_2Djoints_to_gaussian_heatmaps_torch
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import proxy_representation
    j = golden["heat_joints"].to(dev)
    assert maxerr(convert_2Djoints_to_gaussian_heatmaps_torch(j, 64, 4.0), golden["heat_out"]) <= 2e-6
    cfg = configs.get_cfg_defaults()
    cfg.DATA.PROXY_REP_SIZE = 64
    det = CannyEdgeDetector(cfg.DATA.EDGE_NMS, cfg.DATA.EDGE_GAUSSIAN_STD, cfg.DATA.EDGE_GAUSSIAN_SIZE, cfg.DATA.EDGE_THRESHOLD).to(dev)
    vis = torch.ones(2, 17, dtype=torch.bool)
    vis[:, [7, 9]] = False
    proxy = proxy_representation(golden["canny_rgb"].to(dev), j, vis.to(dev), det, cfg)
    want = O.proxy_representation(golden["canny_rgb"], golden["heat_joints"], vis.float(), img_wh=64)
    assert proxy.shape == (2, 18, 64, 64)
    assert maxerr(proxy[:, 1:], want[:, 1:]) <= 2e-6
    assert _fraction_differing(proxy[:, :1], want[:, :1], 2e-6) <= 1e-3


@pytest.mark.gpu
def test_heatmap_argmax_and_sample_ranking(dev, golden):
    """utils/label_conversions.py:127-155 and utils/sampling_utils.py:195-233."""
    from hierarchicalprobabilistic3dhuman_amd.label_conversions import (convert_heatmaps_to_2Djoints_coordinates_torch,
                                                                        ALL_JOINTS_TO_COCO_MAP)
    from hierarchicalprobabilistic3dhuman_amd.sampling_utils import joints2D_error_sorted_verts_sampling
    g = torch.Generator().manual_seed(4)
    j2d = torch.rand(1, 17, 2, generator=g) * 200 + 20
    heat = O.joints2d_to_gaussian_heatmaps(j2d.round(), 256, 4.0)
    heat[:, [7, 9]] = 0.0                                                     # two undetected joints
    heat[:, 3, 10, 10] = heat[:, 3].max()                                     # a tie: the first index must win
    j_ref, v_ref = O.heatmaps_to_joints2d(heat)
    j, v = convert_heatmaps_to_2Djoints_coordinates_torch(heat.to(dev))
    assert torch.equal(j.cpu(), j_ref) and torch.equal(v.cpu(), v_ref)
    assert torch.equal(golden["argmax_joints"], j_ref)                         # the reference's own arg-max on this case (make_golden.py)
    # ranking: the reference function's own ordering on this case (tests/golden/make_golden.py ran
    # utils/sampling_utils.py:195-233 with pytorch3d's so3_exponential_map restated): two invisible joints, and samples 2
    # and 7 tie (same joints, different vertices) -- the earlier sample must come first, as in the reference's CPU sort
    verts, joints, cam = _ranking_case(g)
    got = joints2D_error_sorted_verts_sampling(verts.to(dev), joints.to(dev), heat.to(dev), cam.to(dev))
    order = golden["rank_order"]
    assert torch.equal(got.cpu(), verts[order])
    assert torch.equal(got.cpu()[:, ::53], golden["rank_sorted_verts_sub"])
    want, o_order = O.joints2d_error_sorted(verts, joints, heat, cam, ALL_JOINTS_TO_COCO_MAP)
    assert torch.equal(got.cpu(), want)


def _ranking_case(g, N=12):
    """Continues the generator stream of the arg-max case (seed 4, after rand(1,17,2)), exactly like make_golden.py."""
    joints = torch.randn(N, 90, 3, generator=g) * 0.4
    verts = torch.randn(N, 6890, 3, generator=g)
    joints[7] = joints[2]
    return verts, joints, torch.tensor([[0.9, 0.05, -0.1]])


def test_oracle_sample_ranking_matches_reference(golden):
    """SURVEY 8(f)4 pinned: the oracle's restatement (the 180 degree flip as the exact diag(1,-1,-1)) orders the samples like the
    reference function did (golden rank_order), the tie included."""
    from hierarchicalprobabilistic3dhuman_amd.label_conversions import ALL_JOINTS_TO_COCO_MAP
    g = torch.Generator().manual_seed(4)
    j2d = torch.rand(1, 17, 2, generator=g) * 200 + 20
    heat = O.joints2d_to_gaussian_heatmaps(j2d.round(), 256, 4.0)
    heat[:, [7, 9]] = 0.0
    heat[:, 3, 10, 10] = heat[:, 3].max()
    verts, joints, cam = _ranking_case(g)
    want, order = O.joints2d_error_sorted(verts, joints, heat, cam, ALL_JOINTS_TO_COCO_MAP)
    assert order.tolist() == golden["rank_order"].tolist()
    assert torch.equal(want[:, ::53], golden["rank_sorted_verts_sub"])


# ---------------------------------------------------------------------------------------------
# The reference-signature predict front end (predict/predict_poseMF_shapeGaussian_net.py:61-100)
# ---------------------------------------------------------------------------------------------
def _crop_case(golden):
    return (golden["crop_img"], torch.tensor([[55.0, 83.0]]), torch.tensor([90.0]), torch.tensor([40.0]), golden["crop_joints_in"])


def test_oracle_crops_match_reference(golden):
    """utils/image_utils.py:234-372 (batch_crop_pytorch_affine) as called by predict_hrnet (:86-95) and by the predict harness
    (:78-87): the oracle's restatement against the reference's outputs."""
    img, centre, h, w, jc = _crop_case(golden)
    c1, _ = O.batch_crop_affine((160, 120), (72, 96), img, centre, h, w, scale_factor=1.2)
    assert maxerr(c1, golden["crop1_rgb"]) <= 1e-6
    side = torch.tensor([96.0])
    c2, j2 = O.batch_crop_affine((72, 96), (64, 64), c1, torch.tensor([[48.0, 36.0]]), side, side.clone(), joints2D=jc, scale_factor=1.0)
    assert maxerr(c2, golden["crop2_rgb"]) <= 1e-6 and maxerr(j2, golden["crop2_joints"]) <= 1e-5


def test_package_crops_match_reference(golden):
    """The package's crop glue (device-agnostic torch ops, no libhps kernel) against the same reference outputs."""
    from hierarchicalprobabilistic3dhuman_amd.image_utils import batch_crop_pytorch_affine
    img, centre, h, w, jc = _crop_case(golden)
    c1 = batch_crop_pytorch_affine((160, 120), (72, 96), 1, "cpu", rgb=img, bbox_centres=centre, bbox_heights=h, bbox_widths=w,
                                   orig_scale_factor=1.2)["rgb"]
    assert maxerr(c1, golden["crop1_rgb"]) <= 1e-6
    side = torch.tensor([96.0])
    c2 = batch_crop_pytorch_affine((72, 96), (64, 64), 1, "cpu", joints2D=jc, rgb=c1, bbox_centres=torch.tensor([[48.0, 36.0]]),
                                   bbox_heights=side, bbox_widths=side, orig_scale_factor=1.0)
    assert maxerr(c2["rgb"], golden["crop2_rgb"]) <= 1e-6 and maxerr(c2["joints2D"], golden["crop2_joints"]) <= 1e-5
    with pytest.raises(NotImplementedError):
        batch_crop_pytorch_affine((72, 96), (64, 64), 1, "cpu", rgb=c1)            # boxes from IUV / seg: training only


class _ToyHRNet(torch.nn.Module):
    """Stand-in for the injected 2D keypoint detector (the real HRNet is out of scope): (1,3,h,w) -> (1,17,h/4,w/4)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(123)
        self.conv = torch.nn.Conv2d(3, 17, kernel_size=8, stride=4, padding=2)

    def forward(self, x):
        return self.conv(x)


def _hrnet_cfg():
    from types import SimpleNamespace
    return SimpleNamespace(MODEL=SimpleNamespace(IMAGE_SIZE=[288, 384], HEATMAP_SIZE=[72, 96]))


@pytest.mark.gpu
def test_predict_harness_with_the_reference_signature(dev, net_gpu, smpl_gpu, tmp_path):
    """run_predict.py:77-89 calls predict_poseMF_shapeGaussian_net(pose_shape_model, pose_shape_cfg, smpl_model, hrnet_model,
    hrnet_cfg, edge_detect_model, device, image_dir, save_dir, ...) -- no extra keyword.  The front end (:61-100) must run
    on the injected detector and give the oracle's proxy representation."""
    import os
    import numpy as np
    from PIL import Image
    from hierarchicalprobabilistic3dhuman_amd import configs
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    from hierarchicalprobabilistic3dhuman_amd import predict_poseMF_shapeGaussian_net as P
    cfg = configs.get_cfg_defaults()
    rs = np.random.RandomState(3)
    base = rs.rand(30, 40, 3)
    img = (np.kron(base, np.ones((10, 10, 1))) * 255).astype(np.uint8)                 # 300 x 400 blocky RGB image
    image_dir, save_dir = tmp_path / "images", tmp_path / "out"
    os.makedirs(image_dir)
    Image.fromarray(img).save(str(image_dir / "person.png"))
    hrnet = _ToyHRNet().to(dev)
    edge = CannyEdgeDetector(non_max_suppression=cfg.DATA.EDGE_NMS, gaussian_filter_std=cfg.DATA.EDGE_GAUSSIAN_STD,
                             gaussian_filter_size=cfg.DATA.EDGE_GAUSSIAN_SIZE, threshold=cfg.DATA.EDGE_THRESHOLD).to(dev)
    # the proxy representation the front end builds, against the oracle
    fe = P._reference_front_end(cfg, hrnet, _hrnet_cfg(), edge, None, 0.75, dev)
    proxy = fe(str(image_dir / "person.png"))
    assert proxy.shape == (1, 18, 256, 256)
    image_cpu = torch.from_numpy(img.transpose(2, 0, 1).copy()).float() / 255.0
    with torch.no_grad():
        want, aux = O.predict_front_end(image_cpu, _ToyHRNet(), (288, 384), (72, 96))
    assert maxerr(proxy[:, 1:], want[:, 1:]) <= 1e-4                                  # heat-maps (joint positions agree)
    diff = (proxy[:, :1].cpu() - want[:, :1]).abs() > 1e-4
    assert float(diff.float().mean()) <= 2e-3                                          # edge map: rounding ties of the NMS only
    # the harness, called positionally like run_predict.py does
    P.predict_poseMF_shapeGaussian_net(net_gpu, cfg, smpl_gpu, hrnet, _hrnet_cfg(), edge, dev, str(image_dir), str(save_dir))
    saved = torch.load(str(save_dir / "person.pt"))
    assert saved["verts_mode"].shape == (6890, 3) and saved["unc"].shape == (6890,) and torch.isfinite(saved["verts_mode"]).all()

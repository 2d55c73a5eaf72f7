"""Joint-layout constants and heat-map generation on the hot path, mirroring the reference's
utils/label_conversions.py (names and values of :17-20; function of :105-124)."""
import torch

from . import _capi

# Index maps into the 90-joint superset SMPL.forward returns (24 kinematic + 21 vertex picks + 9 + 19 + 17):
# utils/label_conversions.py:17-20.  They pin the joint layout contract of smpl_official.SMPL.
ALL_JOINTS_TO_COCO_MAP = [24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8]
ALL_JOINTS_TO_H36M_MAP = list(range(73, 90))
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J14 = H36M_TO_J17[:14]


def convert_2Djoints_to_gaussian_heatmaps_torch(joints2D, img_wh, std=4):
    """utils/label_conversions.py:105-124: (B,N,2) joints (u = column, v = row) -> (B,N,img_wh,img_wh)."""
    return make_proxy_representation(None, joints2D, None, img_wh, std)[:, 1:]


def make_proxy_representation(edge, joints2D, joints2D_visib, img_wh, std=4.0, out=None):
    """predict/predict_poseMF_shapeGaussian_net.py:93-100 in one kernel: channel 0 = ``edge`` (B,1,D,D) (zeros if
    None), channels 1..N = visibility-masked Gaussian heat-maps.  Returns (B, N+1, D, D) fp32.

    ``out``: an existing (B, N+1, D, D) tensor whose channel 0 already holds the edge map (CannyEdgeDetector.edge_map_into);
    only the heat-map channels are written then (``edge`` must be None)."""
    _capi.require_device(joints2D, "joints2D")
    j = _capi.f32c(joints2D)
    B, N = j.shape[:2]
    dev = j.device
    e = None
    if out is not None:
        assert edge is None and out.shape == (B, N + 1, img_wh, img_wh) and out.is_contiguous() and out.dtype == torch.float32
    else:
        if edge is None:
            edge = torch.zeros(B, 1, img_wh, img_wh, device=dev, dtype=torch.float32)
        e = _capi.f32c(edge)
        assert e.shape == (B, 1, img_wh, img_wh)
        out = torch.empty(B, N + 1, img_wh, img_wh, device=dev, dtype=torch.float32)
    vis = None if joints2D_visib is None else _capi.f32c(joints2D_visib.to(dev).float()).reshape(B, N)
    P = _capi.ptr
    _capi.call("hps_proxy_rep", P(e) if e is not None else None, P(j), P(vis) if vis is not None else None, P(out), B, N,
               img_wh, img_wh, float(std), _capi.stream())
    return out


def convert_heatmaps_to_2Djoints_coordinates_torch(joints2D_heatmaps, eps=1e-6):
    """utils/label_conversions.py:127-155: (N,K,H,W) heat-maps -> joints2D (N,K,2) = arg-max (x, y), joints2D_vis (N,K) bool
    (max > eps); invisible joints are (-1,-1)."""
    _capi.require_device(joints2D_heatmaps, "joints2D_heatmaps")
    h = _capi.f32c(joints2D_heatmaps)
    N, K, H, W = h.shape
    j = torch.empty(N, K, 2, device=h.device, dtype=torch.float32)
    v = torch.empty(N, K, device=h.device, dtype=torch.float32)
    _capi.call("hps_heatmaps_to_joints2d", _capi.ptr(h), _capi.ptr(j), _capi.ptr(v), N * K, H, W, float(eps), _capi.stream())
    return j, v > 0.5

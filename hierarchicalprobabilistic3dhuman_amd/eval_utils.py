"""Point-set alignment used by the evaluation metrics, mirroring the reference's utils/eval_utils.py on device
tensors (the reference works on numpy arrays on the host)."""
import torch

from . import _capi

MODE_RAW, MODE_SC, MODE_PA = 0, 1, 2


def pointset_errors(pred, target, mode=MODE_RAW, group=1, return_transformed=False):
    """Per-set sums of point-wise L2 errors.  pred (S,P,3); prediction s is compared with target[s // group]
    (target (ceil(S/group),P,3)).  Returns err_sum (S,) float64 [, transformed (S,P,3)]."""
    _capi.require_device(pred, "pred")
    _capi.require_device(target, "target")
    p, t = _capi.f32c(pred), _capi.f32c(target)
    S, P = p.shape[:2]
    assert t.shape[1] == P and t.shape[0] * group >= S
    dev = p.device
    stats = torch.empty(S, 17, device=dev, dtype=torch.float64)
    xf = torch.empty(S, 12, device=dev, dtype=torch.float32)
    err = torch.empty(S, device=dev, dtype=torch.float64)
    out = torch.empty(S, P, 3, device=dev, dtype=torch.float32) if return_transformed else None
    _capi.call("hps_pointset_errors", _capi.ptr(p), _capi.ptr(t), S, group, P, mode, _capi.ptr(stats, torch.float64),
               _capi.ptr(xf), _capi.ptr(err, torch.float64), _capi.ptr(out) if out is not None else None, _capi.stream())
    return (err, out) if return_transformed else err


def procrustes_analysis_batch(S1, S2):
    """utils/eval_utils.py:62-67 (batched compute_similarity_transform, :11-59): S1, S2 (B,N,3) -> S1 aligned to S2."""
    return pointset_errors(S1, S2, MODE_PA, 1, True)[1]


def scale_and_translation_transform_batch(P, T):
    """utils/eval_utils.py:70-89."""
    return pointset_errors(P, T, MODE_SC, 1, True)[1]

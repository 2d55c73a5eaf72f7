"""Rotation conversions of the inference path, same names and argument meaning as the reference's
utils/rigid_transform_utils.py; each call is one HIP kernel behind the C ABI.

Not mirrored (out of scope): the cv2 / pytorch3d based ``aa_rotate_*`` helpers used
for visualisation and label flipping.
"""
import torch

from . import _capi


def rot6d_to_rotmat(x):
    """utils/rigid_transform_utils.py:80-94: (B,6) or (B,24*6) -> (B',3,3).

    The reference's ``torch.cross`` without ``dim`` (line 93) is silently wrong at exactly B == 3;
    the cross product along dim 1 is computed here for every B."""
    _capi.require_device(x, "rot6d_to_rotmat input")
    x6 = _capi.f32c(x).reshape(-1, 6)
    out = torch.empty(x6.shape[0], 3, 3, device=x.device, dtype=torch.float32)
    _capi.call("hps_rot6d_to_rotmat", _capi.ptr(x6), _capi.ptr(out), x6.shape[0], _capi.stream())
    return out


def rotmat_to_rot6d(R, stack_columns=False):
    """utils/rigid_transform_utils.py:97-110 (pure indexing; used once at net construction)."""
    if stack_columns:
        return torch.cat([R[:, :, 0], R[:, :, 1]], dim=1)
    return R[:, :, :2].contiguous().view(-1, 6)


def quat_to_rotmat(quat):
    """utils/rigid_transform_utils.py:113-133: (B,4) (w,x,y,z) -> (B,3,3)."""
    _capi.require_device(quat, "quat_to_rotmat input")
    q = _capi.f32c(quat).reshape(-1, 4)
    out = torch.empty(q.shape[0], 3, 3, device=quat.device, dtype=torch.float32)
    _capi.call("hps_quat_to_rotmat", _capi.ptr(q), _capi.ptr(out), q.shape[0], _capi.stream())
    return out


def batch_rodrigues(rot_vecs):
    """smplx.lbs.batch_rodrigues as imported by the reference (predict/...:7): (N,3) -> (N,3,3)."""
    _capi.require_device(rot_vecs, "batch_rodrigues input")
    r = _capi.f32c(rot_vecs).reshape(-1, 3)
    out = torch.empty(r.shape[0], 3, 3, device=rot_vecs.device, dtype=torch.float32)
    _capi.call("hps_batch_rodrigues", _capi.ptr(r), _capi.ptr(out), r.shape[0], _capi.stream())
    return out

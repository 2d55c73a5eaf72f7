"""Matrix-Fisher sampling and sample-mesh uncertainty with the call surface of the reference's
utils/sampling_utils.py, executed by libhps.so (hps_mf_sample: one wavefront per (image, joint)).

Random numbers
  ``sample_on_cpu=True``  -- the reference's seed-reproducible route (run_evaluate.py:83-94): the
      proposal noise is drawn on the host from torch's global CPU generator in exactly the reference's
      order (per image, per joint: randn(8N,4) then rand(8N); one more pair per discarded round,
      utils/sampling_utils.py:51,60,128-137) and uploaded; the rejection test, compaction, quaternion ->
      rotation and U R V^T run on the device.  Same torch.manual_seed => same samples as the reference's
      CPU path, up to accept decisions that sit on an fp32 rounding tie.
  ``sample_on_cpu=False`` -- counter-based Philox4x32-10 inside the kernel, keyed by
      (seed, global image index, joint, round, proposal): independent of batch size and of how images
      are sharded over GPUs.  ``seed=None`` takes the seed from torch's global CPU generator, so
      torch.manual_seed() still controls it.
"""
import numpy as np
import torch

from . import _capi

_MAX_ROUNDS = 64
last_accepted = None        # ((B*23,) int32 device tensor, N) of the most recent Philox launch: accepted proposals of the final round per call
_pending = []               # (accepted, N, event on the launch stream) of every Philox launch since the last check_sampling()
_failed = {}                # device -> int64 scalar: failed calls of launches already folded out of _pending
_PENDING_MAX = 256
launch_events = None        # bench.py: list collecting (start, end) HIP events around every hps_mf_sample launch
unc_events = None           # bench.py: list collecting (B, N, start, end) HIP events around every hps_vertex_uncertainty launch


def _m_star(b):
    # utils/sampling_utils.py:46 / :125
    return float(np.exp(-(4 - b) / 2) * ((4 / b) ** 2))


def _launch(pose_U, pose_S, pose_V, num_samples, n_prop, b, eps=None, w=None, draw_idx=None, seed=0,
            call_offset=0, bingham_a=None, want_quat=False, acg_override=None, m_star=None, out=None, seed_dev=None):
    B, nj = pose_U.shape[:2]
    C = B * nj
    dev = pose_U.device
    if out is not None:        # caller-provided destination (e.g. the sample rows of the flattened SMPL pose buffer)
        assert out.shape == (B, num_samples, nj, 3, 3) and out.is_contiguous() and out.dtype == torch.float32
    R = out if out is not None else torch.empty(B, num_samples, nj, 3, 3, device=dev, dtype=torch.float32)
    quat = torch.empty(B, num_samples, nj, 4, device=dev, dtype=torch.float32) if want_quat else None
    accepted = torch.empty(C, device=dev, dtype=torch.int32)
    P = _capi.ptr
    ev = None
    if launch_events is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _capi.call("hps_mf_sample", P(pose_U), P(pose_S), P(pose_V), P(bingham_a) if bingham_a is not None else None,
               P(acg_override) if acg_override is not None else None,
               C, nj, num_samples, n_prop, float(b), _m_star(b) if m_star is None else float(m_star),
               P(eps) if eps is not None else None, P(w) if w is not None else None,
               _capi.iptr(draw_idx) if draw_idx is not None else None,
               int(seed) & 0xFFFFFFFFFFFFFFFF, int(call_offset),
               _capi.ptr(seed_dev, torch.int64) if seed_dev is not None else None, _MAX_ROUNDS,
               P(R), P(quat) if quat is not None else None, _capi.iptr(accepted), _capi.stream())
    if ev is not None:
        ev[1].record()
        launch_events.append(ev)
    return R, quat, accepted


def _host_stream_sampling(pose_U, pose_S, pose_V, num_samples, n_prop, b, bingham_a=None, want_quat=False, acg_override=None,
                          m_star=None, out=None):
    """Reference-order host noise; rare discarded rounds (fewer than N accepted, :68-69) shift every later
    call one draw further down the stream, exactly as the sequential reference loop would."""
    B, nj = pose_U.shape[:2]
    C = B * nj
    dev = pose_U.device
    eps_l, w_l = [], []

    def draw(n):
        for _ in range(n):
            eps_l.append(torch.randn(n_prop, 4).float())      # :51
            w_l.append(torch.rand(n_prop))                      # :60

    draw(C)
    assign = torch.arange(C, dtype=torch.int32)
    while True:
        eps = torch.stack(eps_l).to(dev)
        w = torch.stack(w_l).to(dev)
        R, quat, accepted = _launch(pose_U, pose_S, pose_V, num_samples, n_prop, b, eps=eps, w=w,
                                    draw_idx=assign.to(dev), bingham_a=bingham_a, want_quat=want_quat,
                                    acg_override=acg_override, m_star=m_star, out=out)
        fails = (accepted.cpu() < num_samples).nonzero().flatten()
        if fails.numel() == 0:
            return R, quat, accepted
        first = int(fails[0])
        assign[first:] += 1          # call `first` retries with the next pair; later calls shift along
        draw(1)


def _philox_seed(seed):
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    return seed


def bingham_sampling_for_matrix_fisher_torch(A, num_samples, Omega=None, Gaussian_std=None, b=1.5, M_star=None,
                                             oversampling_ratio=8, sample_on_cpu=False, seed=None):
    """utils/sampling_utils.py:10-71: A (4,) diagonal Bingham parameter -> (samples (N,4), accept_ratio).

    Omega / Gaussian_std / M_star default to the values derived from A and b (:42-46); caller-supplied values are used as
    given, like the reference does."""
    _capi.require_device(A, "A")
    assert A.shape == (4,)
    assert A.min() >= 0
    dev = A.device
    override = None
    if Omega is not None or Gaussian_std is not None:
        om = (1.0 + 2.0 * A / b) if Omega is None else Omega.to(dev).float().reshape(4)                 # :42-43
        sd = om ** (-0.5) if Gaussian_std is None else Gaussian_std.to(dev).float().reshape(4)        # :44-45
        override = torch.cat([om, sd]).reshape(1, 8).contiguous()
    eye = torch.eye(3, device=dev).reshape(1, 1, 3, 3).contiguous()
    S = torch.zeros(1, 1, 3, device=dev)
    a = _capi.f32c(A).reshape(1, 4)
    n_prop = num_samples * oversampling_ratio
    if sample_on_cpu:
        _, quat, accepted = _host_stream_sampling(eye, S, eye, num_samples, n_prop, b, bingham_a=a, want_quat=True,
                                                  acg_override=override, m_star=M_star)
    else:
        _, quat, accepted = _launch(eye, S, eye, num_samples, n_prop, b, seed=_philox_seed(seed), bingham_a=a,
                                    want_quat=True, acg_override=override, m_star=M_star)
    accept_ratio = int(accepted[0].item()) / num_samples * 4            # :67 (as meaningless as the original)
    return quat[0, :, 0, :], accept_ratio


def pose_matrix_fisher_sampling_torch(pose_U, pose_S, pose_V, num_samples, b=1.5, oversampling_ratio=8,
                                      sample_on_cpu=False, seed=None, image_offset=0, out=None, seed_dev=None):
    """utils/sampling_utils.py:74-143: (B,23,3,3), (B,23,3), (B,23,3,3) -> R_samples (B,N,23,3,3).

    ``image_offset``: global index of the first image of this batch (multi-GPU sharding); only used by
    the Philox route.  ``out``: optional contiguous (B,N,23,3,3) destination the kernel writes into.
    ``seed_dev``: optional (2,) int64 DEVICE tensor [seed, first call = image_offset * 23] read by the kernel at run time instead of
    ``seed`` / ``image_offset`` (GraphedInfer: a launch captured in a hipGraph must not bake the key in); such launches are not
    recorded for check_sampling() -- the caller reads ``last_accepted`` itself."""
    for t, name in ((pose_U, "pose_U"), (pose_S, "pose_S"), (pose_V, "pose_V")):
        _capi.require_device(t, name)
    U, S, V = _capi.f32c(pose_U), _capi.f32c(pose_S), _capi.f32c(pose_V)
    n_prop = num_samples * oversampling_ratio
    if sample_on_cpu:
        R, _, _ = _host_stream_sampling(U, S, V, num_samples, n_prop, b, out=out)
    else:
        nj = U.shape[1]
        R, _, accepted = _launch(U, S, V, num_samples, n_prop, b, seed=0 if seed_dev is not None else _philox_seed(seed),
                                 call_offset=image_offset * nj, out=out, seed_dev=seed_dev)
        # A call that does not reach N accepts within _MAX_ROUNDS rounds (NaN / Inf pose_S from a bad checkpoint) gets NaN
        # rotations from the kernel -- loud downstream.  The counts stay on the device (no sync on the hot path);
        # check_sampling() is the deferred test the harnesses run per batch.
        # Recorded with an event on the LAUNCH stream (several streams may sample: InferencePipeline's partitions); the counts
        # are reduced by check_sampling() / _fold_pending() on whatever stream is current then, after waiting for the event.
        # The tensors stay referenced here until then, so the caching allocator cannot hand their blocks to another stream.
        global last_accepted
        last_accepted = (accepted, num_samples)
        if seed_dev is not None:        # (an event recorded during stream capture cannot be waited for outside the graph)
            return R
        ev = torch.cuda.Event()
        ev.record()
        _pending.append((accepted, num_samples, ev))
        if len(_pending) > _PENDING_MAX:
            _fold_pending(_PENDING_MAX // 2)
    return R


def _fold_pending(keep=0):
    """Reduce the oldest recorded accept counts to one failure counter per device, ordered after their launches by event
    (stream-ordered: no host synchronisation)."""
    cur = torch.cuda.current_stream()
    while len(_pending) > keep:
        accepted, n, ev = _pending.pop(0)
        cur.wait_event(ev)
        c = (accepted < n).sum()
        _failed[accepted.device] = c if accepted.device not in _failed else _failed[accepted.device] + c


def check_sampling():
    """Raise if ANY Philox sampling launch since the previous check had a call that never reached N accepted proposals
    (synchronises).  Launches are recorded as they are made -- several batches may be in flight (InferencePipeline, on any of
    its streams) or a whole data-loader loop may run between two checks without a failure going unnoticed."""
    if not _pending and not _failed:
        return
    _fold_pending()
    bad = sum(int(c.item()) for c in _failed.values())
    _failed.clear()
    if bad:
        raise _capi.HpsError("matrix-Fisher sampling failed for %d (image, joint) calls within %d rounds "
                             "(the reference prints 'Failed sampling' and loops, utils/sampling_utils.py:68-69); "
                             "are pose_S / pose_U / pose_V finite?" % (bad, _MAX_ROUNDS))


def vertex_uncertainty(vertices_samples, out=None):
    """utils/sampling_utils.py:189-190, batched: (B,N,V,3) -> (B,V).  ``out``: optional contiguous (B,V) fp32 destination."""
    _capi.require_device(vertices_samples, "vertices_samples")
    v = _capi.f32c(vertices_samples)
    B, N, V = v.shape[:3]
    if out is not None:
        assert out.shape == (B, V) and out.is_contiguous() and out.dtype == torch.float32 and out.device == v.device
    unc = out if out is not None else torch.empty(B, V, device=v.device, dtype=torch.float32)
    ev = None
    if unc_events is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _capi.call("hps_vertex_uncertainty", _capi.ptr(v), _capi.ptr(unc), B, N, V, _capi.stream())
    if ev is not None:
        ev[1].record()
        unc_events.append((B, N, ev[0], ev[1]))
    return unc


def compute_vertex_uncertainties_by_poseMF_shapeGaussian_sampling(pose_U, pose_S, pose_V, shape_distribution,
                                                                  glob_rotmats, num_samples, smpl_model,
                                                                  use_mean_shape=False, sample_on_cpu=False,
                                                                  seed=None, image_offset=0, return_rotmats=False):
    """utils/sampling_utils.py:146-192.  The reference asserts batch size 1 (:171); here any B is accepted
    and the result equals looping the B = 1 function over the images (the flattened (B*N) SMPL call is the
    reference's own training-path precedent, train/train_poseMF_shapeGaussian_net.py:304-308).

    B == 1 returns the reference's shapes: unc (6890,), vertices (N,6890,3), joints (N,90,3);
    B > 1 returns unc (B,6890), vertices (B,N,6890,3), joints (B,N,90,3)."""
    B = pose_U.shape[0]
    assert pose_U.shape[0] == pose_S.shape[0] == pose_V.shape[0]
    R = pose_matrix_fisher_sampling_torch(pose_U, pose_S, pose_V, num_samples, b=1.5, oversampling_ratio=8,
                                          sample_on_cpu=sample_on_cpu, seed=seed, image_offset=image_offset)
    if use_mean_shape:
        shape = shape_distribution.loc[:, None, :].expand(B, num_samples, -1)                   # :178-179
    elif sample_on_cpu:
        # the reference's seed-reproducible route: Normal.sample on the host, drawn from torch's global CPU generator
        # right after the pose noise (:180-181 follows :172-177), then uploaded
        host = torch.distributions.Normal(shape_distribution.loc.cpu(), shape_distribution.scale.cpu(), validate_args=False)
        shape = host.sample([num_samples]).transpose(0, 1).to(pose_U.device)                     # (B,N,nb)
    else:
        shape = shape_distribution.sample([num_samples]).transpose(0, 1)                        # :180-181, (B,N,nb)
    glob = glob_rotmats.reshape(B, 1, 1, 3, 3).expand(B, num_samples, 1, 3, 3)
    out = smpl_model(body_pose=R.reshape(B * num_samples, -1, 3, 3),
                     global_orient=glob.reshape(B * num_samples, 1, 3, 3),
                     betas=shape.reshape(B * num_samples, -1), pose2rot=False)                  # :182-185
    V = out.vertices.shape[1]
    verts = out.vertices.view(B, num_samples, V, 3)
    joints = out.joints.view(B, num_samples, -1, 3)
    unc = vertex_uncertainty(verts)                                                             # :189-190
    if B == 1:
        res = (unc[0], verts[0], joints[0])
    else:
        res = (unc, verts, joints)
    return res + (R,) if return_rotmats else res


def joints2D_error_sorted_verts_sampling(pred_vertices_samples, pred_joints_samples, input_joints2D_heatmaps, pred_cam_wp):
    """utils/sampling_utils.py:195-233: order the (N,6890,3) vertex samples by the consistency of their projected COCO
    joints with the input 2D joints (largest per-joint pixel error, ascending).  The 180 degree flip about x the
    reference does through pytorch3d is the exact diag(1,-1,-1).  Pinned by the reference function's own ordering
    (tests/golden rank_order; tests/test_frontend.py)."""
    from .label_conversions import convert_heatmaps_to_2Djoints_coordinates_torch, ALL_JOINTS_TO_COCO_MAP
    _capi.require_device(pred_joints_samples, "pred_joints_samples")
    joints = _capi.f32c(pred_joints_samples)
    N, n_all = joints.shape[:2]
    dev = joints.device
    in_j2d, in_vis = convert_heatmaps_to_2Djoints_coordinates_torch(input_joints2D_heatmaps, eps=1e-6)
    coco = torch.tensor(ALL_JOINTS_TO_COCO_MAP, dtype=torch.int32, device=dev)
    err = torch.empty(N, device=dev, dtype=torch.float32)
    cam = _capi.f32c(pred_cam_wp).reshape(-1)[:3].contiguous()
    _capi.call("hps_sample_joints2d_error", _capi.ptr(joints), _capi.iptr(coco), n_all, _capi.ptr(in_j2d[0].contiguous()),
               _capi.ptr(in_vis[0].float().contiguous()), _capi.ptr(cam), float(input_joints2D_heatmaps.shape[-1]),
               _capi.ptr(err), N, len(ALL_JOINTS_TO_COCO_MAP), _capi.stream())
    order = torch.sort(err, descending=False, stable=True)[1]      # ties keep sample order, like the reference's CPU sort
    return pred_vertices_samples[order]

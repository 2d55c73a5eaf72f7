"""Dataset evaluation harness with the signature of the reference's
evaluate/evaluate_poseMF_shapeGaussian_net.py:19-33, for the 3D metrics (BASELINE configs[3]: 3DPW with gendered SMPL).

Per batch: Canny + heat-maps -> net -> SMPL(mode) -> SMPL(T-pose, mean shape) -> [N pose samples, N shape samples,
SMPL(samples), SMPL(T-pose samples)] -> metrics on the device.  Differences from the reference, none of which change
the numbers beyond fp32 rounding: any batch size (the reference's DataLoader uses 1); targets are posed from rotation
matrices directly instead of going through cv2.Rodrigues' log map and back (:84-92); metric sums stay on the device
and are reduced across ranks once at the end.  Under torch.distributed every rank evaluates its contiguous block of
``eval_dataset`` (sharding.shard_dataset) and the one collective sums the blocks; per-frame records are gathered in
dataset order and rank 0 writes them.  The random samples of a frame are keyed by the frame's index in the whole dataset
(Philox pose samples: seed + global index; shape noise: a host generator seeded per frame), so every metric is the same
for any world size and batch size.

``svd_mode`` selects the head's 3x3 SVD for this evaluation ("host": MKL sgesdd on the host, the reference's very routine;
"device": the in-kernel restatement of sgesdd, bit-identical to this host's MKL in U, S and V once its rounding flavour is
calibrated, _capi.svd_flavor).  Default: the model's own ``svd_mode`` -- except that the reference's seed-reproducible route
(``sample_on_cpu``, run_evaluate.py:83-94) falls back to "host" on a machine whose LAPACK neither flavour reproduces.  The silhouette / 2D-joint metrics need the pytorch3d renderer and the detector
outputs and are out of scope.

The 3DPW frames and the licensed SMPL_{MALE,FEMALE}.pkl files are external assets; any ``eval_dataset`` yielding the
item dict of data/pw3d_eval_dataset.py:72-77 works (tests use a synthetic one).
"""
import numpy as np
import torch
from torch.utils.data import DataLoader

from . import _capi, sharding
from .eval_metrics_tracker import EvalMetricsTracker
from .label_conversions import ALL_JOINTS_TO_H36M_MAP, H36M_TO_J14
from .rigid_transform_utils import batch_rodrigues, rot6d_to_rotmat
from .sampling_utils import pose_matrix_fisher_sampling_torch, check_sampling, _philox_seed


def _h36mlsp(joints):
    return joints[:, ALL_JOINTS_TO_H36M_MAP, :][:, H36M_TO_J14, :]


class _StagedCollate:
    """Collate function for the in-process DataLoader (num_workers = 0): stacks the tensor fields of a batch into two alternating
    sets of REUSED page-locked buffers instead of fresh pageable ones.  A dataset item in the reference's format is 5 MB (crop + 17
    full-resolution heat-maps); default_collate's torch.stack into new memory ran at 0.9 GB/s (page faults) and was half of the
    evaluation loop.  The loop copies the fields to the device with non_blocking=True and calls copied() -- a buffer set is reused
    only after the copies out of it have finished."""

    def __init__(self, pinned):
        self.pinned = pinned
        self.sets, self.events, self.slot = [{}, {}], [None, None], 1

    def __call__(self, items):
        from torch.utils.data import default_collate
        self.slot ^= 1
        if self.events[self.slot] is not None:
            self.events[self.slot].synchronize()
            self.events[self.slot] = None
        bufs, out = self.sets[self.slot], {}
        for k in items[0]:
            v0 = items[0][k]
            if torch.is_tensor(v0) and len(items) == 1:
                # batch size 1 (the reference's): a view, no copy
                out[k] = v0.unsqueeze(0)
            elif torch.is_tensor(v0) and v0.device.type == "cpu" and not v0.requires_grad:
                # numpy's stack (a plain memcpy per item) into the reused buffer: torch.stack costs milliseconds per call -- even
                # for a 72-float pose -- wherever torch's intra-op thread pool is larger than the cores the process may use
                # (containers with a CPU quota: 7 ms per call on the GPU boxes, five calls per batch)
                shape = (len(items),) + tuple(v0.shape)
                buf = bufs.get(k)
                if buf is None or tuple(buf.shape) != shape or buf.dtype != v0.dtype:
                    buf = bufs[k] = torch.empty(shape, dtype=v0.dtype, pin_memory=self.pinned)
                np.stack([it[k].numpy() for it in items], out=buf.numpy())
                out[k] = buf
            else:
                out[k] = default_collate([it[k] for it in items])
        return out

    def copied(self):
        """Call after the batch's host-to-device copies have been enqueued."""
        if self.pinned:
            ev = torch.cuda.Event()
            ev.record()
            self.events[self.slot] = ev


def evaluate_pose_MF_shapeGaussian_net(pose_shape_model, pose_shape_cfg, smpl_model, smpl_model_male, smpl_model_female,
                                       edge_detect_model, device, eval_dataset, metrics, save_path, num_workers=4,
                                       pin_memory=True, save_per_frame_metrics=True, num_samples_for_metrics=10,
                                       sample_on_cpu=False, batch_size=1, reduce_across_ranks=True, svd_mode=None, seed=None):
    device = torch.device(device)
    if device.type == "cuda":
        torch.cuda.set_device(device)              # libhps launches on the current device's current stream
    if svd_mode is None:
        svd_mode = "host" if (sample_on_cpu and not _capi.svd_flavor_is_exact()) else pose_shape_model.svd_mode
    if svd_mode not in ("host", "device"):
        raise ValueError("svd_mode must be 'host' or 'device'")
    frame0 = 0                                     # index in the whole dataset of this rank's first frame
    if reduce_across_ranks:
        rank, world = sharding.world_info()
        frame0 = sharding.shard_range(len(eval_dataset), rank, world)[0]
        eval_dataset = sharding.shard_dataset(eval_dataset)          # this rank's contiguous block of frames
    # one seed for the whole evaluation, from torch's global CPU generator (torch.manual_seed controls it), combined with the
    # global frame index below.  With several ranks it is RANK 0's draw that counts (broadcast): ranks whose generators were not
    # seeded alike would otherwise sample differently and the metrics would depend on the world size without anyone noticing.
    run_seed = _philox_seed(seed) if not sample_on_cpu else None
    if run_seed is not None and reduce_across_ranks:
        run_seed = sharding.broadcast_int(run_seed)
    staged = _StagedCollate(pinned=device.type == "cuda") if num_workers == 0 else None
    loader = DataLoader(eval_dataset, batch_size=batch_size, shuffle=False, drop_last=False, num_workers=num_workers,
                        pin_memory=pin_memory and staged is None, collate_fn=staged)
    tracker = EvalMetricsTracker(metrics, save_path=save_path, save_per_frame_metrics=save_per_frame_metrics)
    tracker.initialise_metric_sums()
    tracker.initialise_per_frame_metric_lists()
    want_samples = any("samples" in m for m in metrics)
    N = num_samples_for_metrics
    flip = torch.diag(torch.tensor([1.0, -1.0, -1.0], device=device))          # rotation by pi about x (:86-91)
    fnames, poses, shapes, cams = [], [], [], []
    pose_shape_model.eval()
    model_svd_mode, pose_shape_model.svd_mode = pose_shape_model.svd_mode, svd_mode
    try:
        return _evaluate_loop(**locals())
    finally:
        pose_shape_model.svd_mode = model_svd_mode


def flipped_target_rotmats(target_pose, flip=None):
    """evaluate/evaluate_poseMF_shapeGaussian_net.py:84-92: target axis-angle poses (B,72) -> the (B,24,3,3) rotation matrices the
    target SMPL call uses, the global orientation pre-multiplied by the rotation by pi about x.

    The reference goes R -> R_x(pi) R -> cv2.Rodrigues (SO(3) log, float64) -> float32 vector -> smplx's batch_rodrigues (exp)
    (utils/rigid_transform_utils.py:34-58); exp(log(.)) is the identity on SO(3), so the matrices are formed directly --
    R_x(pi) is exactly diag(1,-1,-1).  tests/test_oracle_golden.py and tests/test_gpu_evaluate.py pin this against the
    log-then-exp route with scipy's Rotation standing in for cv2 (<= 2e-6 per matrix entry, angle -> pi included)."""
    B = target_pose.shape[0]
    if flip is None:
        flip = torch.diag(torch.tensor([1.0, -1.0, -1.0], device=target_pose.device))
    R = batch_rodrigues(target_pose.reshape(-1, 3)).view(B, 24, 3, 3)
    R[:, 0] = torch.matmul(flip, R[:, 0])                                 # 'pre' multiplication
    return R


def _evaluate_loop(pose_shape_model, pose_shape_cfg, smpl_model, smpl_model_male, smpl_model_female, edge_detect_model, device,
                   loader, staged, tracker, metrics, want_samples, N, flip, fnames, poses, shapes, cams, sample_on_cpu, run_seed, frame0,
                   reduce_across_ranks, save_per_frame_metrics, save_path, **_unused):
    frame = frame0
    for batch in loader:
        with torch.no_grad():
            image = batch["image"].to(device, non_blocking=True)
            heatmaps = batch["heatmaps"].to(device, non_blocking=True)
            edges = edge_detect_model(image)                                                          # :69-71
            edge = edges["thresholded_thin_edges"] if pose_shape_cfg.DATA.EDGE_NMS else edges["thresholded_grad_magnitude"]
            proxy = torch.cat([edge, heatmaps], dim=1)
            B = proxy.shape[0]

            # ------------------ targets (:74-105) ------------------
            target_pose = batch["pose"].to(device, non_blocking=True).float()
            target_shape = batch["shape"].to(device, non_blocking=True).float()
            if staged is not None:
                staged.copied()
            genders = list(batch["gender"])
            R = flipped_target_rotmats(target_pose, flip)
            target_vertices = torch.empty(B, smpl_model.num_verts, 3, device=device)
            target_reposed = torch.empty_like(target_vertices)
            target_joints = torch.empty(B, 14, 3, device=device)
            for g, model in (("m", smpl_model_male), ("f", smpl_model_female)):
                idx = [i for i, x in enumerate(genders) if x == g]
                if not idx:
                    continue
                ii = torch.tensor(idx, device=device)
                out = model(body_pose=R[ii, 1:].contiguous(), global_orient=R[ii, :1].contiguous(), betas=target_shape[ii],
                            pose2rot=False)
                rest = model(betas=target_shape[ii], body_pose=torch.zeros(len(idx), 69, device=device),
                             global_orient=torch.zeros(len(idx), 3, device=device))
                target_vertices[ii], target_reposed[ii], target_joints[ii] = out.vertices, rest.vertices, _h36mlsp(out.joints)
            if any(x not in ("m", "f") for x in genders):
                raise ValueError("gender must be 'm' or 'f' (data/pw3d_eval_dataset.py:66)")

            # ------------------ predictions (:108-132) ------------------
            pose_F, pose_U, pose_S, pose_V, mode, shape_dist, glob, cam = pose_shape_model(proxy)
            glob_R = batch_rodrigues(glob) if glob.shape[-1] == 3 else rot6d_to_rotmat(glob)
            out_mode = smpl_model(body_pose=mode, global_orient=glob_R.unsqueeze(1), betas=shape_dist.loc, pose2rot=False)
            zeros69, zeros3 = torch.zeros(B, 69, device=device), torch.zeros(B, 3, device=device)
            out_rest = smpl_model(betas=shape_dist.loc, body_pose=zeros69, global_orient=zeros3)
            pred = {"verts": out_mode.vertices, "reposed_verts": out_rest.vertices, "joints3D": _h36mlsp(out_mode.joints)}
            target = {"verts": target_vertices, "reposed_verts": target_reposed, "joints3D": target_joints}
            base_metrics = [m for m in metrics if "samples" not in m]
            sample_metrics = [m for m in metrics if "samples" in m]
            tracker.metrics_to_track = base_metrics
            tracker.update_per_batch(pred, target, B)

            if want_samples:                                                                              # :157-179
                R_s = pose_matrix_fisher_sampling_torch(pose_U, pose_S, pose_V, N, b=1.5, oversampling_ratio=8,
                                                        sample_on_cpu=sample_on_cpu, seed=run_seed,
                                                        image_offset=frame)                             # (B,N,23,3,3)
                if sample_on_cpu:      # the reference's CPU route: shape noise from the global CPU generator too
                    cpu_dist = torch.distributions.Normal(shape_dist.loc.cpu(), shape_dist.scale.cpu())
                    shape_s = cpu_dist.rsample([N]).to(device).transpose(0, 1)
                else:                  # loc + scale * eps (rsample, :166) with eps keyed by the frame's global index
                    eps = torch.stack([torch.randn(N, shape_dist.loc.shape[1],
                                                   generator=torch.Generator().manual_seed((run_seed + frame + i) % (2 ** 63)))
                                       for i in range(B)]).to(device)
                    shape_s = shape_dist.loc[:, None] + shape_dist.scale[:, None] * eps                  # (B,N,nb)
                out_s = smpl_model(body_pose=R_s.reshape(B * N, 23, 3, 3),
                                   global_orient=glob_R[:, None, None].expand(B, N, 1, 3, 3).reshape(B * N, 1, 3, 3),
                                   betas=shape_s.reshape(B * N, -1), pose2rot=False)
                verts_s = out_s.vertices.view(B, N, -1, 3).clone()
                joints_s = _h36mlsp(out_s.joints).view(B, N, 14, 3).clone()
                verts_s[:, 0], joints_s[:, 0] = out_mode.vertices, pred["joints3D"]                      # :172-174
                rest_s = smpl_model(betas=shape_s.reshape(B * N, -1), body_pose=torch.zeros(B * N, 69, device=device),
                                    global_orient=torch.zeros(B * N, 3, device=device)).vertices.view(B, N, -1, 3).clone()
                rest_s[:, 0] = out_rest.vertices                                                          # :179
                tracker.metrics_to_track = sample_metrics
                for i in range(B):     # the samples_min metrics are per frame (metrics/eval_metrics_tracker.py:181)
                    tracker.update_per_batch({"verts_samples": verts_s[i], "reposed_verts_samples": rest_s[i],
                                              "joints3D_samples": joints_s[i]},
                                             {k: v[i:i + 1] for k, v in target.items()}, 1)
                tracker.total_samples -= B                                       # already counted by the base update
            tracker.metrics_to_track = list(metrics)
            if save_per_frame_metrics:
                fnames.extend(list(batch["fname"]))
                poses.append(torch.cat([glob_R[:, None], mode], dim=1).cpu().numpy())
                shapes.append(shape_dist.loc.cpu().numpy())
                cams.append(cam.cpu().numpy())
            frame += B
    check_sampling()              # every launch since the last check is recorded (sampling_utils._pending): no batch is missed
    if reduce_across_ranks:
        tracker.reduce_across_ranks()
    final = tracker.compute_final_metrics(verbose=sharding.world_info()[0] == 0)
    if save_per_frame_metrics and save_path is not None:
        import os
        cat = lambda parts, tail: np.concatenate(parts, axis=0) if parts else np.zeros((0,) + tail, np.float32)
        records = {"fname_per_frame.npy": np.array(fnames), "pose_per_frame.npy": cat(poses, (24, 3, 3)),
                   "shape_per_frame.npy": cat(shapes, (pose_shape_model.num_shape_params,)), "cam_per_frame.npy": cat(cams, (3,))}
        if reduce_across_ranks:
            records = {k: sharding.gather_per_frame(v) for k, v in records.items()}
        if sharding.world_info()[0] == 0:
            for name, arr in records.items():
                np.save(os.path.join(save_path, name), arr)
    return final

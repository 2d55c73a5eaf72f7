"""``EvalMetricsTracker`` with the interface of the reference's metrics/eval_metrics_tracker.py for the 3D metrics
(PVE / PVE-SC / PVE-PA / PVE-T / PVE-T-SC / MPJPE / MPJPE-SC / MPJPE-PA and their ``*_samples_min`` forms) and
joints2D-L2E, computed on the device (hps_pointset_errors) from device tensors.  Sums stay on the device as float64
scalars; nothing is synchronised with the host until ``compute_final_metrics``.  ``reduce_across_ranks`` is the one
collective of a multi-GPU evaluation.

Not implemented (out of scope, they need the pytorch3d renderer): silhouette-IOU, silhouettesamples-IOU.
"""
import os

import numpy as np
import torch

from . import eval_utils, sharding

_MODES = {"": eval_utils.MODE_RAW, "-SC": eval_utils.MODE_SC, "-PA": eval_utils.MODE_PA}
# metric name -> (pred key, target key, alignment mode)
_POINT_METRICS = {}
for _sfx, _mode in _MODES.items():
    _POINT_METRICS["PVE" + _sfx] = ("verts", "verts", _mode)
    _POINT_METRICS["MPJPE" + _sfx] = ("joints3D", "joints3D", _mode)
_POINT_METRICS["PVE-T"] = ("reposed_verts", "reposed_verts", eval_utils.MODE_RAW)
_POINT_METRICS["PVE-T-SC"] = ("reposed_verts", "reposed_verts", eval_utils.MODE_SC)


class EvalMetricsTracker:
    def __init__(self, metrics_to_track, img_wh=None, save_path=None, save_per_frame_metrics=False):
        for m in metrics_to_track:
            if "silhouette" in m:
                raise NotImplementedError("%s needs the pytorch3d renderer, which is outside the hot path" % m)
        self.metrics_to_track = list(metrics_to_track)
        self.img_wh = img_wh
        self.metric_sums = None
        self.total_samples = 0
        self.save_per_frame_metrics = save_per_frame_metrics
        self.save_path = save_path

    def initialise_metric_sums(self):
        self.metric_sums = {}
        for m in self.metrics_to_track:
            if m == "joints2Dsamples-L2E":
                self.metric_sums["num_vis_joints2Dsamples"] = 0.0
            self.metric_sums[m] = 0.0

    def initialise_per_frame_metric_lists(self):
        self.per_frame_metrics = {m: [] for m in self.metrics_to_track}

    # ------------------------------------------------------------------------------------------
    def update_per_batch(self, pred_dict, target_dict, num_input_samples, return_transformed_points=False,
                         return_per_frame_metrics=False):
        """metrics/eval_metrics_tracker.py:71-330 for device tensors: pred_dict / target_dict hold 'verts' (B,6890,3),
        'reposed_verts', 'joints3D' (B,14,3) and, for the samples_min metrics, 'verts_samples' (N,6890,3) etc."""
        self.total_samples += num_input_samples
        transformed = {} if return_transformed_points else None
        per_frame = {} if return_per_frame_metrics else None
        names = {"PVE-SC": "pred_vertices_sc", "PVE-PA": "pred_vertices_pa", "PVE-T-SC": "pred_reposed_vertices_sc",
                 "MPJPE-SC": "pred_joints3D_h36mlsp_sc", "MPJPE-PA": "pred_joints3D_h36mlsp_pa"}
        for m in self.metrics_to_track:
            if m in _POINT_METRICS:
                pk, tk, mode = _POINT_METRICS[m]
                want_pts = return_transformed_points and m in names
                res = eval_utils.pointset_errors(pred_dict[pk], target_dict[tk], mode, 1, want_pts)
                err = res[0] if want_pts else res
                if want_pts:
                    transformed[names[m]] = res[1]
                self.metric_sums[m] = self.metric_sums[m] + err.sum()
                mean = err / pred_dict[pk].shape[1]
                self.per_frame_metrics[m].append(mean)
                if return_per_frame_metrics:
                    per_frame[m] = mean
            elif m.endswith("_samples_min") and m[:-len("_samples_min")] in _POINT_METRICS:
                assert num_input_samples == 1, "Batch size must be 1 for min samples metrics!"
                pk, tk, mode = _POINT_METRICS[m[:-len("_samples_min")]]
                samples = pred_dict[pk + "_samples"]                            # (N,P,3) against one target
                err = eval_utils.pointset_errors(samples, target_dict[tk][:1], mode, samples.shape[0])
                best = err.min()                                                # argmin of the per-sample mean == min of the sums
                self.metric_sums[m] = self.metric_sums[m] + best
                self.per_frame_metrics[m].append((best / samples.shape[1]).reshape(1))
            elif m == "joints2D-L2E":
                d = torch.linalg.norm(pred_dict["joints2D"].double() - target_dict["joints2D"].to(pred_dict["joints2D"].device).double(), dim=-1)
                self.metric_sums[m] = self.metric_sums[m] + d.sum()
                self.per_frame_metrics[m].append(d.mean(dim=-1))
                if return_per_frame_metrics:
                    per_frame[m] = d.mean(dim=-1)
            elif m == "joints2Dsamples-L2E":
                pred = pred_dict["joints2Dsamples"].double()                    # (B,N,17,2)
                tgt = target_dict["joints2D"].to(pred.device).double()[:, None].expand_as(pred)
                d = torch.linalg.norm(pred - tgt, dim=-1)                       # (B,N,17)
                if "joints2D_vis" in target_dict:
                    vis = target_dict["joints2D_vis"].to(pred.device)[:, None].expand(d.shape)
                    self.metric_sums[m] = self.metric_sums[m] + (d * vis).sum()
                    self.metric_sums["num_vis_joints2Dsamples"] = self.metric_sums["num_vis_joints2Dsamples"] + vis.sum()
                else:
                    self.metric_sums[m] = self.metric_sums[m] + d.sum()
                    self.metric_sums["num_vis_joints2Dsamples"] = self.metric_sums["num_vis_joints2Dsamples"] + d.numel()
            else:
                raise KeyError("unknown metric %r" % m)
        return transformed, per_frame

    # ------------------------------------------------------------------------------------------
    def reduce_across_ranks(self):
        """One all-gather of [total_samples, metric sums...]; afterwards every rank holds the global sums (reduced in
        rank order).  No-op without an initialised process group."""
        keys = sorted(self.metric_sums)
        dev = next((v.device for v in self.metric_sums.values() if torch.is_tensor(v)), torch.device("cpu"))
        vec = torch.stack([torch.as_tensor(float(self.total_samples), dtype=torch.float64, device=dev)] +
                          [torch.as_tensor(self.metric_sums[k], dtype=torch.float64, device=dev).reshape(()) for k in keys])
        _, total = sharding.gather_metric_sums(vec)
        self.total_samples = int(round(float(total[0])))
        for i, k in enumerate(keys):
            self.metric_sums[k] = total[1 + i]
        # per-frame vectors of every rank in rank order = dataset order (ranks own contiguous blocks: sharding.shard_dataset)
        if self.save_per_frame_metrics and getattr(self, "per_frame_metrics", None) is not None and sharding.world_info()[1] > 1:
            for m, chunks in self.per_frame_metrics.items():
                local = torch.cat([c.reshape(-1) for c in chunks]).cpu().numpy() if chunks else np.zeros(0)
                self.per_frame_metrics[m] = [torch.from_numpy(sharding.gather_per_frame(local))]

    def compute_final_metrics(self, verbose=True):
        """metrics/eval_metrics_tracker.py:332-368: per-point means (metres / pixels); prints millimetres for 3D metrics."""
        final = {}
        for m in self.metrics_to_track:
            mult = 1.0
            if m == "joints2Dsamples-L2E":
                final[m] = float(self.metric_sums[m]) / float(self.metric_sums["num_vis_joints2Dsamples"])
            else:
                if "PVE" in m:
                    num_per_sample, mult = 6890, 1000.0
                elif "MPJPE" in m:
                    num_per_sample, mult = 14, 1000.0
                else:
                    num_per_sample = 17
                final[m] = float(self.metric_sums[m]) / (self.total_samples * num_per_sample)
            if verbose:
                print(m, "{:.2f}".format(final[m] * mult))
        if self.save_per_frame_metrics and self.save_path is not None and sharding.world_info()[0] == 0:
            for m in self.metrics_to_track:
                if "samples" not in m:
                    np.save(os.path.join(self.save_path, m + "_per_frame.npy"),
                            torch.cat(self.per_frame_metrics[m]).cpu().numpy())
        return final

"""Evaluation metrics.  The golden values come from the reference's EvalMetricsTracker run on
tests/metric_scenario.py (tests/golden/make_golden.py).  Stated tolerance: 1e-5 relative on every metric sum (the
reference reduces in fp32 numpy, the kernels in fp64)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from metric_scenario import METRICS, make_frames


def test_oracle_metric_sums_match_reference_tracker(golden):
    sums = {m: 0.0 for m in METRICS}
    for pred, target in make_frames():
        for k, v in O.metric_sums(pred, target, METRICS).items():
            sums[k] += v
    got = torch.tensor([sums[m] for m in METRICS], dtype=torch.float64)
    assert float(((got - golden["metrics_sums"]).abs() / golden["metrics_sums"].abs()).max()) <= 1e-6


def test_procrustes_recovers_a_similarity_transform():
    rs = np.random.RandomState(1)
    t = rs.randn(50, 3)
    q, _ = np.linalg.qr(rs.randn(3, 3))
    q *= np.sign(np.linalg.det(q))
    p = 0.7 * t.dot(q.T) + np.array([0.3, -0.2, 0.9])
    assert np.abs(O.compute_similarity_transform(p, t) - t).max() <= 1e-9


@pytest.mark.gpu
def test_device_tracker_matches_reference_tracker(dev, golden):
    from hierarchicalprobabilistic3dhuman_amd.eval_metrics_tracker import EvalMetricsTracker
    tr = EvalMetricsTracker(METRICS)
    tr.initialise_metric_sums()
    tr.initialise_per_frame_metric_lists()
    for pred, target in make_frames():
        tr.update_per_batch({k: torch.from_numpy(v).to(dev) for k, v in pred.items()},
                            {k: torch.from_numpy(v).to(dev) for k, v in target.items()}, 1)
    sums = torch.stack([torch.as_tensor(tr.metric_sums[m]).cpu().double() for m in METRICS])
    rel = ((sums - golden["metrics_sums"]).abs() / golden["metrics_sums"].abs())
    assert float(rel.max()) <= 1e-5, dict(zip(METRICS, rel.tolist()))
    final = tr.compute_final_metrics(verbose=False)
    got = torch.tensor([final[m] for m in METRICS], dtype=torch.float64)
    assert float(((got - golden["metrics_final"]).abs() / golden["metrics_final"].abs()).max()) <= 1e-5
    per_frame = torch.cat(tr.per_frame_metrics["PVE-PA"]).cpu()
    assert float((per_frame - golden["metrics_pve_pa_per_frame"]).abs().max()) <= 1e-6


@pytest.mark.gpu
def test_alignment_functions_and_batched_sets(dev):
    from hierarchicalprobabilistic3dhuman_amd import eval_utils
    frames = make_frames(num_frames=2, num_samples=4, seed=3)
    P = np.concatenate([f[0]["verts_samples"] for f in frames])          # (8,6890,3): 4 samples per frame
    T = np.concatenate([f[1]["verts"] for f in frames])                  # (2,6890,3)
    Tt = np.repeat(T, 4, axis=0)
    pa = eval_utils.procrustes_analysis_batch(torch.from_numpy(P).to(dev), torch.from_numpy(Tt).to(dev))
    sc = eval_utils.scale_and_translation_transform_batch(torch.from_numpy(P).to(dev), torch.from_numpy(Tt).to(dev))
    assert np.abs(pa.cpu().numpy() - O.procrustes_analysis_batch(P, Tt)).max() <= 2e-5
    assert np.abs(sc.cpu().numpy() - O.scale_and_translation_transform_batch(P, Tt)).max() <= 2e-5
    # group > 1: predictions s compare with target s // group
    err = eval_utils.pointset_errors(torch.from_numpy(P).to(dev), torch.from_numpy(T).to(dev), eval_utils.MODE_PA, group=4)
    want = np.linalg.norm(O.procrustes_analysis_batch(P, Tt) - Tt, axis=-1).sum(-1)
    assert np.abs(err.cpu().numpy() - want).max() <= 1e-5 * want.max()
    # a reflected copy must not be matched by an improper rotation (det fix, eval_utils.py:41-42)
    refl = P[:1].copy()
    refl[..., 0] *= -1
    e_pa = float(eval_utils.pointset_errors(torch.from_numpy(refl).to(dev), torch.from_numpy(P[:1]).to(dev), eval_utils.MODE_PA))
    assert e_pa > 1.0
    w = np.linalg.norm(O.procrustes_analysis_batch(refl, P[:1]) - P[:1], axis=-1).sum()
    assert abs(e_pa - w) <= 1e-4 * w


def test_silhouette_metrics_are_refused():
    from hierarchicalprobabilistic3dhuman_amd.eval_metrics_tracker import EvalMetricsTracker
    with pytest.raises(NotImplementedError):
        EvalMetricsTracker(["PVE", "silhouette-IOU"])

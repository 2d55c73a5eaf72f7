"""GPU: the dataset-evaluation harness (BASELINE configs[3] path: gendered SMPL targets, mode / reposed / sample
meshes, metrics on the device) against the oracle's restatement of evaluate/evaluate_poseMF_shapeGaussian_net.py on a
synthetic three-frame dataset (3DPW and the licensed gendered SMPL files are external assets; synthetic 'male' /
'female' SMPL-shaped models with different seeds stand in).  sample_on_cpu=True so both sides draw the same samples.

Stated tolerance: 1e-4 relative on every final metric."""
import pytest
import torch

from oracle import ref_cpu as O
from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data
from metric_scenario import METRICS

pytestmark = pytest.mark.gpu


class _SyntheticEvalDataset(torch.utils.data.Dataset):
    """Items with the keys of data/pw3d_eval_dataset.py:72-77."""

    def __init__(self, n, wh=64):
        g = torch.Generator().manual_seed(5)
        self.items = []
        for i in range(n):
            img = torch.nn.functional.interpolate(torch.rand(1, 3, wh // 4, wh // 4, generator=g), size=(wh, wh), mode="bilinear",
                                                  align_corners=False)[0] + 0.05 * torch.rand(3, wh, wh, generator=g)
            joints = torch.rand(1, 17, 2, generator=g) * wh
            heat = O.joints2d_to_gaussian_heatmaps(joints.round(), wh, 4.0)[0]
            self.items.append({"image": img, "heatmaps": heat, "pose": torch.randn(72, generator=g) * 0.4,
                               "shape": torch.randn(10, generator=g), "fname": "frame_%03d.png" % i,
                               "gender": "m" if i % 2 == 0 else "f"})

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_evaluate_matches_oracle(dev, net_gpu, net_cpu, smpl_gpu, smpl_assets, tmp_path):
    from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
    from hierarchicalprobabilistic3dhuman_amd.evaluate_poseMF_shapeGaussian_net import evaluate_pose_MF_shapeGaussian_net
    from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
    extra = smpl_assets[1]
    models = {"m": smpl_data.synthetic_smpl_model(1), "f": smpl_data.synthetic_smpl_model(2)}
    gpu_models = {k: SMPL(v, gender={"m": "male", "f": "female"}[k]).to(dev) for k, v in models.items()}
    cpu_models = {k: O.SMPLParams(v, extra, configs.SMPLX_EXTRA_VERTEX_IDS) for k, v in models.items()}
    cfg = configs.get_cfg_defaults()
    ds = _SyntheticEvalDataset(3, wh=256)
    N = 4
    frames = [{"image": it["image"][None], "heatmaps": it["heatmaps"][None], "pose": it["pose"][None],
               "shape": it["shape"][None], "gender": it["gender"]} for it in ds.items]
    torch.manual_seed(21)
    want = O.evaluate_frames(net_cpu[1], smpl_assets[2], cpu_models, configs.SMPL_PARENTS, frames, METRICS, N)
    det = CannyEdgeDetector(cfg.DATA.EDGE_NMS, cfg.DATA.EDGE_GAUSSIAN_STD, cfg.DATA.EDGE_GAUSSIAN_SIZE, cfg.DATA.EDGE_THRESHOLD).to(dev)
    torch.manual_seed(21)
    got = evaluate_pose_MF_shapeGaussian_net(net_gpu, cfg, smpl_gpu, gpu_models["m"], gpu_models["f"], det, dev, ds, METRICS,
                                             str(tmp_path), num_workers=0, pin_memory=False, save_per_frame_metrics=True,
                                             num_samples_for_metrics=N, sample_on_cpu=True, batch_size=1)
    for m in METRICS:
        assert abs(got[m] - want[m]) <= 1e-4 * abs(want[m]), (m, got[m], want[m])
    assert (tmp_path / "pose_per_frame.npy").exists() and (tmp_path / "PVE_per_frame.npy").exists()
    # batched evaluation (3 frames in one batch) gives the same non-sample metrics
    base = [m for m in METRICS if "samples" not in m]
    got_b = evaluate_pose_MF_shapeGaussian_net(net_gpu, cfg, smpl_gpu, gpu_models["m"], gpu_models["f"], det, dev, ds, base,
                                               None, num_workers=0, pin_memory=False, save_per_frame_metrics=False,
                                               batch_size=3)
    for m in base:
        assert abs(got_b[m] - want[m]) <= 1e-4 * abs(want[m]), m

"""Throughput of the dataset-evaluation harness (BASELINE configs[3] path) on the GPU box.

    python tools/evaluate_time.py [frames] [batch] [samples] [loader workers]

A synthetic 3DPW-like dataset (tests/test_gpu_evaluate.py: 256 x 256 crops, 17 heat-maps, SMPL pose / shape targets, genders),
every metric of the reference's tracker, gendered target meshes, metrics on the device.  Prints frames/s for the whole loop
(front end + network + SMPL + metrics, host included) -- there is no reference number to compare with (the reference evaluates
3DPW frame by frame with batch_size 1 on one GPU); the CPU oracle's time per frame is printed beside it on a few frames.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.evaluate_poseMF_shapeGaussian_net import evaluate_pose_MF_shapeGaussian_net  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL  # noqa: E402
from metric_scenario import METRICS  # noqa: E402
from test_gpu_evaluate import _SyntheticEvalDataset  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    samples = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda:0")
    cfg = configs.get_cfg_defaults()
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, cfg).eval().to(dev)
    smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
    male = SMPL(smpl_data.synthetic_smpl_model(1), gender="male").to(dev)
    female = SMPL(smpl_data.synthetic_smpl_model(2), gender="female").to(dev)
    det = CannyEdgeDetector(cfg.DATA.EDGE_NMS, cfg.DATA.EDGE_GAUSSIAN_STD, cfg.DATA.EDGE_GAUSSIAN_SIZE, cfg.DATA.EDGE_THRESHOLD).to(dev)
    small = _SyntheticEvalDataset(16, wh=256)
    ds = torch.utils.data.ConcatDataset([small] * (frames // 16))
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    kw = dict(num_workers=workers, pin_memory=workers > 0, save_per_frame_metrics=False, num_samples_for_metrics=samples,
              sample_on_cpu=False, batch_size=batch)
    evaluate_pose_MF_shapeGaussian_net(net, cfg, smpl, male, female, det, dev, small, METRICS, None, **kw)      # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    final = evaluate_pose_MF_shapeGaussian_net(net, cfg, smpl, male, female, det, dev, ds, METRICS, None, **kw)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("evaluate harness: %d frames, batch %d, %d samples per frame, %d metrics, %d loader workers: %.3f s = %.0f frames/s (%.2f ms per frame)"
          % (len(ds), batch, samples, len(METRICS), workers, dt, len(ds) / dt, 1e3 * dt / len(ds)))
    print("   ", ", ".join("%s %.4g" % (k, v) for k, v in list(final.items())[:4]), "...")


if __name__ == "__main__":
    main()

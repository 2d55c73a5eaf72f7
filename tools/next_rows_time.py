"""Timing of the widened rows' kernels against their algorithmic HBM bytes, on the GPU box.

    python tools/next_rows_time.py  ->  one line per kernel: ms (median of 9 rounds of 10 launches), algorithmic MB, GB/s, fraction of 8 TB/s

Sizes: the front end at the bench batch (64 crops of 3 x 256 x 256, 17 heat-maps of 64 x 64 per crop as HRNet-w48 emits them,
proxy representations of 18 x 256 x 256); the metrics at one evaluation batch of 64 frames x 10 samples of 6890 vertices
(evaluate_poseMF_shapeGaussian_net.py: PVE, PVE-SC, PVE-PA); the sample ranking at N = 100 samples of one frame.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import eval_utils, label_conversions  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector  # noqa: E402

HBM = 8000.0   # GB/s, MI355X_MICROARCH.md
dev = torch.device("cuda:0")


def timed(fn):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def report(name, ms, nbytes, note=""):
    print("%-46s %8.4f ms (min %.4f max %.4f)  %8.1f MB algorithmic  %7.1f GB/s = %.2f of HBM  %s" %
          (name, ms[0], ms[1], ms[2], nbytes / 1e6, nbytes / ms[0] / 1e6, nbytes / ms[0] / 1e6 / HBM, note))


def main():
    g = torch.Generator().manual_seed(0)
    B, H = 64, 256
    img = torch.rand(B, 3, H, H, generator=g).to(dev)
    canny = CannyEdgeDetector(non_max_suppression=True, gaussian_filter_std=1.0, gaussian_filter_size=5, threshold=0.0).to(dev)
    # reads the image once; writes blurred (3 planes) + magnitude, orientation, thresholded magnitude, thin, thresholded thin
    report("hps_canny_edges (64 x 3 x 256 x 256, NMS)", timed(lambda: canny(img)), B * H * H * 4.0 * (3 + 3 + 5))
    # the predict front end's form: the edge map alone, written into channel 0 of the proxy representation (3 planes in, 1 out),
    # then the 17 heat-map channels in place
    proxy = torch.empty(B, 18, H, H, device=dev)
    report("hps_canny_edge_map (64 x 3 x 256 x 256 -> ch 0)", timed(lambda: canny.edge_map_into(img, proxy)), B * H * H * 4.0 * (3 + 1))
    j2 = (torch.rand(B, 17, 2, generator=g) * H).to(dev)
    v2 = (torch.rand(B, 17, generator=g) > 0.2).to(dev)
    report("hps_proxy_rep in place (64 x 17 x 256 x 256)", timed(lambda: label_conversions.make_proxy_representation(None, j2, v2, H, 4.0, out=proxy)),
           B * H * H * 4.0 * 17)
    edge = torch.rand(B, 1, H, H, generator=g).to(dev)
    j2d = (torch.rand(B, 17, 2, generator=g) * H).to(dev)
    vis = (torch.rand(B, 17, generator=g) > 0.2).to(dev)
    report("hps_proxy_rep (64 x 18 x 256 x 256)", timed(lambda: label_conversions.make_proxy_representation(edge, j2d, vis, H, 4.0)),
           B * H * H * 4.0 * (1 + 18))
    heat = torch.rand(B, 17, 64, 64, generator=g).to(dev)
    report("hps_heatmaps_to_joints2d (64 x 17 x 64 x 64)", timed(lambda: label_conversions.convert_heatmaps_to_2Djoints_coordinates_torch(heat)),
           B * 17 * 64 * 64 * 4.0)
    S, P = 64 * 10, 6890
    pred = torch.randn(S, P, 3, generator=g).to(dev)
    tgt = torch.randn(64, P, 3, generator=g).to(dev)
    for mode, name in ((eval_utils.MODE_RAW, "raw"), (eval_utils.MODE_SC, "scale + translation"), (eval_utils.MODE_PA, "Procrustes")):
        # algorithmic: predictions and (grouped) targets read once; the transformed modes need the statistics first, i.e. two sweeps
        sweeps = 1 if mode == eval_utils.MODE_RAW else 2
        report("hps_pointset_errors %s (640 x 6890 x 3)" % name, timed(lambda: eval_utils.pointset_errors(pred, tgt, mode, 10)),
               sweeps * (S * P * 12.0 + 64 * P * 12.0), "(%d sweep%s)" % (sweeps, "" if sweeps == 1 else "s"))


if __name__ == "__main__":
    main()

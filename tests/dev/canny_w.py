"""Times hps_canny_edges on the same number of pixels as 64 x 3 x 256 x 256 at different widths (W = 256: one wave per SIMD with full
lanes; W = 128: twice the waves, half of each wave's lanes idle) -- does a second wave per SIMD pay?  usage: canny_w.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector

dev = torch.device("cuda:0")
det = CannyEdgeDetector(True, 1.0, 5, 0.0).to(dev)
for (B, H, W) in [(64, 256, 256), (128, 256, 128), (64, 512, 128), (256, 256, 64)]:
    img = torch.rand(B, 3, H, W, device=dev)
    for _ in range(3):
        det(img)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); det(img); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("B %d H %d W %d: median %.4f ms (min %.4f)" % (B, H, W, ts[len(ts) // 2], ts[0]))

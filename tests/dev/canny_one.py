"""Runs the Canny detector (all outputs, or the edge map alone with `edge`) on B crops of 3 x 256 x 256 a few times, for
rocprofv3 --pmc / --kernel-trace (tools/canny_pmc.sh).  usage: canny_one.py [B] [full|edge] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "full"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
g = torch.Generator().manual_seed(3)
img = torch.nn.functional.interpolate(torch.rand(B, 3, 64, 64, generator=g), size=(256, 256), mode="bilinear",
                                      align_corners=False).to(dev)
det = CannyEdgeDetector(True, 1.0, 5, 0.0).to(dev)
proxy = torch.empty(B, 18, 256, 256, device=dev)
for _ in range(iters):
    if mode == "edge":
        det.edge_map_into(img, proxy)
    else:
        det(img)
torch.cuda.synchronize()

"""Runs one ResNet-18 convolution shape under the v3 kernel variants named on the command line (for rocprofv3 --pmc)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN

dev = torch.device("cuda:0")
H, Cin, Cout, k, st, pd = 32, 128, 128, 3, 1, 1
conv = torch.nn.Conv2d(Cin, Cout, k, st, pd, bias=False).to(dev)
bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
cb = _ConvBN(conv, bn)
x = torch.relu(torch.randn(64, H, H, Cin, device=dev))
for v in [int(a) for a in sys.argv[1:]]:
    cb.kernel, cb.variant = "v3", v
    for _ in range(6):
        cb(x)
    torch.cuda.synchronize()

"""Runs one ResNet-18 convolution shape (128->128 3x3 at 32x32, B=64) under the kernels named on the command line, for
rocprofv3 --pmc (tools/conv_pmc.sh):  v3:<variant> = conv.hip LDS-DMA kernel (21 = no DMA, 31 = no MFMA ablations),
pad:<variant> = conv_pad.hip halo-padded kernel."""
import os
import sys

import torch
import os as _os, sys as _sys
_sys.path[:0] = [_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))]          # tests/ (devlib) and the repository root
import devlib
devlib.enable_plain_call()          # cb(x) = the un-padded kernel generation of libhps_dev.so (tests/devlib.py)

import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN

dev = torch.device("cuda:0")
H, Cin, Cout, k, st, pd = 32, 128, 128, 3, 1, 1
conv = torch.nn.Conv2d(Cin, Cout, k, st, pd, bias=False).to(dev)
bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
cb = _ConvBN(conv, bn)
x = torch.relu(torch.randn(64, H, H, Cin, device=dev))
xp = F.pad(x, (0, 0, 1, 1, 1, 1)).contiguous()
out = torch.zeros(64, H + 2, H + 2, Cout, device=dev)
for a in sys.argv[1:]:
    kind, v = a.split(":")
    cb.kernel, cb.variant = "v3", int(v)
    for _ in range(6):
        if kind == "pad":
            cb.padded(xp, 1, out, 1)
        else:
            cb(x)
    torch.cuda.synchronize()

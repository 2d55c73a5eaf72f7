"""Runs one Winograd convolution shape (B = 64) a few times for rocprofv3 --pmc.  usage: wino_one.py H C [ablate]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd import _capi
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN

dev = torch.device("cuda:0")
H, C = int(sys.argv[1]), int(sys.argv[2])
ab = int(sys.argv[3]) if len(sys.argv) > 3 else 0
conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
bn = torch.nn.BatchNorm2d(C).eval().to(dev)
cb = _ConvBN(conv, bn)
x = F.pad(torch.relu(torch.randn(64, H, H, C, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
out = torch.zeros(64, H + 2, H + 2, C, device=dev)
P = _capi.ptr
with _capi.dev_library():
    for _ in range(6):
        _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), None, P(out), 64, H, H, 1, C, C, 1, 1, None, ab,
                   _capi.stream())
torch.cuda.synchronize()

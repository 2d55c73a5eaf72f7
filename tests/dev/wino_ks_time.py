"""A/B on one box: K slices of the Winograd layer4 geometry (8 x 8 maps, 512 -> 512, B = 64): 4 (product) against 2 and 8 and 1,
interleaved rounds, median HIP-event time of one layer (kernel + slice-sum pass).  usage: wino_ks_time.py [B]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd import _capi
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, C = 8, 512
conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
bn = torch.nn.BatchNorm2d(C).eval().to(dev)
cb = _ConvBN(conv, bn)
x = F.pad(torch.relu(torch.randn(B, H, H, C, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
res = F.pad(torch.randn(B, H, H, C, device=dev), (0, 0, 1, 1, 1, 1)).contiguous()
out = torch.zeros(B, H + 2, H + 2, C, device=dev)
ws = torch.empty(8 * B * 64 * C, device=dev)
P = _capi.ptr
times, outs = {}, {}
with _capi.dev_library():
    for rnd in range(10):
        for ks in (4, 2, 8, 1):
            _capi.call("hps_dev_wino_quad_ksplit", ks)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), P(res), P(out), B, H, H, 1, C, C, 1, 1, P(ws), 0,
                           _capi.stream())
            e1.record()
            torch.cuda.synchronize()
            if rnd >= 2:
                times.setdefault(ks, []).append(e0.elapsed_time(e1) * 100.0)
            outs[ks] = out.clone()
    _capi.call("hps_dev_wino_quad_ksplit", 0)
for ks, v in times.items():
    v.sort()
    print("B = %d, %d K slices: median %.1f us per layer (min %.1f max %.1f); max |y - y(4 slices)| = %.2e" % (
        B, ks, v[len(v) // 2], v[0], v[-1], float((outs[ks] - outs[4]).abs().max())))

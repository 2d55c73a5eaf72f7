"""Race hunt: many repetitions of every ResNet-18 layer shape on the halo-padded kernel against the conv.hip kernel
(bit-identical by construction), with fresh random data each time, plus whole-encoder determinism."""
import os
import sys

import torch
import os as _os, sys as _sys
_sys.path[:0] = [_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))]          # tests/ (devlib) and the repository root
import devlib
devlib.enable_plain_call()          # cb(x) = the un-padded kernel generation of libhps_dev.so (tests/devlib.py)

import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN, resnet18

dev = torch.device("cuda:0")
shapes = [(64, 64, 64, 3, 1, 1), (64, 64, 128, 3, 2, 1), (64, 64, 128, 1, 2, 0), (32, 128, 128, 3, 1, 1), (32, 128, 256, 3, 2, 1),
          (16, 256, 256, 3, 1, 1), (16, 256, 512, 3, 2, 1), (8, 512, 512, 3, 1, 1)]
bad = 0
for (H, Cin, Cout, k, st, pd) in shapes:
    conv = torch.nn.Conv2d(Cin, Cout, k, st, pd, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(Cout).eval().to(dev)
    cb = _ConvBN(conv, bn)
    cb.use_winograd = False          # the bit-for-bit claim is about the direct kernel (the Winograd layers have their own tests)
    Ho = (H + 2 * pd - k) // st + 1
    out = torch.zeros(64, Ho + 2, Ho + 2, Cout, device=dev)
    for it in range(20):
        x = torch.relu(torch.randn(64, H, H, Cin, device=dev))
        res = torch.randn(64, Ho, Ho, Cout, device=dev)
        want = cb(x, residual=res)
        cb.padded(F.pad(x, (0, 0, 1, 1, 1, 1)).contiguous(), 1, out, 1, residual=F.pad(res, (0, 0, 1, 1, 1, 1)).contiguous())
        if not torch.equal(out[:, 1:-1, 1:-1], want):
            bad += 1
            print("MISMATCH", (H, Cin, Cout, k, st), it, float((out[:, 1:-1, 1:-1] - want).abs().max()))
    print("shape", (H, Cin, Cout, k, st), "ok")
torch.manual_seed(0)
enc = resnet18(18).eval().to(dev)
x = torch.rand(64, 18, 256, 256, device=dev)
ref = enc(x).clone()
for it in range(30):
    if not torch.equal(enc(x), ref):
        bad += 1
        print("encoder run", it, "differs")
print("stress done, mismatches:", bad)

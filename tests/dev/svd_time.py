"""Device time of hps_svd3_packed (the gesdd-faithful 3x3 SVD, one lane per matrix) for a few counts.  python tests/dev/svd_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hierarchicalprobabilistic3dhuman_amd import _capi

dev = torch.device("cuda:0")
torch.manual_seed(0)
for n, kind in ((4, "near I"), (64, "near I"), (192, "near I"), (192, "random"), (192, "identical"), (65536, "near I")):
    if kind == "near I":
        f = torch.eye(3, device=dev).expand(n, 3, 3) + 0.3 * torch.randn(n, 3, 3, device=dev)
    elif kind == "random":
        f = 5 * torch.randn(n, 3, 3, device=dev)
    else:
        f = (torch.eye(3, device=dev) + 0.3 * torch.randn(3, 3, device=dev)).expand(n, 3, 3)
    f = f.contiguous()
    usv = torch.empty(n, 21, device=dev)
    s = _capi.stream()
    for _ in range(3):
        _capi.call("hps_svd3_packed", _capi.ptr(f), _capi.ptr(usv), n, _capi.svd_flavor(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        _capi.call("hps_svd3_packed", _capi.ptr(f), _capi.ptr(usv), n, _capi.svd_flavor(), s)
    e1.record()
    torch.cuda.synchronize()
    print("svd3 n=%6d %-9s %.1f us per launch" % (n, kind, e0.elapsed_time(e1) / 50 * 1e3))

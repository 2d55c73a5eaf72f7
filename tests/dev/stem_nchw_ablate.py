"""Dev: the NCHW-fed pooled stem (hps_stem_winograd_pooled_nchw) against phase split + frame-fed pooled stem, and its gather ablations
(8 = no loads, 9 = no LDS stores), interleaved medians.  usage: stem_nchw_ablate.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hierarchicalprobabilistic3dhuman_amd import _capi
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN

dev = torch.device("cuda:0")
P = _capi.ptr
B, H, W = 64, 256, 256
torch.manual_seed(0)
conv = torch.nn.Conv2d(18, 64, 7, 2, 3, bias=False).to(dev)
bn = torch.nn.BatchNorm2d(64).to(dev).eval()
cb = _ConvBN(conv, bn, cin_pad=20)
x = torch.randn(B, 18, H, W, device=dev)
with _capi.dev_library():
    s = _capi.stream()
    L = _capi.load(dev=True)
    fr = torch.zeros(int(L.hps_stem_phase_frames_bytes(B, H, W)) // 4, device=dev)
    side = torch.empty(int(L.hps_stem_pool_side_bytes(B, H, W)) // 4, device=dev)
    y = torch.zeros(B, H // 4 + 2, W // 4 + 2, 64, device=dev)
    fns = {
        "phase split": lambda: _capi.call("hps_stem_phase_split", P(x), P(fr), B, 18, H, W, s),
        "frame-fed stem + pool": lambda: _capi.call("hps_stem_winograd_pooled", P(fr), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), P(side), B, H, W, 1, 1, s),
        "NCHW-fed stem + pool": lambda: _capi.call("hps_dev_stem_winograd_pooled_nchw", P(x), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), P(side), B, H, W, 1, 1, 0, s),
        "  without the gather's loads": lambda: _capi.call("hps_dev_stem_winograd_pooled_nchw", P(x), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), P(side), B, H, W, 1, 1, 8, s),
        "  without its LDS stores": lambda: _capi.call("hps_dev_stem_winograd_pooled_nchw", P(x), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), P(side), B, H, W, 1, 1, 9, s),
        "  duplicate input transform skipped (upper bound of removing it)": lambda: _capi.call("hps_dev_stem_winograd_pooled_nchw", P(x), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), P(side), B, H, W, 1, 1, 10, s),
        "  transform SPLIT between the pair's waves, no exchange (upper bound)": lambda: _capi.call("hps_dev_stem_winograd_pooled_nchw", P(x), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), P(side), B, H, W, 1, 1, 12, s),
        "  ... + 23 LDS stores, 23 reads per lane and row and a second barrier": lambda: _capi.call("hps_dev_stem_winograd_pooled_nchw", P(x), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), P(side), B, H, W, 1, 1, 13, s),
        "  ... replaced by 45 LDS reads per lane and row + a second barrier": lambda: _capi.call("hps_dev_stem_winograd_pooled_nchw", P(x), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), P(side), B, H, W, 1, 1, 11, s),
    }
    for f in fns.values():
        f()
    ts = {k: [] for k in fns}
    for rnd in range(9):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 10)
    for k in fns:
        t = sorted(ts[k])
        print("%-70s median %.4f ms (min %.4f max %.4f)" % (k, t[len(t) // 2], t[0], t[-1]))

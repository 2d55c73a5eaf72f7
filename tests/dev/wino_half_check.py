"""The half-item form of the Winograd kernel (hps_dev_conv3x3_winograd_half: 4 x 8-tile items, two four-wave workgroups per CU) against the product
(hps_dev_conv3x3_winograd, ablate = 0): identical bits, and the time of both.  Dev library.  usage: wino_half_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd import _capi  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN  # noqa: E402

dev = torch.device("cuda:0")
P = _capi.ptr


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def pack_u4(u, C):
    """wino_u [chunk8][ct][16][kq 2][64][4]  ->  [half-chunk][ct][16][64][kl 2][slot 2]: channel 8 c + 2 s + slot + 4 kl"""
    c8, nct = C // 8, C // 64
    v = u.view(c8, nct, 16, 2, 64, 2, 2)                     # (c8, ct, p, kl, co, s, slot)
    return v.permute(0, 5, 1, 2, 4, 3, 6).contiguous().view(-1)


with _capi.dev_library():
    torch.manual_seed(0)
    for (H, C) in ((64, 64), (32, 128), (16, 256)):
        conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(C).eval().to(dev)
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        cb = _ConvBN(conv, bn)
        u4 = pack_u4(cb.wino_u, C)
        for B in (64, 3):
            x = F.pad(torch.randn(B, H, H, C, device=dev), (0, 0, 1, 1, 1, 1)).contiguous()
            res = torch.randn(B, H + 2, H + 2, C, device=dev)
            for use_res in (False, True):
                for relu in (1, 0):
                    out0 = torch.zeros(B, H + 2, H + 2, C, device=dev)
                    out1 = torch.zeros(B, H + 2, H + 2, C, device=dev)
                    _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), P(res) if use_res else None, P(out0),
                               B, H, H, 1, C, C, 1, relu, None, 0, _capi.stream())
                    _capi.call("hps_dev_conv3x3_winograd_half", P(x), P(u4), P(cb.scale), P(cb.shift), P(res) if use_res else None, P(out1),
                               B, H, H, 1, C, C, 1, relu, 0, 2, _capi.stream())
                    torch.cuda.synchronize()
                    halo = float(out1[:, 0].abs().max() + out1[:, :, 0].abs().max() + out1[:, -1].abs().max() + out1[:, :, -1].abs().max())
                    print("%2dx%-2d C=%3d B=%2d residual=%d relu=%d: identical=%s max|diff|=%.3g halo=%.1g" % (
                        H, H, C, B, use_res, relu, torch.equal(out0, out1), float((out0 - out1).abs().max()), halo), flush=True)
        x = F.pad(torch.relu(torch.randn(64, H, H, C, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
        out = torch.zeros(64, H + 2, H + 2, C, device=dev)
        res = torch.randn(64, H + 2, H + 2, C, device=dev)
        for use_res in (False, True):
            r = P(res) if use_res else None
            fns = {
                "product": lambda: _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), r, P(out), 64, H, H, 1, C, C, 1, 1,
                                              None, 0, _capi.stream()),
                "half x2": lambda: _capi.call("hps_dev_conv3x3_winograd_half", P(x), P(u4), P(cb.scale), P(cb.shift), r, P(out), 64, H, H, 1, C, C, 1, 1,
                                              0, 2, _capi.stream()),
                "half x1": lambda: _capi.call("hps_dev_conv3x3_winograd_half", P(x), P(u4), P(cb.scale), P(cb.shift), r, P(out), 64, H, H, 1, C, C, 1, 1,
                                              0, 1, _capi.stream()),
                "half x2, no epilogue": lambda: _capi.call("hps_dev_conv3x3_winograd_half", P(x), P(u4), P(cb.scale), P(cb.shift), r, P(out), 64, H, H, 1,
                                                           C, C, 1, 1, 4, 2, _capi.stream()),
            }
            ts = {k: [] for k in fns}
            for rep in range(3):
                for k, fn in fns.items():
                    ts[k].append(timeit(fn))
            print("%2dx%-2d C=%3d B=64 residual=%d: " % (H, H, C, use_res) + " | ".join("%s %.4f ms" % (k, sorted(v)[1]) for k, v in ts.items()), flush=True)

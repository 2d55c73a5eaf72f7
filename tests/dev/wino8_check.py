"""The eight-wave forms of the Winograd kernel (the product: hps_dev_conv3x3_winograd, ablate = 0 -- a lane owns a channel, 4-byte stores; ablate = 23 --
a lane owns a tile, 16-byte stores) against the four-wave kernel (ablate = 21): identical bits (with / without residual and ReLU, full and small
batches), and the time of all three (+ the product without its epilogue, ablate = 24).  Dev library.  usage: wino8_check.py"""
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


with _capi.dev_library():
    torch.manual_seed(0)
    for (H, C) in ((64, 64), (32, 128), (16, 256), (8, 512)):
        conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(C).eval().to(dev)
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        cb = _ConvBN(conv, bn)
        ws = torch.empty(4 * 64 * 64 * C, device=dev)          # split-K workspace of the 8 x 8 geometry
        for B in (64, 3):
            x = F.pad(torch.randn(B, H, H, C, device=dev), (0, 0, 1, 1, 1, 1)).contiguous()
            res = torch.randn(B, H + 2, H + 2, C, device=dev)
            for use_res in (False, True):
                for relu in (1, 0):
                    outs = []
                    for ab in (21, 23, 0):
                        out = torch.zeros(B, H + 2, H + 2, C, device=dev)
                        _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), P(res) if use_res else None, P(out),
                                   B, H, H, 1, C, C, 1, relu, P(ws) if H == 8 else None, ab, _capi.stream())
                        torch.cuda.synchronize()
                        outs.append(out)
                    same = torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
                    err = float((outs[0] - outs[2]).abs().max())
                    halo = float(outs[2][:, 0].abs().max() + outs[2][:, :, 0].abs().max() + outs[2][:, -1].abs().max() + outs[2][:, :, -1].abs().max())
                    print("%2dx%-2d C=%3d B=%2d residual=%d relu=%d: identical=%s max|diff|=%.3g halo=%.1g" % (H, H, C, B, use_res, relu, same, err, halo), flush=True)
        x = F.pad(torch.relu(torch.randn(64, H, H, C, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
        out = torch.zeros(64, H + 2, H + 2, C, device=dev)
        res = torch.randn(64, H + 2, H + 2, C, device=dev)
        for use_res in (False, True):
            ts = []
            for rep in range(3):
                for ab in (21, 23, 0) + ((24, 31, 33, 35, 36, 40, 42) if H != 8 else ()):
                    fn = lambda: _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), P(res) if use_res else None,
                                            P(out), 64, H, H, 1, C, C, 1, 1, P(ws) if H == 8 else None, ab, _capi.stream())
                    ts.append((ab, timeit(fn)))
            t4 = sorted(t for a, t in ts if a == 21)[1]
            t8t = sorted(t for a, t in ts if a == 23)[1]
            t8 = sorted(t for a, t in ts if a == 0)[1]
            print("%2dx%-2d C=%3d B=64 residual=%d: four waves %.4f ms, eight waves (lane = channel, product) %.4f ms (%+.1f %%), eight waves (lane = tile) %.4f ms"
                  % (H, H, C, use_res, t4, t8, 100 * (t8 / t4 - 1), t8t), flush=True)
            if H != 8:
                print("      product form without: epilogue %.4f | patch reads, transform, window DMA %.4f | filter DMA %.4f | transform (window DMA kept) %.4f | "
                      "window DMA (transform kept) %.4f | barrier per chunk %.4f || with s_setprio around the MFMAs %.4f" % tuple(sorted(t for a, t in ts if a == ab)[1] for ab in (24, 31, 33, 35, 36, 40, 42)), flush=True)

"""Dev, timing only: what the Winograd layers would cost with their MFMAs as bf16x3 piece products (DESIGN.md section 11,
docs/experiments_r6.md section 9).  hps_dev_conv3x3_winograd ablate 43 = every position's four fp32 MFMAs per 8 k replaced by the three
v_mfma_f32_32x32x16_bf16 that six piece products per 16 k amount to, on whatever bits the fp32 fragments hold; ablate 44 = that + the VALU
work of splitting every transformed input value into three bf16 pieces (v_cvt_pk_bf16_f32, two values per instruction).  The LDS
layout, the operand bytes and the DMA are the fp32 kernel's (a real kernel would move 1.5 x the operand bytes): an optimistic bound.
Results are garbage.  usage: wino_bf16x3_price.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd import _capi  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN  # noqa: E402

dev = torch.device("cuda:0")
P = _capi.ptr


def timeit(fn, iters=30, warmup=5):
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


LAYERS = {64: 4, 32: 3, 16: 3}          # stride-1 3x3 layers of the 16 x 16-block geometry per map size (layer4's 8 x 8 maps take the quad geometry: not priced)
with _capi.dev_library():
    torch.manual_seed(0)
    tot = {0: 0.0, 43: 0.0, 44: 0.0, 24: 0.0}
    for (H, C) in ((64, 64), (32, 128), (16, 256)):
        conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(C).eval().to(dev)
        cb = _ConvBN(conv, bn)
        x = F.pad(torch.relu(torch.randn(64, H, H, C, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
        out = torch.zeros(64, H + 2, H + 2, C, device=dev)
        res = torch.randn(64, H + 2, H + 2, C, device=dev)
        ts = {}
        for rep in range(3):
            for ab in (0, 43, 44, 24):
                fn = lambda: _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), P(res), P(out), 64, H, H, 1, C, C, 1, 1,
                                        None, ab, _capi.stream())
                ts.setdefault(ab, []).append(timeit(fn))
        med = {ab: sorted(v)[1] for ab, v in ts.items()}
        for ab in tot:
            tot[ab] += LAYERS[H] * med[ab]
        print("%2dx%-2d C=%3d B=64: fp32 kernel %.4f ms | MFMAs as bf16x3 %.4f | + split VALU %.4f | (fp32 kernel without its epilogue %.4f)"
              % (H, H, C, med[0], med[43], med[44], med[24]), flush=True)
    print("ten layers of the 16 x 16-block geometry per step: fp32 %.3f ms | MFMAs as bf16x3 %.3f | + split VALU %.3f" % (tot[0], tot[43], tot[44]))

"""Where a work item of the Winograd kernel spends its time between two K loops: shader-clock stamps (hps_dev_conv3x3_winograd, ablate = 11) of
every workgroup's transition from its first to its second item.  Dev library.  usage: wino_phases.py"""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hierarchicalprobabilistic3dhuman_amd import _capi  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN  # noqa: E402

dev = torch.device("cuda:0")
P = _capi.ptr
NAMES = ["K loop ends", "barrier (LDS free)", "next item's DMAs issued", "row transform + exchange stores + residual loads issued", "barrier (exchange)",
         "column transform + stores issued", "vmcnt(0)", "barrier", "first input transform", "chunk 0", "chunk 1"]

with _capi.dev_library():
    torch.manual_seed(0)
    for (H, C) in ((64, 64), (32, 128)):
        conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(C).eval().to(dev)
        cb = _ConvBN(conv, bn)
        x = F.pad(torch.relu(torch.randn(64, H, H, C, device=dev)), (0, 0, 1, 1, 1, 1)).contiguous()
        out = torch.zeros(64, H + 2, H + 2, C, device=dev)
        res = torch.randn(64, H + 2, H + 2, C, device=dev)
        for use_res in (False, True):
            for _ in range(3):
                _capi.call("hps_dev_conv3x3_winograd", P(x), P(cb.wino_u), P(cb.scale), P(cb.shift), P(res) if use_res else None, P(out),
                           64, H, H, 1, C, C, 1, 1, None, 11, _capi.stream())
            torch.cuda.synchronize()
            buf = np.zeros(256 * 16, dtype=np.uint64)
            _capi.call("hps_dev_wino_stamps", buf.ctypes.data_as(ctypes.c_void_p), buf.size)
            st = buf.reshape(256, 16)[:, :11].astype(np.int64)
            d = np.diff(st, axis=1)
            print("%dx%d C=%d residual=%d: clock ticks between stamps, median over 256 workgroups (min .. max)" % (H, H, C, use_res))
            for k in range(10):
                print("   %-62s -> %-62s %7d  (%d .. %d)" % (NAMES[k], NAMES[k + 1], np.median(d[:, k]), d[:, k].min(), d[:, k].max()))
            print("   K loop end -> first chunk done: %d ticks" % np.median(st[:, 9] - st[:, 0]))

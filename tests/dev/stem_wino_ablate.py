"""Dev: interleaved timing of the Winograd stem's ablations (libhps_dev.so)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hierarchicalprobabilistic3dhuman_amd import _capi
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN

dev = torch.device("cuda:0")
P = _capi.ptr
modes = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5]
B, H, W = 64, 256, 256
torch.manual_seed(0)
conv = torch.nn.Conv2d(18, 64, 7, 2, 3, bias=False).to(dev)
bn = torch.nn.BatchNorm2d(64).to(dev).eval()
cb = _ConvBN(conv, bn, cin_pad=20)
x = torch.randn(B, 18, H, W, device=dev)
with _capi.dev_library():
    s = _capi.stream()
    fr = torch.zeros(int(_capi.load(dev=True).hps_stem_phase_frames_bytes(B, H, W)) // 4, device=dev)
    _capi.call("hps_stem_phase_split", P(x), P(fr), B, 18, H, W, s)
    y = torch.empty(B, H // 2, W // 2, 64, device=dev)
    fn = lambda m: _capi.call("hps_dev_stem_winograd", P(fr), P(cb.stem_u), P(cb.scale), P(cb.shift), P(y), B, H, W, 0, 1, m, s)
    for m in modes:
        fn(m)
    ts = {m: [] for m in modes}
    for rnd in range(9):
        for m in modes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn(m)
            e1.record()
            torch.cuda.synchronize()
            ts[m].append(e0.elapsed_time(e1) / 10)
    for m in modes:
        t = sorted(ts[m])
        print("stem winograd ablate %d: median %.4f ms (min %.4f max %.4f)" % (m, t[len(t) // 2], t[0], t[-1]))

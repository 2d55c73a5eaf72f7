"""Dev check: Winograd stem (csrc/stem_wino.hip) against the direct row-mode stem and an fp64 CPU convolution; timing."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hierarchicalprobabilistic3dhuman_amd import _capi
from hierarchicalprobabilistic3dhuman_amd.resnet import _ConvBN

dev = torch.device("cuda:0")
P = _capi.ptr


def run(B, H, W, seed=0, time_it=False):
    torch.manual_seed(seed)
    conv = torch.nn.Conv2d(18, 64, 7, 2, 3, bias=False)
    bn = torch.nn.BatchNorm2d(64)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    conv, bn = conv.to(dev), bn.to(dev).eval()
    cb = _ConvBN(conv, bn, cin_pad=20)
    x = torch.randn(B, 18, H, W, device=dev)
    s = _capi.stream()
    # reference fp64 on the CPU
    with torch.no_grad():
        ref = torch.relu(bn.cpu().double()(conv.cpu().double()(x.cpu().double()))).permute(0, 2, 3, 1).contiguous()
    # direct
    xin = torch.zeros(B, H + 6, W + 6, 18, device=dev)
    _capi.call("hps_nchw_to_padded_nhwc", P(x), P(xin), B, 18, H, W, 3, s)
    yd = torch.empty(B, H // 2, W // 2, 64, device=dev)
    cb.padded(xin, 3, yd, 0, relu=True)
    # winograd
    nbytes = int(_capi.load().hps_stem_phase_frames_bytes(B, H, W))
    fr = torch.zeros(nbytes // 4, device=dev)
    _capi.call("hps_stem_phase_split", P(x), P(fr), B, 18, H, W, s)
    yw = torch.full((B, H // 2, W // 2, 64), float("nan"), device=dev)
    _capi.call("hps_stem_winograd", P(fr), P(cb.stem_u), P(cb.scale), P(cb.shift), P(yw), B, H, W, 0, 1, s)
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    ed = float((yd.cpu().double() - ref).abs().max()) / sc
    ew = float((yw.cpu().double() - ref).abs().max()) / sc
    print("B=%d %dx%d: scale %.3f  direct err %.2e  winograd err %.2e  (nan: %d)" % (B, H, W, sc, ed, ew, int(torch.isnan(yw).sum())))
    if time_it:
        for name, fn in (("direct", lambda: cb.padded(xin, 3, yd, 0, relu=True)),
                         ("winograd", lambda: _capi.call("hps_stem_winograd", P(fr), P(cb.stem_u), P(cb.scale), P(cb.shift), P(yw), B, H, W, 0, 1, s)),
                         ("split", lambda: _capi.call("hps_stem_phase_split", P(x), P(fr), B, 18, H, W, s)),
                         ("relayout", lambda: _capi.call("hps_nchw_to_padded_nhwc", P(x), P(xin), B, 18, H, W, 3, s))):
            for _ in range(3):
                fn()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            ts.sort()
            print("   %-9s median %.4f ms (min %.4f max %.4f)" % (name, ts[len(ts) // 2], ts[0], ts[-1]))
    return ew


if __name__ == "__main__":
    run(1, 32, 32)
    run(2, 64, 96)
    run(3, 256, 256)
    run(64, 256, 256, time_it=True)
    run(16, 256, 256, time_it=True)

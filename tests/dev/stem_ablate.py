import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hierarchicalprobabilistic3dhuman_amd import _capi, configs
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, configs.get_cfg_defaults()).eval().to(dev)
enc = net.image_encoder
enc.set_winograd(False)          # the direct row-mode stem (the product runs the Winograd stem: tests/dev/stem_wino_ablate.py)
x = torch.rand(64, 18, 256, 256, device=dev)
with torch.no_grad(), _capi.dev_library():
    enc(x); torch.cuda.synchronize()
    fs = next(iter(enc._frames.values()))
    ops = fs["ops"]
    one = (_capi.EncOp * 1)(ops[1])
    s = _capi.stream()
    import sys as _s
    modes = [int(a) for a in _s.argv[1:]] or [0, 1]
    # clocks drift by a few per cent over seconds: interleave the modes over several rounds and compare medians
    times = {m: [] for m in modes}
    for _ in range(3):
        for _ in range(5): _capi.call("hps_encoder_run", one, 1, s)
    for rnd in range(9):
        for ab in modes:
            _capi.call("hps_dev_conv_pad_ablate", ab)
            for _ in range(2): _capi.call("hps_encoder_run", one, 1, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): _capi.call("hps_encoder_run", one, 1, s)
            e1.record(); torch.cuda.synchronize()
            times[ab].append(e0.elapsed_time(e1) / 10)
    for ab in modes:
        t = sorted(times[ab])
        print("stem ablate %d: median %.4f ms (min %.4f max %.4f, 9 rounds of 10 launches, modes interleaved)" % (ab, t[4], t[0], t[-1]))
    _capi.call("hps_dev_conv_pad_ablate", 0)

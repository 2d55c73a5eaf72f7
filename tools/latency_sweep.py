"""Batch-1 sweep of every convolution of the encoder: Winograd vs direct with ksplit in {1, 2, 4, 8, ...} (what
ResNet.set_latency_mode picks per layer).     python tools/latency_sweep.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import configs  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet  # noqa: E402


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, configs.get_cfg_defaults()).eval().to(dev).image_encoder
    prep = enc.prepare()
    h = w = 64
    names = []
    for li in range(4):
        for bi in range(2):
            names.append("layer%d.%d" % (li + 1, bi))
    for name, (c1, c2, down) in zip(names, prep["blocks"]):
        for tag, c, hin in (("down", down, h), ("conv1", c1, h), ("conv2", c2, None)):
            if c is None:
                continue
            if tag == "conv2":
                hin = ho
            ho = c.out_hw(hin, hin)[0]
            xp = torch.randn(B, hin + 2, hin + 2, c.cin_p, device=dev)
            out = torch.zeros(B, ho + 2, ho + 2, c.cout, device=dev)
            res = []
            chunks = c.kh * c.kw * c.cin_p // 32
            c.use_winograd = True
            if c.winograd_ok(hin, hin, 1):
                res.append("wino %.1f" % timed(lambda: c.padded(xp, 1, out, 1)))
            c.use_winograd = False
            for ks in (1, 2, 3, 4, 6, 8, 9, 12, 16, 18):
                if chunks % ks or (ks > 1 and chunks // ks < 2):
                    continue
                c.ksplit = ks
                ws = torch.empty(ks, B * ho * ho, c.cout, device=dev) if ks > 1 else None
                try:
                    res.append("k%d %.1f" % (ks, timed(lambda: c.padded(xp, 1, out, 1, ws=ws))))
                except Exception as e:
                    res.append("k%d err" % ks)
            c.ksplit = 0
            print("%-10s %-6s %3dx%-3d %3d->%-3d  us: %s" % (name, tag, hin, hin, c.cin_p, c.cout, "  ".join(res)), flush=True)
        h = ho


if __name__ == "__main__":
    main()

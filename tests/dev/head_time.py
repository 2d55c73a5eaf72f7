"""Head alone (device SVD): ms per call and per kinematic level, B = 64.  python tests/dev/head_time.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet
from hierarchicalprobabilistic3dhuman_amd import configs

torch.manual_seed(0)
dev = torch.device("cuda:0")
net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, configs.get_cfg_defaults()).eval().to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if len(sys.argv) > 2:
    net.svd_mode = sys.argv[2]
feats = torch.randn(B, 512, device=dev)
for _ in range(3):
    net(None, input_feats=feats)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
t0 = time.perf_counter()
e0.record()
for _ in range(n):
    net(None, input_feats=feats)
e1.record()
t_host = (time.perf_counter() - t0) / n * 1e3
torch.cuda.synchronize()
print("head alone, B=%d: %.3f ms per call on the device, %.3f ms of host time to enqueue" % (B, e0.elapsed_time(e1) / n, t_host))

import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import ref_cpu as O
from hierarchicalprobabilistic3dhuman_amd.canny_edge_detector import CannyEdgeDetector
dev = torch.device('cuda:0')
for shape in [(1,3,64,64),(1,3,256,256),(2,1,33,32),(1,3,40,300)]:
    B,C,H,W = shape
    g = torch.Generator().manual_seed(1)
    img = torch.rand(B,C,H,W,generator=g)
    ref = O.canny_edge_detector(img, True, 1.0, 5, 0.05)
    out = CannyEdgeDetector(True, 1.0, 5, 0.05).to(dev)(img.to(dev))
    for k in ref:
        d = (out[k].cpu()-ref[k]).abs()
        bad = (d > 2e-6).nonzero()
        print(shape, k, float(d.max()), len(bad), bad[:6].tolist())

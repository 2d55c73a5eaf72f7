import os, sys, torch
sys.path.insert(0, "/root/repo")
from hierarchicalprobabilistic3dhuman_amd import sampling_utils as su
from oracle import ref_cpu as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
for n, nv, b in ((769, 50, 1), (769, 64, 1), (769, 6890, 1), (800, 16, 1), (1000, 6890, 2), (1024, 16, 3), (1000, 48, 1), (1000, 50, 1)):
    vv = torch.randn(b, n, nv, 3, generator=g) * 0.3 + torch.randn(b, 1, nv, 3, generator=g)
    want = torch.stack([O.vertex_uncertainty(vv[i]) for i in range(b)])
    got = su.vertex_uncertainty(vv.to(dev)).cpu()
    bad = ~torch.isfinite(got) | ((got - want).abs() > 1e-5)
    print(n, nv, b, "bad", int(bad.sum()), "of", bad.numel(), "first bad idx", bad.nonzero()[:6].tolist(), flush=True)

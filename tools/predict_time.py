"""Throughput of predict_poseMF_shapeGaussian_net (the reference's run_predict.py entry point) on the GPU box, with the
image-dependent front end replaced by stored proxy representations (proxy_rep_fn) and a no-op result_fn: the loop the host layer adds
around infer().

    python tools/predict_time.py [images] [batch] [samples] [--pageable] [--latency]
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import predict_poseMF_shapeGaussian_net  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(pos[0]) if len(pos) > 0 else 128
    batch = int(pos[1]) if len(pos) > 1 else 16
    samples = int(pos[2]) if len(pos) > 2 else 50
    dev = torch.device("cuda:0")
    cfg = configs.get_cfg_defaults()
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, cfg).eval().to(dev)
    if "--latency" in sys.argv:              # the per-model switch for one-image-at-a-time deployments (direct kernels with many K slices)
        net.set_latency_mode(True)
    smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
    g = torch.Generator().manual_seed(1)
    pinned = "--pageable" not in sys.argv
    proxies = [torch.rand(1, 18, 256, 256, generator=g) for _ in range(8)]
    if pinned:                               # stored proxy representations in page-locked memory: the H2D copies are asynchronous
        proxies = [t.pin_memory() for t in proxies]
    with tempfile.TemporaryDirectory() as d:
        for i in range(n):
            open(os.path.join(d, "img_%04d.png" % i), "wb").close()          # names only: proxy_rep_fn supplies the content
        fn = lambda path: proxies[hash(path) % 8]
        sink = lambda name, item: None
        kw = dict(proxy_rep_fn=fn, num_samples=samples, batch_size=batch, result_fn=sink)
        predict_poseMF_shapeGaussian_net(net, cfg, smpl, None, None, None, dev, d, os.path.join(d, "out"), **kw)      # warm-up
        torch.cuda.synchronize()
        t0 = time.time()
        predict_poseMF_shapeGaussian_net(net, cfg, smpl, None, None, None, dev, d, os.path.join(d, "out"), **kw)
        torch.cuda.synchronize()
        dt = time.time() - t0
    print("predict loop (%s host tensors%s): %d images, batch %d, %d samples: %.3f s = %.0f images/s (%.3f ms per image)" % (
        "page-locked" if pinned else "pageable", ", latency mode" if "--latency" in sys.argv else "", n, batch, samples, dt, n / dt, 1e3 * dt / n))


if __name__ == "__main__":
    main()

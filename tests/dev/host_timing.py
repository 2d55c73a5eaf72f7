"""Host-side time per phase of the pipelined bench loop (un-profiled): where does the Python thread spend a step?

    python tests/dev/host_timing.py [steps]
"""
import os
import sys
import time
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data, sharding, sampling_utils  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd import predict_poseMF_shapeGaussian_net as pred  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet  # noqa: E402
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL  # noqa: E402

T = collections.defaultdict(float)


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        T[name] += time.perf_counter() - t
        return r
    return w


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    cfg = configs.get_cfg_defaults()
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, cfg).eval().to(dev)
    smpl = SMPL(smpl_data.synthetic_smpl_model(0), batch_size=1, gender="neutral", num_betas=10).to(dev)
    x = bench.synthetic_inputs(0, 64).to(dev)
    pipe = pred.InferencePipeline(net, smpl, num_samples=100, use_mean_shape=True)
    sums = torch.zeros(4, dtype=torch.float64, device=dev)

    def loop(count):
        ticket = pipe.submit(x)
        for i in range(count):
            nxt = pipe.submit(x) if i + 1 < count else None
            res = pipe.finish(ticket, seed=1234 + i, image_offset=0, after=nxt)
            sums.add_(sharding.batch_metric_sums(res))
            ticket = nxt

    loop(8)
    torch.cuda.synchronize()
    # instrument
    pipe.submit = timed("submit", pipe.submit)
    pipe.finish = timed("finish (total)", pipe.finish)
    net.forward = timed("  head (net.forward)", net.forward)
    smpl.forward = timed("  smpl.forward launches", smpl.forward)
    pred.pose_matrix_fisher_sampling_torch = timed("  sampler launch", pred.pose_matrix_fisher_sampling_torch)
    pred.vertex_uncertainty = timed("  uncertainty launch", pred.vertex_uncertainty)
    sharding.batch_metric_sums = timed("metric sums", sharding.batch_metric_sums)
    t0 = time.perf_counter()
    loop(steps)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("steps %d: %.3f ms/step wall (host loop returned after %.3f ms/step)" % (steps, t_all / steps * 1e3, t_host / steps * 1e3))
    for k, v in T.items():
        print("  %-28s %.3f ms/step" % (k, v / steps * 1e3))


if __name__ == "__main__":
    main()

"""Which overlapped work slows the pipelined encoder?  python tests/dev/contention.py <drop> [bench args]
drop: comma list of {unc, sums, none} -- the named work is skipped (results are then wrong: timing experiment only)."""
import os
import runpy
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import predict_poseMF_shapeGaussian_net as pred, sharding

drop = set(sys.argv[1].split(","))
if "unc" in drop:
    pred.vertex_uncertainty = lambda v: torch.zeros(v.shape[0], v.shape[2], device=v.device)
if "sums" in drop:
    z = {}
    def no_sums(res):
        d = res["unc"].device
        if d not in z:
            z[d] = torch.zeros(4, dtype=torch.float64, device=d)
        return z[d]
    sharding.batch_metric_sums = no_sums
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

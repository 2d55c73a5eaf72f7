"""Every launch of ONE steady-state step of the pipelined loop, named.

    python tools/step_launches.py aten            # every ATen operator torch dispatches during one step, with its Python call site
    python tools/step_launches.py trace <kernel_trace.csv>   # every kernel of one step of a `rocprofv3 --kernel-trace` of bench.py

`aten`: the loop of bench.py (InferencePipeline over two resident input batches, B = 64, N = 100) runs a few steps, then one step under a
TorchDispatchMode that logs each operator that reaches the dispatcher (the libhps launches go through ctypes and are not torch operators:
what the mode sees is exactly the non-`hps` work of a step).  Operators that only allocate or view (empty, view, as_strided, ...) launch
nothing and are listed separately.
`trace`: kernels between two consecutive mesh-kernel launches of the timed region, grouped by name with their queue, split into hps:: kernels
and everything else.
"""
import collections
import csv
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NO_LAUNCH = ("aten.empty", "aten.view", "aten.as_strided", "aten._unsafe_view", "aten.slice", "aten.select", "aten.unbind", "aten.t.",
             "aten.transpose", "aten.expand", "aten.reshape", "aten.detach", "aten.alias", "aten.squeeze", "aten.unsqueeze",
             "aten.record_stream", "aten.is_pinned", "aten._local_scalar_dense", "aten.lift_fresh", "aten.permute", "aten.split")


def aten():
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode
    from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data, sharding
    from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import InferencePipeline
    from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, configs.get_cfg_defaults()).eval().to(dev)
    smpl = SMPL(smpl_data.synthetic_smpl_model(0), batch_size=1, gender="neutral", num_betas=10).to(dev)
    xs = [torch.rand(64, 18, 256, 256, device=dev) for _ in range(2)]
    pipe = InferencePipeline(net, smpl, num_samples=100, use_mean_shape=True)
    sums = torch.zeros(4, dtype=torch.float64, device=dev)
    log = []

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            frames = [f for f in traceback.extract_stack() if "site-packages" not in f.filename and "step_launches" not in f.filename
                      and "/lib/python" not in f.filename]
            site = "%s:%d" % (os.path.relpath(frames[-1].filename, ROOT), frames[-1].lineno) if frames else "?"
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            log.append((str(func), site, shapes))
            return func(*args, **(kwargs or {}))

    def step(i, ticket, last=False):
        nxt = None if last else pipe.submit(xs[(i + 1) % 2], input_ready=False)
        res = pipe.finish(ticket, seed=1234 + i, image_offset=0, after=nxt)
        sharding.batch_metric_sums(res, accumulate=sums)
        return nxt

    t = pipe.submit(xs[0], input_ready=False)
    for i in range(6):
        t = step(i, t)
    with Log():
        t = step(6, t)
    t = step(7, t, last=True)
    torch.cuda.synchronize()
    launching = [(f, s, sh) for f, s, sh in log if not any(f.startswith(n) for n in NO_LAUNCH)]
    print("one steady-state step (submit of batch i + 1, finish of batch i, checksum): %d ATen operators reach the dispatcher, %d of them "
          "launch device work" % (len(log), len(launching)))
    for f, s, sh in launching:
        print("  LAUNCH  %-34s %-70s %s" % (f, s, sh))
    quiet = collections.Counter((f, s) for f, s, sh in log if any(f.startswith(n) for n in NO_LAUNCH))
    for (f, s), n in sorted(quiet.items(), key=lambda kv: -kv[1]):
        print("  no launch x%-3d %-30s %s" % (n, f, s))


def trace(path):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
    mesh = [e for e in ev if "mesh_fused_kernel" in e[2]]
    k = min(6, len(mesh) - 2)
    s0, s1 = mesh[k][0], mesh[k + 1][0]
    step = [e for e in ev if s0 <= e[0] < s1]
    short = lambda n: n.split("(")[0].replace("void ", "")[:90]
    hps = collections.Counter((short(n), q) for s, e, n, q in step if "hps::" in n)
    other = collections.Counter((short(n), q) for s, e, n, q in step if "hps::" not in n)
    print("one steady-state step of the timed region (mesh kernel to mesh kernel): %.3f ms, %d kernels: %d hps, %d other"
          % ((s1 - s0) / 1e6, len(step), sum(hps.values()), sum(other.values())))
    for (n, q), c in sorted(hps.items(), key=lambda kv: kv[0]):
        print("  hps    x%-3d queue %-3s %s" % (c, q, n))
    for (n, q), c in sorted(other.items(), key=lambda kv: kv[0]):
        print("  OTHER  x%-3d queue %-3s %s" % (c, q, n))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        aten()

"""Batch-1 latency of infer() (the reference's run_predict operating point: one image, num_samples = 50): wall clock per call, and --
under `rocprofv3 --kernel-trace` -- the kernel timeline of one call (tools/latency_b1.py analyse <kernel_trace.csv>).

    python tools/latency_b1.py [reps] [--direct | --latency]   # ResNet.set_winograd(False) / set_latency_mode(True)
    python tools/latency_b1.py analyse <kernel_trace.csv>
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(path):
    import csv
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    # one call = from a phase-split / relayout kernel to the uncertainty kernel
    starts = [i for i, e in enumerate(ev) if "stem_phase_split" in e[2] or "nchw_to_padded" in e[2]]
    ends = [i for i, e in enumerate(ev) if "uncertainty" in e[2]]
    if not starts or not ends:
        print("no calls found"); return
    calls = []
    for s in starts:
        e = next((x for x in ends if x > s), None)
        if e is not None:
            calls.append((s, e))
    s, e = calls[len(calls) // 2]
    t0 = ev[s][0]
    busy = 0
    prev_end = t0
    print("%9s %8s %7s  kernel" % ("start us", "dur us", "gap us"))
    for a, b, name in ev[s:e + 1]:
        nm = name.split("(")[0].replace("void ", "").replace("hps::", "")[:60]
        print("%9.1f %8.1f %7.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, nm))
        busy += b - a
        prev_end = max(prev_end, b)
    total = (ev[e][1] - t0) / 1e3
    print("kernels %d, first start -> last end %.1f us, sum of kernel times %.1f us, gaps %.1f us" % (e - s + 1, total, busy / 1e3, total - busy / 1e3))
    durs = sorted((ev[b][1] - ev[a][0]) / 1e3 for a, b in calls[2:])
    print("calls %d: median device span %.1f us" % (len(durs), durs[len(durs) // 2]))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "analyse":
        return analyse(sys.argv[2])
    import torch
    from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data
    from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import infer
    from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(pos[0]) if pos else 40
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, configs.get_cfg_defaults()).eval().to(dev)
    if "--direct" in sys.argv:
        net.image_encoder.set_winograd(False)
    if "--latency" in sys.argv:
        net.set_latency_mode(True)
    smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
    x = torch.rand(1, 18, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    lat, host = [], []
    for i in range(reps + 5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        infer(net, smpl, x, num_samples=50, seed=7 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
        host.append((t1 - t0) * 1e3)
    lat, host = sorted(lat[5:]), sorted(host[5:])
    print("batch 1, N = 50%s: latency median %.3f ms (min %.3f max %.3f); host enqueue median %.3f ms" % (
        " (direct convolutions)" if "--direct" in sys.argv else (" (latency mode)" if "--latency" in sys.argv else ""), lat[len(lat) // 2], lat[0], lat[-1], host[len(host) // 2]))


if __name__ == "__main__":
    main()

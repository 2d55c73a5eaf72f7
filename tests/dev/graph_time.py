"""Dev: one batch-1 infer() call issued launch by launch against its hipGraph replay (GraphedInfer), both encoder modes: host wall clock of
issue + wait, and of back-to-back replays without a wait in between.  usage: graph_time.py [reps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hierarchicalprobabilistic3dhuman_amd import configs, smpl_data
from hierarchicalprobabilistic3dhuman_amd.poseMF_shapeGaussian_net import PoseMFShapeGaussianNet
from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import infer, GraphedInfer
from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL

reps = int([a for a in sys.argv[1:] if not a.startswith('--')][0]) if [a for a in sys.argv[1:] if not a.startswith('--')] else 40
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = PoseMFShapeGaussianNet(configs.SMPL_PARENTS, configs.get_cfg_defaults()).eval().to(dev)
smpl = SMPL(smpl_data.synthetic_smpl_model(0)).to(dev)
x = torch.rand(1, 18, 256, 256, device=dev)
med = lambda v: sorted(v)[len(v) // 2]
if "--prewarm" in sys.argv:          # what bench.py has done before its latency leg: the B = 64, N = 100 loop
    xb = torch.rand(64, 18, 256, 256, device=dev)
    for i in range(4):
        infer(net, smpl, xb, num_samples=100, seed=i)
    torch.cuda.synchronize()
    del xb
if "--pipeline" in sys.argv:         # ... through the three-stream pipeline, as bench.py runs it
    from hierarchicalprobabilistic3dhuman_amd.predict_poseMF_shapeGaussian_net import InferencePipeline
    from hierarchicalprobabilistic3dhuman_amd import sharding
    xs = [torch.rand(64, 18, 256, 256, device=dev) for _ in range(2)]
    pipe = InferencePipeline(net, smpl, num_samples=100)
    if "--normal-priority" in sys.argv:
        pipe.head_stream = torch.cuda.Stream()
    if "--no-inline" in sys.argv:
        pipe.inline_mesh = False
    sums = torch.zeros(4, dtype=torch.float64, device=dev)
    t = pipe.submit(xs[0], input_ready=False)
    for i in range(8):
        nxt = pipe.submit(xs[(i + 1) % 2], input_ready=False) if i < 7 else None
        sharding.batch_metric_sums(pipe.finish(t, seed=i, after=nxt), accumulate=sums)
        t = nxt
    torch.cuda.synchronize()
    if "--drop" in sys.argv:
        del pipe, xs
order = (False, True) if "--throughput-first" in sys.argv else (True, False)
for latency in order:
    net.set_latency_mode(latency)
    for _ in range(5):
        infer(net, smpl, x, num_samples=50, seed=1)
    eager = []
    for i in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        infer(net, smpl, x, num_samples=50, seed=i)
        torch.cuda.synchronize(); eager.append((time.perf_counter() - t0) * 1e3)
    g = GraphedInfer(net, smpl, batch=1, num_samples=50, slots=1)
    for _ in range(5):
        g(x, seed=1)
    call = []
    for i in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g(x, seed=i)
        torch.cuda.synchronize(); call.append((time.perf_counter() - t0) * 1e3)
    s = g._slots[0]
    raw = []
    for i in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(s["stream"]):
            s["graph"].replay()
        torch.cuda.synchronize(); raw.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s["stream"]):
        for i in range(reps):
            s["graph"].replay()
    torch.cuda.synchronize(); b2b = (time.perf_counter() - t0) * 1e3 / reps
    if "--bisect" in sys.argv:
        cur = torch.cuda.current_stream()
        def variant(name, fn):
            ts = []
            for i in range(reps):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                fn(i)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            print("   %-60s %.3f ms" % (name, med(ts)))
        st = s["stream"]
        def v_xcopy(i):
            with torch.cuda.stream(st):
                s["x"].copy_(x, non_blocking=True); s["graph"].replay()
        def v_keycopy(i):
            with torch.cuda.stream(st):
                s["key_dev"].copy_(s["key_host"], non_blocking=True); s["graph"].replay()
        def v_wait(i):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                s["graph"].replay()
        def v_event(i):
            with torch.cuda.stream(st):
                s["graph"].replay(); e = torch.cuda.Event(); e.record(st)
            cur.wait_event(e)
        def v_newseed(i):
            s["key_host"][0] = 1000 + i
            with torch.cuda.stream(st):
                s["key_dev"].copy_(s["key_host"], non_blocking=True); s["graph"].replay()
        variant("replay only", lambda i: v_wait.__call__ and [torch.cuda.stream(st).__enter__(), s["graph"].replay(), torch.cuda.set_stream(cur)])
        variant("x copy + replay", v_xcopy)
        variant("key copy (same key) + replay", v_keycopy)
        variant("key copy (new key each call) + replay", v_newseed)
        variant("wait_stream(caller) + replay", v_wait)
        variant("replay + event + caller waits", v_event)
    print("%s mode, batch 1, N = 50: eager %.3f ms | GraphedInfer call %.3f | bare replay %.3f | %d replays back to back %.3f ms each"
          % ("latency" if latency else "throughput", med(eager), med(call), med(raw), reps, b2b))

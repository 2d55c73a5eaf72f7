"""Cost of an event record between two kernels of one stream, and of a cross-stream wait (device side, no profiler)."""
import time
import torch

dev = torch.device("cuda:0")
a = torch.zeros(1 << 16, device=dev)
s = torch.cuda.Stream()
s2 = torch.cuda.Stream()
big = torch.randn(4096, 4096, device=dev)


def run(kind, n=2000):
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        t0 = time.perf_counter()
        for i in range(n):
            a.add_(1.0)
            if kind == "record":
                torch.cuda.Event().record(s)
            elif kind == "record_timing":
                torch.cuda.Event(enable_timing=True).record(s)
            elif kind == "record_waited":
                e = torch.cuda.Event()
                e.record(s)
                s2.wait_event(e)
            elif kind == "wait_signalled":
                s.wait_event(old)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / n * 1e6, host / n * 1e6


def gap(kind, reps=50):
    """long kernel, [marker], short kernel; time from events around the pair minus the long kernel alone"""
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tot = []
    for r in range(reps):
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            e0.record(s)
            c = big @ big
            if kind == "record":
                torch.cuda.Event().record(s)
            elif kind == "record_waited":
                e = torch.cuda.Event()
                e.record(s)
                s2.wait_event(e)
                with torch.cuda.stream(s2):
                    a2.add_(1.0)
            for _ in range(20):
                a.add_(1.0)
            e1.record(s)
        torch.cuda.synchronize()
        tot.append(e0.elapsed_time(e1) * 1e3)
    tot.sort()
    return tot[len(tot) // 2]


old = torch.cuda.Event()
old.record(s)
a2 = torch.zeros(1 << 16, device=dev)
for k in ("none", "record", "record_timing", "record_waited", "wait_signalled", "none"):
    run(k, 200)
    print("%-16s %.2f us per iteration (host %.2f)" % ((k,) + run(k)))
for k in ("none", "record", "record_waited", "none"):
    gap(k, 5)
    print("gemm + 20 adds, %-14s %.1f us" % (k, gap(k)))

"""Dev timing: hps_vertex_uncertainty alone at BASELINE configs[4] (B = 16, N = 1000) and configs[1] (B = 64, N = 100),
one-sweep / two-sweep variants (dev library modes).  HIP events, median of 10 after 3 warm-up launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import _capi, sampling_utils as su  # noqa: E402


def timed(x, reps=10):
    for _ in range(3):
        su.vertex_uncertainty(x)
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); su.vertex_uncertainty(x); e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    return ms[len(ms) // 2], ms[0], ms[-1]


def main():
    dev = torch.device("cuda:0")
    for B, N in ((16, 1000), (16, 500), (16, 250), (64, 100)):
        x = torch.randn(B, N, 6890, 3, device=dev)
        gb = x.numel() * 4 / 1e9
        with _capi.dev_library():
            for mode, name in ((0, "product"), (1, "two-sweep"), (5, "one-sweep, 32-vertex chunks")):
                if mode == 5 and N <= 768:
                    continue
                _capi.call("hps_dev_unc_mode", mode)
                med, lo, hi = timed(x)
                print("unc B=%d N=%d %-28s %.4f ms (min %.4f max %.4f)  %.2f TB/s of %.3f GB algorithmic" % (B, N, name, med, lo, hi, gb / med, gb))
            _capi.call("hps_dev_unc_mode", 0)


if __name__ == "__main__":
    main()

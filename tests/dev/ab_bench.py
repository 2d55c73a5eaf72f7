"""A/B helper: python tests/dev/ab_bench.py <hps_dev_hook> <int value> [bench args] -- sets a tuning hook, then runs bench.py
in the same process (same box, same session: the only comparison that means anything, boxes differ by +-3 %)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import _capi

with _capi.dev_library():                    # the hooks exist in libhps_dev.so only; the whole bench then runs on it
    _capi.call(sys.argv[1], int(sys.argv[2]))
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[3:]
    runpy.run_path(sys.argv[0], run_name="__main__")

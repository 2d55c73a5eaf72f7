"""bench.py with every convolution on the direct kernel (A/B against the Winograd layers on one box): python tests/dev/bench_direct.py [bench args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hierarchicalprobabilistic3dhuman_amd import resnet

_orig = resnet._ConvBN.__init__


def _init(self, *a, **k):
    _orig(self, *a, **k)
    self.use_winograd = False


resnet._ConvBN.__init__ = _init
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")

"""Walk-kernel phase cycles (library built with `make EXTRA=-DFF_PROFILE`)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth, DotProdClassifier, _lib
from sitator_amd.dotprod_classifier import LandmarkVectors
F = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
host = synth.config_host("C2")
ctx, *_ = _setup(host, 64, F, seed=31)
assert ctx.fill()[0] == 0
lib = _lib.load()
out = (ctypes.c_ulonglong * 12)()
lib.sit_debug_ff_prof(out, 1)
t = time.time()
clf = DotProdClassifier(threshold=0.45, min_samples=1)
clf.fit_centers(LandmarkVectors(ctx))
ctx.synchronize()
print("fit wall %.3f s" % (time.time() - t), ctx.info())
lib.sit_debug_ff_prof(out, 0)
v = [int(x) for x in out]
names = ["listing", "group set-up", "look-ups", "join loops", "general joins"]
joins, groups, waves = v[5], v[6], v[7]
print("joins %d groups %d waves-with-joins %d" % (joins, groups, waves))
for n, c in zip(names, v[:5]):
    print("%-14s %.3g cycles  (%.0f per join, %.0f per group, %.0f per wave)" % (n, c, c / max(joins, 1), c / max(groups, 1), c / max(waves, 1)))
print("longest walking wave: %d cycles, %d joins, %d groups; workgroups of the pair kernel whose two waves share a SIMD: %d" % (v[8], v[9], v[10], v[11]))

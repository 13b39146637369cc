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
out = (ctypes.c_ulonglong * 8)()
lib.sit_debug_ff_prof(out, 1)
t = time.time()
clf = DotProdClassifier(threshold=0.45, min_samples=1)
clf.fit_centers(LandmarkVectors(ctx))
ctx.synchronize()
print("fit wall %.3f s" % (time.time() - t), ctx.info())
lib.sit_debug_ff_prof(out, 0)
v = [int(x) for x in out]
print("cycles: sort %.3g  group-head %.3g  joins %.3g | joins %d groups %d waves %d | max wave %.3g  sum wave %.3g" % tuple(v[:3] + v[3:6] + v[6:8]))
print("per join cycles %.0f; per group head cycles %.0f; sort per wave %.0f" % (v[2] / max(v[3], 1), v[1] / max(v[4], 1), v[0] / max(v[5], 1)))

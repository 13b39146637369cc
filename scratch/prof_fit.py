"""Times fit_centers (speculative vs serial) on a C2 cut."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth, DotProdClassifier
from sitator_amd.dotprod_classifier import LandmarkVectors
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
host = synth.config_host("C2")
for mode in (["fast", "serial"] if F <= 5000 else ["fast"]):
    if mode == "serial": os.environ["SITATOR_FIT"] = "serial"
    ctx, *_ = _setup(host, 64, F, seed=31)
    os.environ.pop("SITATOR_FIT", None)
    assert ctx.fill()[0] == 0
    t = time.time()
    clf = DotProdClassifier(threshold=0.45, min_samples=1)
    clf.fit_centers(LandmarkVectors(ctx))
    ctx.synchronize()
    print(mode, "F", F, "rows", F * 64, "K", len(clf.cluster_centers), "wall %.3f s" % (time.time() - t), ctx.info())

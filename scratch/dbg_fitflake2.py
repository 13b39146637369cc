import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup, _fit_once
from sitator_amd import synth
from sitator_amd.dotprod_classifier import LandmarkVectors
host = synth.config_host("C2")
kern = sys.argv[1]
def factory():
    ctx, *_ = _setup(host, 64, 1500, seed=31, kernel=kern)
    assert ctx.fill()[0] == 0
    return LandmarkVectors(ctx)
ref = None
bad = 0
for rep in range(40):
    fast, info = _fit_once(factory, "fast")
    if ref is None:
        ref, _ = _fit_once(factory, "serial")
    ok = fast.shape == ref.shape and np.allclose(fast, ref, rtol=1e-12, atol=1e-300)
    if not ok:
        bad += 1
        d = np.argwhere(~np.isclose(fast, ref, rtol=1e-12, atol=1e-300)) if fast.shape == ref.shape else None
        print("kernel", kern, "rep", rep, "MISMATCH", fast.shape, ref.shape, None if d is None else d[:6].tolist(), info, flush=True)
print("kernel", kern, "mismatches", bad, "of 40", flush=True)

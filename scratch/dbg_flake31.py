"""The case of test_speculative_fit_equals_serial_fit that once disagreed (C2, 64 ions, 1500 frames, seed 31): step chain
and serial stream alternating, many times; which side moves?   usage: python3 scratch/dbg_flake31.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth, DotProdClassifier
from sitator_amd.dotprod_classifier import LandmarkVectors
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
host = synth.config_host("C2")


def fit(mode):
    if mode == "serial": os.environ["SITATOR_FIT"] = "serial"
    try:
        ctx, *_ = _setup(host, 64, 1500, seed=31)
        assert ctx.fill()[0] == 0
    finally:
        os.environ.pop("SITATOR_FIT", None)
    rows = ctx.rows_dense()
    clf = DotProdClassifier(threshold=0.45, min_samples=1)
    clf.fit_centers(LandmarkVectors(ctx))
    c = clf.cluster_centers.copy()
    ctx.close()
    return rows, c


rows0, good = fit("fast")
nbad = {"fast": 0, "serial": 0, "rows": 0}
for rep in range(reps):
    for mode in ("fast", "serial"):
        rows, c = fit(mode)
        if not np.array_equal(rows, rows0):
            nbad["rows"] += 1
            print("rep", rep, mode, "ROWS differ in", int((rows != rows0).sum()), "entries", flush=True)
        if c.shape != good.shape or not np.allclose(c, good, rtol=1e-12, atol=1e-300):
            nbad[mode] += 1
            print("rep", rep, mode, "centres differ: K", len(c), "vs", len(good), "entries", int((~np.isclose(c, good, rtol=1e-12, atol=1e-300)).sum()) if c.shape == good.shape else -1, flush=True)
print("differing from the first step-chain fit:", nbad, "of", reps)

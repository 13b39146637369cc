"""Randomised comparison of the step-chain fit with the serial single-workgroup stream (same arithmetic: the centres
must be bit-identical).  usage: python3 scratch/fuzz_fit.py [n_cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import DotProdClassifier

def fit(X, thr, serial):
    if serial: os.environ["SITATOR_FIT"] = "serial"
    try:
        clf = DotProdClassifier(threshold=thr, min_samples=1, max_converge_iters=30)
        clf.fit_centers(X)
    finally:
        os.environ.pop("SITATOR_FIT", None)
    return clf.cluster_centers

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(1000 + case)
    D = int(rng.choice([24, 64, 200, 512, 1500]))
    N = int(rng.choice([300, 3000, 20000, 70000]))
    maxnnz = int(rng.choice([2, 4, 7, 13]))
    nproto = int(rng.choice([5, 40, 400, 4000]))
    thr = float(rng.choice([0.3, 0.6, 0.9]))
    dwell = int(rng.choice([1, 8, 200]))          # consecutive rows drawn from one prototype (trajectory-like)
    proto = rng.integers(0, D, size=(nproto, maxnnz))
    pw = rng.uniform(0.05, 1.0, size=(nproto, maxnnz))
    X = np.zeros((N, D))
    i = 0
    while i < N:
        p = int(rng.integers(nproto))
        for _ in range(dwell):
            if i >= N: break
            k = int(rng.integers(1, maxnnz + 1))
            X[i, proto[p, :k]] = pw[p, :k] * rng.uniform(0.9, 1.1, size=k)
            i += 1
    try:
        a = fit(X, thr, False); b = fit(X, thr, True)
        ok = a.shape == b.shape and np.array_equal(a, b)
    except Exception as e:      # noqa: BLE001
        ok = False; a = b = np.zeros((0, 0)); print("  exception:", e)
    bad += 0 if ok else 1
    print("case %2d D %4d N %5d nnz<=%2d protos %4d thr %.1f dwell %3d: K %5d %s" % (case, D, N, maxnnz, nproto, thr, dwell, len(a), "ok" if ok else "MISMATCH (serial K %d)" % len(b)), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

"""Soak script (not collected by pytest): random centre sets and rows through the three forms of the assignment kernel
(packed columns in LDS, split arrays in LDS, global memory) and the oracle.  python tests/soak_predict.py [n]  (FUZZ_BASE=<first seed>)"""
import sys, os, time
os.environ.setdefault("SITATOR_PROGRESSBAR", "false")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from sitator_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
base = int(os.environ.get("FUZZ_BASE", "9000"))
bad = 0
t0 = time.time()
for it in range(n):
    rng = np.random.default_rng(base + it)
    D = int(rng.choice([8, 40, 130, 512, 1100]))
    K = int(rng.choice([1, 3, 24, 200, 600]))
    N = int(rng.choice([1, 63, 64, 65, 1000, 20000]))
    normed = bool(rng.integers(2))
    sup_max = int(rng.choice([1, 3, 8, 14]))
    centers = np.zeros((K, D))
    grid = rng.choice([0.25, 0.5, -0.5, 1.0, 0.3, 0.7, -0.9, 1e-3, 0.1])
    for k in range(K):
        sup = rng.choice(D, size=int(rng.integers(1, min(sup_max, D) + 1)), replace=False)
        centers[k, sup] = rng.random(len(sup)) if rng.random() < 0.5 else rng.choice([0.25, 0.5, -0.5, 1.0, 0.3, 0.7, -0.9], size=len(sup))
    for _ in range(K // 4):                                   # duplicates, mirror images, last-place neighbours
        a, b = rng.integers(K, size=2)
        centers[b] = centers[a] * rng.choice([1.0, -1.0, 1.0 + 2.0 ** -52, 1.0 - 2.0 ** -53])
    dev = centers.copy()
    if normed:
        for k in range(K):
            n2 = 0.0
            for d in range(D):
                n2 += centers[k, d] * centers[k, d]
            dev[k] = centers[k] / np.sqrt(n2)
    X = np.zeros((N, D))
    wmax = int(rng.choice([2, 4, 9, 20]))
    for r in range(N):
        w = int(rng.integers(0, min(wmax, D) + 1))
        cols = rng.choice(D, size=w, replace=False)
        X[r, cols] = rng.random(w) if rng.random() < 0.5 else rng.choice([1.0, 0.5, 0.25, 0.75, 1e-3, 3.0], size=w)
    for r in range(0, N, 5):
        X[r] = np.abs(centers[int(rng.integers(K))]) * rng.choice([1.0, 2.0, 0.5])
    ctx = _lib.HipContext(np.eye(3))
    ctx.set_rows_dense(X)
    ctx.set_centers(dev, normed)
    ok = True
    for thr in (0.0, float(rng.choice([0.3, 0.45, 0.8, 0.99]))):
        lab_o, conf_o = oracle.predict(X, centers, thr, normed)
        got = {}
        for name, var in (("packed", None), ("split", "SITATOR_PREDICT_REC"), ("global", "SITATOR_PREDICT_LDS")):
            if var:
                os.environ[var] = "0"
            try:
                got[name] = ctx.predict(thr)
            finally:
                if var:
                    os.environ.pop(var, None)
        for name in ("split", "global"):
            ok = ok and all(np.array_equal(a, b) for a, b in zip(got["packed"], got[name]))
        lab, conf, cnt = got["packed"]
        ok = ok and np.array_equal(lab, lab_o) and np.allclose(conf, conf_o, rtol=1e-12, atol=0) \
            and np.array_equal(cnt, np.bincount(lab[lab >= 0], minlength=K))
    ctx.close()
    if not ok:
        bad += 1
        print("MISMATCH seed", base + it, "D", D, "K", K, "N", N, "normed", normed, "sup", sup_max, "w", wmax, flush=True)
    if it % 10 == 9:
        print(it + 1, "done", "%.0fs" % (time.time() - t0), flush=True)
print("seeds", n, "bad", bad)

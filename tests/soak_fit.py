"""Soak script (not collected by pytest): speculative fit vs the ordered single-workgroup stream on medium
trajectories of every configuration, many seeds.  python tests/soak_fit.py [n]"""
import sys, os, time
os.environ.setdefault("SITATOR_PROGRESSBAR", "false")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup, _fit_once
from sitator_amd import synth
from sitator_amd.dotprod_classifier import LandmarkVectors
bad = 0
t0 = time.time()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    cfg, M, F = [("C2", 64, 2500), ("C3", 448, 300), ("C5", 160, 400), ("C1b", 4, 6000), ("C4", 256, 500)][i % 5]
    host = synth.config_host(cfg)
    seed = 700 + i
    p_hop = [None, 1 / 30.0, 1 / 300.0][i % 3]
    def factory():
        from sitator_amd import _lib
        kw = {} if p_hop is None else {"p_hop": p_hop}
        frames, sm, mm, ref = synth.make_trajectory(host, M, F, seed=seed, **kw)
        from tests.test_gpu_kernels import _ctx_from
        return LandmarkVectors(_ctx_from(host, frames, sm, mm, ref))
    fast, info = _fit_once(factory, "fast")
    serial, _ = _fit_once(factory, "serial")
    if os.environ.get("SOAK_ROWS") == "1":
        # the two contexts must have been given the same rows by their fills
        a, b = factory(), factory()
        ra, rb = a.ctx.rows_dense(), b.ctx.rows_dense()
        if not np.array_equal(ra, rb):
            d = np.where(np.any(ra != rb, axis=1))[0]
            print("   ROWS DIFFER between two fills: %d rows, first %d" % (len(d), d[0]), flush=True)
        a.ctx.close(); b.ctx.close()
    ok = fast.shape == serial.shape and np.allclose(fast, serial, rtol=1e-12, atol=1e-300)
    print(cfg, seed, "K", fast.shape[0], "batches", info["fit_batches"], "rewalks", info["fit_rewalks"], "OK" if ok else "MISMATCH", "%.0fs" % (time.time() - t0), flush=True)
    if not ok:
        # who is right?  the CPU oracle streams the rows in the reference's order and arithmetic
        from oracle import oracle as orc
        from sitator_amd import DotProdClassifier
        lv = factory()
        X = lv.ctx.rows_dense()
        exp = DotProdClassifier.__new__(DotProdClassifier)
        cen = orc.fit_centers(X, 0.45)
        again_fast, info2 = _fit_once(factory, "fast")
        again_serial, _ = _fit_once(factory, "serial")
        def eq(a, b): return a.shape == b.shape and np.allclose(a, b, rtol=1e-12, atol=1e-300)
        print("   K fast %d serial %d oracle %d | fast==oracle %s serial==oracle %s | rerun: fast==fast' %s serial==serial' %s fast'==oracle %s serial'==oracle %s | capacity %s"
              % (len(fast), len(serial), len(cen), eq(fast, cen), eq(serial, cen), eq(fast, again_fast), eq(serial, again_serial),
                 eq(again_fast, cen), eq(again_serial, cen), info.get("fit_capacity_hit")), flush=True)
    bad += not ok
print("bad", bad)

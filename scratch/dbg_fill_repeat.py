"""Is the fill bit-reproducible?  Repeated fills of the same trajectory (fresh contexts and re-fills), rows compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _ctx_from
from sitator_amd import synth
cases = [("C5", 160, 400, 722, 1 / 30.0), ("C5", 160, 400, 727, 1 / 30.0), ("C2", 64, 2500, 725, 1 / 30.0), ("C3", 448, 300, 721, 1 / 30.0)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for cfg, M, F, seed, ph in cases:
    host = synth.config_host(cfg)
    frames, sm, mm, ref = synth.make_trajectory(host, M, F, seed=seed, p_hop=ph)
    base = None
    nbad = 0
    for rep in range(reps):
        ctx = _ctx_from(host, frames, sm, mm, ref)
        for again in range(2):
            if again: assert ctx.fill()[0] == 0
            X = ctx.rows_dense()
            if base is None: base = X.copy()
            elif not np.array_equal(base, X):
                nbad += 1
                d = np.where(np.any(base != X, axis=1))[0]
                print("  %s rep %d refill %d: %d rows differ, first %d: base dims %s vals %s | now dims %s vals %s" % (
                    cfg, rep, again, len(d), d[0], np.where(base[d[0]] != 0)[0], base[d[0]][base[d[0]] != 0], np.where(X[d[0]] != 0)[0], X[d[0]][X[d[0]] != 0]), flush=True)
        ctx.close()
    print(cfg, seed, "fills", 2 * reps, "differing", nbad, flush=True)

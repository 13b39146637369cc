"""Soak script (not collected by pytest): many more seeds of tests/test_gpu_fuzz_e2e.py against the oracle;
prints every mismatch.  python tests/soak_e2e.py [n]   (FUZZ_BASE=<first seed>)"""
import sys, os, time
os.environ.setdefault("SITATOR_PROGRESSBAR", "false")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from tests.test_gpu_fuzz_e2e import _run_both
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0; kinds = {}
t0 = time.time()
for i in range(n):
    cfg = ("C1", "C1d", "C1b")[i % 3]
    seed = int(os.environ.get("FUZZ_BASE", "5000")) + i
    try:
        exp, exp_err, st, got_err, la = _run_both(oracle, cfg, seed)
        if exp_err is not None:
            ok = got_err is not None and type(got_err).__name__ == exp_err.kind and getattr(got_err, "frame", None) == getattr(exp_err, "frame", None)
            kinds[exp_err.kind] = kinds.get(exp_err.kind, 0) + 1
        else:
            ok = got_err is None and np.array_equal(st.traj, exp["labels"]) and np.allclose(np.asarray(st.site_network.centers), exp["site_centers"], rtol=1e-6, atol=1e-9) \
                and la.n_multiple_assignments == exp["n_multiple_assignments"]
            kinds["ok"] = kinds.get("ok", 0) + 1
    except Exception as e:
        ok = False; print("EXC", cfg, seed, repr(e)[:200])
    if not ok:
        bad += 1; print("MISMATCH", cfg, seed, exp_err, got_err, flush=True)
    if i % 50 == 49: print(i + 1, "done", "%.0fs" % (time.time() - t0), flush=True)
print("seeds", n, "bad", bad, kinds)

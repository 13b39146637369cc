import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from tests.test_gpu_fuzz_e2e import _run_both
cfg, seed = sys.argv[1], int(sys.argv[2])
exp, exp_err, st, got_err, la = _run_both(oracle, cfg, seed)
print("opts", la.__dict__.get("_clustering_algorithm", None), "err", exp_err, got_err, "kernel", la._ctx.info()["fill_kernel"])
lv = np.asarray(la.landmark_vectors)
print("pattern equal", np.array_equal(lv != 0, exp["lvecs"] != 0), "nnz", (lv != 0).sum(), (exp["lvecs"] != 0).sum())
d = np.abs(lv - exp["lvecs"]) / np.maximum(np.abs(exp["lvecs"]), 1e-300)
print("max rel lvec diff", d.max())
bad = np.argwhere((lv != 0) != (exp["lvecs"] != 0))
print("pattern mismatches", bad[:10])
rows = np.unique(np.argwhere(d > 1e-9)[:, 0])
print("rows differing", len(rows), rows[:10])
for r in rows[:3]:
    print(r, "got", {int(k): float(lv[r, k]) for k in np.nonzero(lv[r])[0]}, "exp", {int(k): float(exp["lvecs"][r, k]) for k in np.nonzero(exp["lvecs"][r])[0]})
m = exp["labels"] >= 0
print("conf max rel", np.max(np.abs(st.confidences[m] - exp["confs"][m]) / exp["confs"][m]), "nsites", st.site_network.n_sites, len(exp["site_centers"]))

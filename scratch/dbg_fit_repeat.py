"""Is fit_centers reproducible run to run?  The soak case that once disagreed, step chain and serial stream, many times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _ctx_from
from sitator_amd import synth, DotProdClassifier
from sitator_amd.dotprod_classifier import LandmarkVectors
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for cfg, M, F, seed in (("C5", 160, 400, 722), ("C5", 160, 400, 727), ("C1b", 4, 6000, 728)):
    host = synth.config_host(cfg)
    frames, sm, mm, ref = synth.make_trajectory(host, M, F, seed=seed, p_hop=1 / 30.0)
    res = {}
    for mode in ("fast", "serial"):
        base = None
        diff = 0
        for rep in range(reps if mode == "fast" else 3):
            if mode == "serial": os.environ["SITATOR_FIT"] = "serial"
            try:
                ctx = _ctx_from(host, frames, sm, mm, ref)
            finally:
                os.environ.pop("SITATOR_FIT", None)
            clf = DotProdClassifier(threshold=0.45, min_samples=1)
            clf.fit_centers(LandmarkVectors(ctx))
            c = clf.cluster_centers
            if base is None: base = c.copy()
            elif base.shape != c.shape or not np.array_equal(base, c):
                diff += 1
                print("  %s %s rep %d differs: K %d vs %d" % (cfg, mode, rep, len(c), len(base)), ctx.info().get("fit_capacity_hit"), flush=True)
            ctx.close()
        res[mode] = base
        print(cfg, seed, mode, "runs differing from the first:", diff, flush=True)
    print(cfg, seed, "fast == serial:", res["fast"].shape == res["serial"].shape and np.array_equal(res["fast"], res["serial"]), flush=True)

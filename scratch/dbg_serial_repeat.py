"""The serial single-workgroup fit flaked (about 1 % of the soak's cases, always the serial side): how often on the
two cases that showed it?  (Cause: a missing barrier in the founding branch of k_fit_stream, DESIGN.md section 3;
0 of 50 on both cases since.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _ctx_from
from sitator_amd import synth, DotProdClassifier
from sitator_amd.dotprod_classifier import LandmarkVectors
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for cfg, M, F, seed, ph in (("C5", 160, 400, 722, 1 / 30.0), ("C2", 64, 2500, 855, None)):
    host = synth.config_host(cfg)
    kw = {} if ph is None else {"p_hop": ph}
    frames, sm, mm, ref = synth.make_trajectory(host, M, F, seed=seed, **kw)
    def fit(mode):
        if mode == "serial": os.environ["SITATOR_FIT"] = "serial"
        try:
            ctx = _ctx_from(host, frames, sm, mm, ref)
        finally:
            os.environ.pop("SITATOR_FIT", None)
        clf = DotProdClassifier(threshold=0.45, min_samples=1)
        clf.fit_centers(LandmarkVectors(ctx))
        c = clf.cluster_centers.copy()
        ctx.close()
        return c
    good = fit("fast")
    bad = 0
    for rep in range(reps):
        fit("fast")
        s = fit("serial")
        ok = s.shape == good.shape and np.allclose(s, good, rtol=1e-12, atol=1e-300)
        if not ok:
            bad += 1
            print("  %s %d rep %d: serial K %d vs %d" % (cfg, seed, rep, len(s), len(good)), flush=True)
    print(cfg, seed, "serial fits that differ from the step chain:", bad, "of", reps, flush=True)

"""Pipelined LandmarkAnalysis.run against the separate calls on random trajectories of every configuration:
labels, confidences, centres and the zero-vector count must be identical.  usage: python3 scratch/soak_pipeline.py [seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bad = 0
cases = (("C1b", 20000), ("C2", 9000), ("C5", 8300), ("C3", 8200), ("C4", 8200), ("C2", 33000))
if os.environ.get("SOAK_ONLY"):
    cases = tuple(c for c in cases if "%s:%d" % c == os.environ["SOAK_ONLY"])
for cfg, F in cases:
    host = synth.config_host(cfg)
    for seed in range(nseeds):
        gen = synth.TrajectoryGenerator(host, synth.CONFIG_MOBILE[cfg], seed=1000 + 17 * seed, p_hop=1 / (20.0 + 60 * seed))
        frames = gen.generate(F + 131 * seed)
        sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask)
        sn.centers = host.centers; sn.vertices = host.vertices
        res = {}
        for mode in (("0", "1") if os.environ.get("SOAK_ORDER") == "01" else ("1", "0")):   # the first run of a new trajectory pays for it
            os.environ["SITATOR_PIPELINE"] = mode
            la = LandmarkAnalysis(verbose=False, check_for_zero_landmarks=False, max_mobile_per_site=64)
            t0 = time.time(); st = la.run(sn, frames); dt = time.time() - t0
            res[mode] = (st.traj.copy(), st.confidences.copy(), np.asarray(la.cluster_centers_).copy(), la.n_all_zero_lvecs, st.site_network.centers.copy(), dt, la.wall_timings.get("fill", 0.0))
        a, b = res["1"], res["0"]
        same = all(np.array_equal(a[q], b[q]) for q in (0, 1, 2, 4)) and a[3] == b[3]
        took = a[6] < 1e-3
        bad += (not same) or (not took)
        print(cfg, "F", len(frames), "seed", seed, "sites", len(a[2]), "identical" if same else "DIFFERENT", "" if took else "(pipeline not taken)", "%.3f / %.3f s" % (a[5], b[5]), flush=True)
print("bad", bad)

"""Timings of the steps after the path: jumps, JumpAnalysis, assign_to_last_known_site, SmoothSiteTrajectory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure, JumpAnalysis, SmoothSiteTrajectory
host = synth.config_host("C2"); M = 64
gen = synth.TrajectoryGenerator(host, M, seed=2)
ref = gen.reference_positions(); frames = gen.generate(100000)
sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
st = LandmarkAnalysis(verbose=False).run(sn, frames)
T = time.perf_counter
for rep in range(2):
    t0 = T(); nj = sum(1 for _ in st.jumps()); t1 = T()
    JumpAnalysis().run(st); t2 = T()
    st2 = st.copy(); t2b = T(); res = st2.assign_to_last_known_site(frame_threshold=3); t3 = T()
    sm = SmoothSiteTrajectory().run(st, threshold=3); t4 = T()
    print("rep", rep, "jumps %d %.1f ms | JumpAnalysis %.1f ms | st.copy %.1f ms | assign_last_known %.1f ms | smooth %.1f ms" % (nj, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t2b - t2), 1e3 * (t3 - t2b), 1e3 * (t4 - t3)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); JumpAnalysis().run(st); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

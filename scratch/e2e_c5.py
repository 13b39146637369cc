"""BASELINE configs[4] on one GPU: LGPS-like ragged FCC host, 160 mobile ions, full pipeline with the Markov-clustering
plugin and jump detection (frames per GPU of the 8-GPU configuration: 500 000 / 8 = 62 500)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure
F = int(sys.argv[1]) if len(sys.argv) > 1 else 62500
host = synth.config_host("C5"); M = 160
gen = synth.TrajectoryGenerator(host, M, seed=5, threads=16)
ref = gen.reference_positions()
frames = gen.generate(F)
sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
for algo, kw in (("mcl", dict(max_mobile_per_site=2)), ("dotprod", {})):
    la = LandmarkAnalysis(clustering_algorithm=algo, verbose=False, **kw)
    t = time.time(); st = la.run(sn, frames); dt = time.time() - t
    t = time.time(); jumps = st._jump_arrays(); tj = time.time() - t
    print("C5 %s F %d M %d: run %.3f s = %.3e lvec/s end to end; sites %d unassigned %.4f; %d jumps in %.4f s" % (
        algo, F, M, dt, F * M / dt, st.site_network.n_sites, st.percent_unassigned, len(jumps[0]), tj))
    print("  timers(ms)", {k: round(float(v), 2) for k, v in la.timings.items()}, "wall(s)", {k: round(v, 3) for k, v in la.wall_timings.items()})

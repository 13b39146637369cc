"""One pipelined LandmarkAnalysis.run on the C4 trajectory of scratch/ab_pipeline.py (seed 5) that took 105 s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 125000
host = synth.config_host(cfg)
gen = synth.TrajectoryGenerator(host, synth.CONFIG_MOBILE[cfg], seed=5, p_hop=1 / 200.0)
frames = gen.generate(F)
sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask)
sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(verbose=False, check_for_zero_landmarks=False)
t0 = time.time(); st = la.run(sn, frames); dt = time.time() - t0
print("run %.3f s" % dt, la.wall_timings, flush=True)
info = la._ctx.info()
print({k: info[k] for k in info if k.startswith("fit")}, "sites", st.site_network.n_sites, flush=True)

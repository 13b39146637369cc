import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure
F = int(sys.argv[1]) if len(sys.argv) > 1 else 62500
host = synth.config_host("C5"); M = 160
gen = synth.TrajectoryGenerator(host, M, seed=5, threads=16)
ref = gen.reference_positions(); frames = gen.generate(F)
sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(verbose=False)
t = time.time(); st = la.run(sn, frames); dt = time.time() - t
i = la._ctx.info()
print("C5 dotprod F", F, "run %.2f s" % dt, "sites", st.site_network.n_sites, {k: i[k] for k in i if k.startswith("fit")}, "D", len(host.centers), "row_width", i["row_width"])

import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure
from oracle import oracle as orc
host = synth.sc_grid((19, 19, 19), cell=np.diag([76.0, 77.9, 79.8]))
frames, sm, mm, ref = synth.make_trajectory(host, 8, 60, seed=5, p_hop=1/20.)
sn = SiteNetwork(Structure(ref, host.cell), sm, mm); sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(verbose=False, minimum_site_occupancy=0.0)
st = la.run(sn, frames)
exp = orc.landmark_analysis(host.cell, ref, sm, mm, host.centers, host.vertices, frames, minimum_site_occupancy=0.0)
print("sites", st.site_network.n_sites, "labels equal", np.array_equal(st.traj, exp["labels"]), "kernel", la._ctx.info()["fill_kernel"], "jumps", len(list(st.jumps())))

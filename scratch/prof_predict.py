"""Assignment kernel alone vs behind the fill: python scratch/prof_predict.py [frames] [config]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import _lib, synth, LandmarkAnalysis, SiteNetwork, Structure
F = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[cfg])
ref = gen.reference_positions()
frames = gen.generate(F)
sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(verbose=False)
la.run(sn, frames)
centers = np.asarray(la.cluster_centers_)
ctx = la._ctx
ctx.set_centers(centers / np.linalg.norm(centers, axis=1)[:, None], True)
alone = []
for i in range(30):
    ctx.predict(0.8, fetch=False)
    alone.append(ctx.timers()["predict"])
both = []
for i in range(30):
    ctx.fill(False, False, True, assign=True, predict_threshold=0.8)
    both.append((ctx.timers()["fill"], ctx.timers()["predict"]))
print("predict alone ms", np.round(alone[-5:], 4), "behind fill", np.round(both[-5:], 4), "K", len(centers))

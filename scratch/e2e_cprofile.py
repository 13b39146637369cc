"""cProfile of one end-to-end LandmarkAnalysis.run at C2 (host-side overheads around the kernels)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else synth.CONFIG_FRAMES[cfg]
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[cfg])
ref = gen.reference_positions()
frames = gen.generate(F)
def once():
    sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
    la = LandmarkAnalysis(clustering_algorithm="dotprod", verbose=False)
    t = time.time(); st = la.run(sn, frames); dt = time.time() - t
    return la, dt
la, dt = once(); print("first run %.3f s" % dt, {k: round(v, 3) for k, v in la.wall_timings.items()})
la, dt = once(); print("second run %.3f s" % dt, {k: round(v, 3) for k, v in la.wall_timings.items()})
pr = cProfile.Profile(); pr.enable(); la, dt = once(); pr.disable()
print("profiled run %.3f s" % dt, {k: round(v, 3) for k, v in la.wall_timings.items()})
pstats.Stats(pr).sort_stats("tottime").print_stats(18); pstats.Stats(pr).sort_stats("cumulative").print_stats(30)

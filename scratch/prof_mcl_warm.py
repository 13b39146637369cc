"""cProfile of a WARM LandmarkAnalysis.run with the mcl plugin at C5 (62 500 frames): python scratch/prof_mcl_warm.py"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure
cfg, F = "C5", int(sys.argv[1]) if len(sys.argv) > 1 else 62500
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=5)
ref = gen.reference_positions(); frames = gen.generate(F)
def run(prof=None):
    sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
    la = LandmarkAnalysis(verbose=False, clustering_algorithm="mcl", max_mobile_per_site=2)
    t = time.perf_counter()
    if prof: prof.enable()
    st = la.run(sn, frames)
    if prof: prof.disable()
    return time.perf_counter() - t, la
print("cold %.3f" % run()[0]); print("warm %.3f" % run()[0])
pr = cProfile.Profile(); dt, la = run(pr)
print("profiled %.3f" % dt, {k: round(v, 4) for k, v in la.wall_timings.items()}, {k: round(v, 2) for k, v in la._ctx.timers().items()})
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])

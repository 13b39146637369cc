"""Read-back of the label / confidence arrays (2 x 51 MB at C2) into fresh numpy arrays vs arrays whose pages have been
touched before: is the page-faulting of the destination what the copy costs?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure, _lib
import ctypes as C
cfg = "C2"; F = 100000
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[cfg])
frames = gen.generate(F)
sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(verbose=False); st = la.run(sn, frames)
ctx = la._ctx
N = ctx.N
for rep in range(4):
    t0 = time.perf_counter(); l, c, k = ctx.assignments(); t1 = time.perf_counter()
    labels = np.empty(N, dtype=np.int64); confs = np.empty(N)
    tt0 = time.perf_counter(); labels[::512] = 0; confs[::512] = 0; tt1 = time.perf_counter()
    counts = np.zeros(ctx.K, dtype=np.int64)
    t2 = time.perf_counter()
    ctx._check(ctx.lib.sit_get_assignments(ctx._h, _lib._i(labels), _lib._d(confs), _lib._i(counts)))
    t3 = time.perf_counter()
    assert np.array_equal(l, labels) and np.array_equal(c, confs)
    print("fresh arrays %.2f ms | touching %.2f ms | touched arrays %.2f ms" % ((t1 - t0) * 1e3, (tt1 - tt0) * 1e3, (t3 - t2) * 1e3), flush=True)
    del l, c, labels, confs

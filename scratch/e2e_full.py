"""End-to-end LandmarkAnalysis.run() on a full-size config; prints stage timings."""
import sys, os, time, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else synth.CONFIG_FRAMES[cfg]
algo = sys.argv[3] if len(sys.argv) > 3 else "dotprod"
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[cfg])
ref = gen.reference_positions()
t = time.time(); frames = gen.generate(F); tg = time.time() - t
sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(clustering_algorithm=algo, verbose=False)
t = time.time(); st = la.run(sn, frames); dt = time.time() - t
print(cfg, algo, "F", F, "M", M, "gen %.1fs" % tg, "run %.3fs" % dt, "=> %.3e lvec/s end-to-end" % (F * M / dt))
print("sites", st.site_network.n_sites, "unassigned %.4f" % st.percent_unassigned, "multi", la.n_multiple_assignments, "avg", la.avg_mobile_per_site)
print("timers(ms)", {k: round(v, 2) for k, v in la.timings.items()})
print("wall(s)", {k: round(v, 3) for k, v in la.wall_timings.items()})
print("info", la._ctx.info())
t = time.time(); nj = sum(1 for _ in st.jumps()); print("jumps", nj, "%.2fs" % (time.time() - t))
try:                                                        # device memory held by this process (contexts + idle pool)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    print("device memory in use %.2f GB; rows %d slots x %d rows" % ((total.value - free.value) / 1e9, la._ctx.row_width(), la._ctx.N))
except Exception as e:
    print("hipMemGetInfo unavailable:", e)
t = time.time(); st2 = LandmarkAnalysis(clustering_algorithm=algo, verbose=False).run(sn, frames); print("second run %.3fs" % (time.time() - t))

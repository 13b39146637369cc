"""Wall-clock stages of consecutive LandmarkAnalysis.run calls: scratch/e2e_walls.py [config] [frames] [runs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import synth, LandmarkAnalysis, SiteNetwork, Structure
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[cfg], threads=16)
frames = gen.generate(F)
sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
kw = {"clustering_algorithm": "mcl", "max_mobile_per_site": 2} if (cfg == "C5" and os.environ.get("E2E_ALGO", "mcl") == "mcl") else {}
for i in range(runs):
    t = time.perf_counter()
    la = LandmarkAnalysis(verbose=False, **kw)
    st = la.run(sn, frames)
    dt = time.perf_counter() - t
    inf = la._ctx.info()
    print("run %d: %.4f s = %.3e lvec/s; wall %s; fit steps %d serial %d bad %d" % (i, dt, F * M / dt, {k: round(v * 1e3, 1) for k, v in la.wall_timings.items()},
          inf["fit_batches"], inf["fit_serial_rows"], inf["fit_rewalks"]), flush=True)
    del la, st

"""Interleaved A/B of LandmarkAnalysis.run with and without the pipelined upload (SITATOR_PIPELINE) in one process.
usage: python3 scratch/ab_pipeline.py [config] [frames] [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
host = synth.config_host(cfg)
gen = synth.TrajectoryGenerator(host, synth.CONFIG_MOBILE[cfg], seed=5, p_hop=1 / 200.0)
frames = gen.generate(F)
sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask)
sn.centers = host.centers; sn.vertices = host.vertices
ts = {"1": [], "0": []}
for r in range(R + 1):
    for mode in ("1", "0"):
        os.environ["SITATOR_PIPELINE"] = mode
        la = LandmarkAnalysis(verbose=False, check_for_zero_landmarks=False)
        t0 = time.time(); st = la.run(sn, frames); dt = time.time() - t0
        if r: ts[mode].append(dt)
        print(cfg, "pipeline", mode, "run %.4f s" % dt, {k: round(v, 4) for k, v in la.wall_timings.items()}, flush=True)
for mode in ("1", "0"):
    print("pipeline", mode, "median %.4f min %.4f" % (float(np.median(ts[mode])), min(ts[mode])))

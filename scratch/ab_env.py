"""Interleaved A/B of LandmarkAnalysis.run under two values of one environment switch of the library (AB_VAR, values
AB_ORDER="1,0"; e.g. AB_VAR=SITATOR_PIPELINE), in one process, with the step trace of the fit if a file name is given.
AB_ORDER="1,1" shows whether successive runs of a process take the same time (they did not while every context made
its own copy stream).  usage: python3 scratch/ab_env.py [config] [frames] [repeats] [trace-file]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else synth.CONFIG_FRAMES[cfg]
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
trace = sys.argv[4] if len(sys.argv) > 4 else None
host = synth.config_host(cfg)
gen = synth.TrajectoryGenerator(host, synth.CONFIG_MOBILE[cfg], seed=synth.CONFIG_SEED[cfg])
frames = gen.generate(F)
sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask)
sn.centers = host.centers; sn.vertices = host.vertices
VAR = os.environ.get("AB_VAR", "SITATOR_PIPELINE")
ORDER = os.environ.get("AB_ORDER", "1,0").split(",")
ts = {m: [] for m in ORDER}
for r in range(R + 1):
    for mode in ORDER:
        os.environ[VAR] = mode
        if trace and r == R:
            os.environ["SITATOR_FF_TRACE"] = trace + "." + mode
        la = LandmarkAnalysis(verbose=False, check_for_zero_landmarks=False)
        t0 = time.time(); st = la.run(sn, frames); dt = time.time() - t0
        os.environ.pop("SITATOR_FF_TRACE", None)
        if r: ts[mode].append(dt)
        info = la._ctx.info()
        print(cfg, VAR, mode, "run %.4f s" % dt, "steps", info["fit_batches"], "rewalks", info["fit_rewalks"], "serial rows", info["fit_serial_rows"],
              "sites", st.site_network.n_sites,
              {k: round(v * 1e3, 1) for k, v in la.wall_timings.items()}, flush=True)
for mode in ts:
    print(VAR, mode, "median %.4f min %.4f" % (float(np.median(ts[mode])), min(ts[mode])))
if trace:
    for mode in ts:
        t = np.loadtxt(trace + "." + mode).reshape(-1, 6)
        nb = t[:, 1]
        print("trace %s: %d steps; nb <= 32: %d steps holding %d rows; 33..256: %d; > 256: %d" % (
            mode, len(t), int((nb <= 32).sum()), int(nb[nb <= 32].sum()), int(((nb > 32) & (nb <= 256)).sum()), int((nb > 256).sum())))

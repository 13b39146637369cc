"""Soak script (not collected by pytest): speculative fit vs the ordered single-workgroup stream on medium
trajectories of every configuration, many seeds.  python tests/soak_fit.py [n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup, _fit_once
from sitator_amd import synth
from sitator_amd.dotprod_classifier import LandmarkVectors
bad = 0
t0 = time.time()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    cfg, M, F = [("C2", 64, 2500), ("C3", 448, 300), ("C5", 160, 400), ("C1b", 4, 6000), ("C4", 256, 500)][i % 5]
    host = synth.config_host(cfg)
    seed = 700 + i
    p_hop = [None, 1 / 30.0, 1 / 300.0][i % 3]
    def factory():
        from sitator_amd import _lib
        kw = {} if p_hop is None else {"p_hop": p_hop}
        frames, sm, mm, ref = synth.make_trajectory(host, M, F, seed=seed, **kw)
        from tests.test_gpu_kernels import _ctx_from
        return LandmarkVectors(_ctx_from(host, frames, sm, mm, ref))
    fast, info = _fit_once(factory, "fast")
    serial, _ = _fit_once(factory, "serial")
    ok = fast.shape == serial.shape and np.allclose(fast, serial, rtol=1e-12, atol=1e-300)
    print(cfg, seed, "K", fast.shape[0], "batches", info["fit_batches"], "rewalks", info["fit_rewalks"], "OK" if ok else "MISMATCH", "%.0fs" % (time.time() - t0), flush=True)
    bad += not ok
print("bad", bad)

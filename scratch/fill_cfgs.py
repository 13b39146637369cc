"""Fill-kernel time (generation 3, default launch shape) on every benchmark configuration's per-GPU cut."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth
for cfg, F in (("C2", 100000), ("C3", 12500), ("C4", 25000), ("C5", 62500), ("C1b", 20000)):
    host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
    ctx, *_ = _setup(host, M, F, seed=2, kernel="3")
    ts = []
    for _ in range(7):
        rc, nz, err = ctx.fill(check_for_zeros=False)
        assert rc == 0
        ts.append(ctx.timers()["fill"])
    i = ctx.info()
    print("%s F=%d M=%d: min %.4f med %.4f ms  (%.2f ns/ion; nw %d fpb %d rcap %d)" % (cfg, F, M, min(ts), float(np.median(ts)), 1e6 * min(ts) / (F * M), i["waves_per_workgroup"], i["frames_per_workgroup"], i["survivors_per_wave"]), flush=True)
    ctx.close()

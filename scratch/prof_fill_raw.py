import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"
host = synth.config_host(cfg)
ctx, *_ = _setup(host, synth.CONFIG_MOBILE[cfg], F, seed=2, kernel=os.environ.get("F_KERNEL", "3"))
ts = []
for i in range(int(os.environ.get("F_REPS", "4"))):
    rc, nz, err = ctx.fill(check_for_zeros=False)
    ts.append(ctx.timers()["fill"])
if len(ts) > 12:
    print("steady fill ms %.4f (median of the last 8 of %d)" % (float(np.median(ts[-8:])), len(ts)))
print("stop", os.environ.get("SITATOR_DEBUG_STOP", "0"), cfg, "F", F, "rc", rc, "fill ms", [round(t, 4) for t in ts[:6]], "fpb", ctx.info()["frames_per_workgroup"], "kernel", ctx.info()["fill_kernel"])

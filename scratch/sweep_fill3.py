"""Fill-kernel timing sweep on the GPU box: generations 2 and 3, launch shapes of generation 3, phase ablation.
usage: python3 scratch/sweep_fill3.py [config] [frames]"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
host = synth.config_host(cfg)
M = synth.CONFIG_MOBILE[cfg]


def run(ctx, reps=5, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        ts = []
        for _ in range(reps):
            rc, nz, err = ctx.fill(check_for_zeros=False)
            assert rc == 0, (rc, ctx.message())
            ts.append(ctx.timers()["fill"])
        info = ctx.info()
    finally:
        for k in env:
            os.environ.pop(k, None)
    return min(ts), float(np.median(ts)), info


ctx3, *_ = _setup(host, M, F, seed=2, kernel="3")
ctx2, *_ = _setup(host, M, F, seed=2, kernel="2")
t3 = run(ctx3)
t2 = run(ctx2)
print("%s F=%d  gen2 min %.4f med %.4f ms | gen3 min %.4f med %.4f ms (kernel %d rcap %d nw %d fpb %d)" % (
    cfg, F, t2[0], t2[1], t3[0], t3[1], t3[2]["fill_kernel"], t3[2]["survivors_per_wave"], t3[2]["waves_per_workgroup"],
    t3[2]["frames_per_workgroup"]), flush=True)
a = ctx2.rows_dense(0, min(ctx2.N, 64 * 200)); b = ctx3.rows_dense(0, min(ctx3.N, 64 * 200))
print("pattern equal", bool(np.array_equal(a != 0, b != 0)), "max rel diff", float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300))), flush=True)
t = run(ctx3, reps=1, SITATOR_DEBUG_STOP=9)
cs = t[2]["census"]; ni = F * M
print("  census: landmark tasks/ion %.2f past the critical vertex %.2f survivors/ion %.2f ions/batch %.2f" % (cs[1] / ni, cs[0] / ni, cs[2] / ni, ni / max(cs[3], 1)), flush=True)
for stop in (1, 4):
    t = run(ctx3, SITATOR_DEBUG_STOP=stop)
    print("  gen3 debug_stop=%d: %.4f ms" % (stop, t[0]), flush=True)
shapes = [(4, 1, 48, 16), (4, 1, 40, 16), (4, 1, 32, 16), (4, 2, 48, 32), (4, 2, 64, 32), (8, 2, 40, 16), (8, 1, 24, 8), (4, 1, 40, 32), (8, 1, 40, 16)]
if len(sys.argv) > 3:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[3:]]
for shape in shapes:
    nw, fpb, rcap, iw = shape[:4]
    env = dict(SITATOR_FILL_WAVES=nw, SITATOR_FILL_FPB=fpb, SITATOR_FILL_RCAP=rcap, SITATOR_FILL_IW=iw)
    if len(shape) > 4:
        env["SITATOR_FILL_TCAP"] = shape[4]
    try:
        t = run(ctx3, **env)
        print("  nw %2d fpb %d rcap %2d iw %2d tcap %s: min %.4f med %.4f ms (nw used %d)" % (nw, fpb, rcap, iw, shape[4] if len(shape) > 4 else "-", t[0], t[1], t[2]["waves_per_workgroup"]), flush=True)
    except AssertionError as e:
        print("  nw %d fpb %d rcap %d iw %d: failed %s" % (nw, fpb, rcap, iw, e), flush=True)

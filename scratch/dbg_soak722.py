"""The soak case that disagreed (C5, seed 722, p_hop 1/30): where do the step chain and the serial stream part?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _ctx_from
from sitator_amd import synth
from sitator_amd.dotprod_classifier import LandmarkVectors
host = synth.config_host("C5")
frames, sm, mm, ref = synth.make_trajectory(host, 160, 400, seed=722, p_hop=1 / 30.0)
def stages(serial):
    if serial: os.environ["SITATOR_FIT"] = "serial"
    try:
        ctx = _ctx_from(host, frames, sm, mm, ref)
    finally:
        os.environ.pop("SITATOR_FIT", None)
    out = []
    ctx.fit_reset(); ctx.fit_push_stored_rows(0.45)
    c, n = ctx.fit_get_state(); out.append((c.copy(), n.copy()))
    for it in range(6):
        ctx.fit_reset(); ctx.fit_push_dense_rows(c, n, 0.45)
        c, n = ctx.fit_get_state(); out.append((c.copy(), n.copy()))
    return out, ctx.info()
f, fi = stages(False)
s, si = stages(True)
print({k: v for k, v in fi.items() if k.startswith("fit")})
for i, (a, b) in enumerate(zip(f, s)):
    same = a[0].shape == b[0].shape and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    print("stage", i, "K fast", len(a[0]), "serial", len(b[0]), "same", same)
    if not same:
        if a[0].shape == b[0].shape:
            d = np.where(np.any(a[0] != b[0], axis=1) | (a[1] != b[1]))[0]
            print("  centres that differ", d[:8], "counts", a[1][d][:8], b[1][d][:8])
            k = d[0]; nz = np.where((a[0][k] != 0) | (b[0][k] != 0))[0]
            print("  centre", k, "dims", nz, "\n   fast  ", a[0][k][nz], "\n   serial", b[0][k][nz])
        else:
            # first centre that differs
            m = min(len(a[0]), len(b[0]))
            d = np.where(np.any(a[0][:m] != b[0][:m], axis=1) | (a[1][:m] != b[1][:m]))[0]
            print("  first differing centre", d[:5], "counts fast", a[1][d][:5], "serial", b[1][d][:5])
        # the input rows of this stage were the previous stage's centres: bisect on the prefix
        if i > 0:
            import ctypes
            pc, pn = f[i - 1]
            def run(nrows, serial):
                if serial: os.environ["SITATOR_FIT"] = "serial"
                try:
                    from sitator_amd import _lib
                    cx = _lib.HipContext(np.eye(3)); cx.set_rows_dense(pc[:1])
                finally:
                    os.environ.pop("SITATOR_FIT", None)
                cx.fit_reset(); cx.fit_push_dense_rows(pc[:nrows], pn[:nrows], 0.45)
                return cx.fit_get_state()
            lo, hi = 0, len(pc)
            while hi - lo > 1:
                mid = (lo + hi) // 2
                x, y = run(mid, False), run(mid, True)
                if x[0].shape == y[0].shape and np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]): lo = mid
                else: hi = mid
            print("  smallest differing prefix of the stage's input:", hi, "rows")
            x, y = run(hi, False), run(hi, True)
            print("  K fast", len(x[0]), "serial", len(y[0]))
            r = pc[hi - 1]; print("  last row dims", np.where(r != 0)[0], "weight", pn[hi - 1], "nnz", int((r != 0).sum()))
        break

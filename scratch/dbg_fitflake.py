import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup, _fit_once
from sitator_amd import synth
from sitator_amd.dotprod_classifier import LandmarkVectors
host = synth.config_host("C2")
rows = []
for rep in range(6):
    ctx, *_ = _setup(host, 64, 1500, seed=31)
    assert ctx.fill()[0] == 0
    nnz, idx, val = ctx.rows_sparse()
    rows.append((nnz.copy(), idx.copy(), val.copy()))
    if rep:
        same = np.array_equal(nnz, rows[0][0])
        m = np.arange(idx.shape[0])[:, None] < nnz[None, :]
        print("rep", rep, "nnz equal", same, "idx equal", np.array_equal(idx[m], rows[0][1][m]) if same else None,
              "val equal", np.array_equal(val[m], rows[0][2][m]) if same else None, flush=True)
        if not same:
            bad = np.nonzero(nnz != rows[0][0])[0]
            print("  rows differing in nnz:", bad[:10], nnz[bad[:10]], rows[0][0][bad[:10]])
def factory():
    ctx, *_ = _setup(host, 64, 1500, seed=31)
    assert ctx.fill()[0] == 0
    return LandmarkVectors(ctx)
for rep in range(4):
    fast, info = _fit_once(factory, "fast")
    serial, _ = _fit_once(factory, "serial")
    print("fit rep", rep, fast.shape, serial.shape, "equal", np.allclose(fast, serial, rtol=1e-12, atol=1e-300), flush=True)
